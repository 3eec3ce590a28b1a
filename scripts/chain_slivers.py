"""Development: the float chains of one GPU's share of configs[3] with and without the non-manifold slivers (each sliver is a
component of its own: a chain per attribute component).  python scripts/chain_slivers.py [COMPONENTS]"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from harry_amd import codec as hc, meshgen as mg
nc = int(sys.argv[1]) if len(sys.argv) > 1 else 128
cx = hc.Codec(0)
for name, nm in (("plain", False), ("with slivers", True)):
    mesh = mg.multi_component(nc, 221, 222, seed=4, polys="mixed")
    if nm:
        mesh = mg.with_nonmanifold(mesh, n_edges=max(1, mesh.ntri // 1000), n_vtx=max(1, mesh.ntri // 2000))
    m = hc.Mesh.from_arrays(mesh.verts, mesh.degrees, mesh.indices)
    out = cx.write_hry(m, profile=hc.PROFILE_CHUNKED)
    best = None
    for it in range(4):
        t = time.time(); d = cx.read_hry(out); dt = time.time() - t
        tm = cx.timing()
        if best is None or tm["k_chain_ms"] < best[1]["k_chain_ms"]:
            best = (dt, tm)
    print(f"{name}: {mesh.ntri} triangles, decode {best[0]*1e3:.1f} ms, k_chain {best[1]['k_chain_ms']:.2f} ms, k_predict {best[1].get('k_predict_ms', 0):.2f}, replay {best[1]['host_walk_ms']:.1f}", flush=True)
