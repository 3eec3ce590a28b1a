"""cfg4-like run: many mixed-polygon components with non-manifold edges/vertices, lossless, chunked encode + decode,
verified against the CPU oracle (optional).  python tests/tools/cfg4_check.py NCOMP NU NV [--no-verify]"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from harry_amd import codec as hc, meshgen as mg
nc, nu, nv = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
t = time.time()
mesh = mg.multi_component(nc, nu, nv, seed=4, polys="mixed")
mesh = mg.with_nonmanifold(mesh, n_edges=max(1, mesh.ntri // 1000), n_vtx=max(1, mesh.ntri // 2000))
print(f"mesh: {mesh.ntri} tris, {mesh.nf} faces, {mesh.nv} verts, built in {time.time()-t:.1f}s", flush=True)
t = time.time()
m0 = hc.Mesh.from_arrays(mesh.verts, mesh.degrees, mesh.indices)
print(f"half-edge build {time.time()-t:.1f}s", flush=True)
cx = hc.Codec(0)
r = lambda tm: json.dumps({k: round(v, 1) if isinstance(v, float) else v for k, v in tm.items() if v})
for it in range(2):   # the first pass pays for module loading, stream creation and first-touch of the pools
    m = m0.clone(); cx.upload(m)
    t = time.time(); out = cx.write_hry(m, profile=hc.PROFILE_CHUNKED); te = time.time() - t
    print(f"pass {it}: encode {te*1e3:.0f} ms {mesh.ntri/te/1e6:.1f} Mtri/s bytes {len(out)} bpv {8*len(out)/mesh.nv:.2f} " + r(cx.timing()), flush=True)
    t = time.time(); dec = cx.read_hry(out); td = time.time() - t
    print(f"pass {it}: decode {td*1e3:.0f} ms {mesh.ntri/td/1e6:.1f} Mtri/s " + r(cx.timing()), flush=True)
if "--no-verify" not in sys.argv:
    from oracle import oracle_py as op
    t = time.time()
    o = op.Mesh.from_ply(mesh.to_ply())
    ref = op.Mesh.from_hry(o.encode().data)
    ok = np.array_equal(dec.org(), ref.org()) and np.array_equal(dec.list_data(1), ref.list_data(1)) and np.array_equal(dec.face_offsets(), ref.face_offsets())
    print(f"oracle check {'OK' if ok else 'MISMATCH'} ({time.time()-t:.1f}s)")
    assert ok
