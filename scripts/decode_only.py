"""Development: decode a .hry file a few times on one context and print the times (python scripts/decode_only.py FILE [passes]);
with --make COMPONENTS the file is written first (a configs[3]-shaped mesh, chunked profile)."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from harry_amd import codec as hc
path = sys.argv[1]
cx = hc.Codec(0)
if "--make" in sys.argv:
    from harry_amd import meshgen as mg
    nc = int(sys.argv[sys.argv.index("--make") + 1])
    mesh = mg.multi_component(nc, 221, 222, seed=4, polys="mixed")
    mesh = mg.with_nonmanifold(mesh, n_edges=max(1, mesh.ntri // 1000), n_vtx=max(1, mesh.ntri // 2000))
    m = hc.Mesh.from_arrays(mesh.verts, mesh.degrees, mesh.indices)
    with open(path, "wb") as f:
        f.write(cx.write_hry(m, profile=hc.PROFILE_CHUNKED))
    print(f"{path}: {mesh.ntri} triangles", flush=True)
data = open(path, "rb").read()
r = lambda tm: json.dumps({k: round(v, 1) if isinstance(v, float) else v for k, v in tm.items() if v})
ts = []
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 4):
    t = time.time(); d = cx.read_hry(data); ts.append(time.time() - t)
    last = r(cx.timing())
    del d
print(f"decode ms: {[round(x * 1e3, 1) for x in ts]}  best {min(ts) * 1e3:.1f}  " + last, flush=True)
