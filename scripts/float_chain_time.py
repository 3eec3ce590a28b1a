"""Lossless-float reconstruction chain (k_unpredict2<float>) on ONE component: ms per decode and ns per vertex.
python scripts/float_chain_time.py [SIDE] [--flat] [--offset]   (--flat: z = 0 everywhere, the chain's fallback case;
--offset: all coordinates positive; --mixed: 40 % quads, 5 % pentagons, as BASELINE configs[3])"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from harry_amd import codec as hc, meshgen as mg

side = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 708
mesh = mg.torus(side, side, seed=2, sigma=1e-4, polys="mixed" if "--mixed" in sys.argv else "tri")
v = mesh.verts.copy()
if "--flat" in sys.argv:
    v["z"][:] = 0
if "--offset" in sys.argv:
    for k in "xyz":
        v[k] += np.float32(5.0)
mesh = mg.Mesh(v, mesh.degrees, mesh.indices, None)
m0 = hc.Mesh.from_arrays(mesh.verts, mesh.degrees, mesh.indices)
cx = hc.Codec(0)
out = cx.write_hry(m0.clone(), profile=hc.PROFILE_CHUNKED)
ts = []
for _ in range(4):
    t = time.perf_counter(); d = cx.read_hry(out); ts.append(time.perf_counter() - t)
    tm = cx.timing()
# lossless: the decoded records are a permutation of the input's
rec = lambda a: np.sort(np.ascontiguousarray(a).view(np.dtype((np.void, 12))).reshape(-1))
ok = np.array_equal(rec(d.list_data(1)), rec(m0.list_data(1)))
print(f"{mesh.ntri} triangles, {mesh.nv} vertices: decode {min(ts)*1e3:.2f} ms, k_chain {tm['k_chain_ms']:.3f} ms = {tm['k_chain_ms']*1e6/mesh.nv:.1f} ns/vertex, "
      f"replay {tm['host_walk_ms']:.2f} ms, entropy {tm['k_entropy_ms']:.2f} ms; records {'OK' if ok else 'MISMATCH'}")
assert ok
