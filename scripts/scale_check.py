"""Scale check on one GPU: chunked encode + decode of a large synthetic mesh, verified against the CPU oracle
(decode of the oracle's compat stream).  python scripts/scale_check.py SIDE [normals]"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from harry_amd import codec as hc, meshgen as mg
from oracle import oracle_py as op

side = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
normals = len(sys.argv) > 2
t = time.time()
mesh = mg.torus(side, side, seed=3, sigma=1e-4, normals=normals)
quant = [(1, 0, 14), (1, 1, 14), (1, 2, 14), (1, 3, 10), (1, 4, 10), (1, 5, 10)] if normals else [(1, -1, 14)]
m0 = hc.Mesh.from_arrays(mesh.verts, mesh.degrees, mesh.indices)
print(f"mesh {mesh.ntri} tris built in {time.time()-t:.1f}s", flush=True)
cx = hc.Codec(0)
cx.requant(m0, quant)
r = lambda tm: json.dumps({k: round(v, 2) if isinstance(v, float) else v for k, v in tm.items() if v})
m = m0.clone(); cx.upload(m)
t = time.time(); out = cx.write_hry(m, profile=hc.PROFILE_CHUNKED); te = time.time() - t
print(f"encode {te*1e3:.0f} ms {mesh.ntri/te/1e6:.1f} Mtri/s bytes {len(out)} bpv {8*len(out)/mesh.nv:.2f} " + r(cx.timing()), flush=True)
t = time.time(); dec = cx.read_hry(out); td = time.time() - t
print(f"decode {td*1e3:.0f} ms {mesh.ntri/td/1e6:.1f} Mtri/s " + r(cx.timing()), flush=True)
if "--no-verify" not in sys.argv:
    t = time.time()
    o = op.Mesh.from_ply(mesh.to_ply()); o.requant(quant)
    ref = op.Mesh.from_hry(o.encode().data)
    ok = np.array_equal(dec.org(), ref.org()) and np.array_equal(dec.list_data(1), ref.list_data(1)) and np.array_equal(dec.face_offsets(), ref.face_offsets())
    print(f"oracle check {'OK' if ok else 'MISMATCH'} ({time.time()-t:.1f}s)")
    assert ok
