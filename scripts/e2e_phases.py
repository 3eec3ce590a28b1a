"""Phase times of the end-to-end flow (PLY bytes -> .hry -> PLY bytes) on one GPU; flags: noparse, noply, threads=N"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from harry_amd import codec as hc, meshgen as mg
flags = sys.argv[1:]
mesh = mg.torus(708, 708, seed=2, sigma=1e-4)
ply = mesh.to_ply()
cx = hc.Codec(0)
m0 = hc.Mesh.from_ply(ply)
for it in range(4):
    t0 = time.perf_counter(); m = m0.clone() if "noparse" in flags else hc.Mesh.from_ply(ply); t1 = time.perf_counter()
    cx.requant(m, [(1, -1, 14)]); t2 = time.perf_counter()
    out = cx.write_hry(m, profile=hc.PROFILE_CHUNKED); t3 = time.perf_counter()
    d = cx.read_hry(out); t4 = time.perf_counter()
    p = b"" if "noply" in flags else d.to_ply(packed=True); t5 = time.perf_counter()
    print(f"{' '.join(flags):16s} from_ply {1e3*(t1-t0):.1f} ms, requant(+upload) {1e3*(t2-t1):.1f}, encode {1e3*(t3-t2):.1f}, decode {1e3*(t4-t3):.1f}, to_ply {1e3*(t5-t4):.1f}")
