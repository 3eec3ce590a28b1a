"""Ad-hoc: does importing torch / initialising its CUDA context change the host-side timings? (development aid)"""
import sys, time, json, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mode = sys.argv[1] if len(sys.argv) > 1 else "none"
if mode != "none":
    import torch
    if mode == "cuda":
        torch.cuda.set_device(0); torch.zeros(1, device="cuda"); torch.cuda.synchronize()
import numpy as np
from harry_amd import codec as hc, meshgen as mg
mesh = mg.torus(708, 708, seed=2, sigma=1e-4)
m0 = hc.Mesh.from_arrays(mesh.verts, mesh.degrees, mesh.indices)
cx = hc.Codec(0)
cx.requant(m0, [(1, -1, 14)])
for it in range(4):
    m = m0.clone(); cx.upload(m)
    t = time.time(); out = cx.write_hry(m, profile=hc.PROFILE_CHUNKED); dt = time.time() - t
    te = cx.timing()
    t = time.time(); d = cx.read_hry(out); dd = time.time() - t
    td = cx.timing()
    print(f"{mode} it{it}: enc {dt*1e3:.1f} ms (walk {te['host_walk_ms']:.1f})  dec {dd*1e3:.1f} ms (replay {td['host_walk_ms']:.1f})")
