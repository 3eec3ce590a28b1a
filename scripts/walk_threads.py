import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from harry_amd import codec as hc, meshgen as mg
mesh = mg.multi_component(32, 230, 214, seed=4, polys="mixed")
m0 = hc.Mesh.from_arrays(mesh.verts, mesh.degrees, mesh.indices)
print("tris", mesh.ntri)
for th in (1, 2, 4, 8, 16, 32):
    os.environ["HRY_HOST_THREADS"] = str(th)
    m = m0.clone()
    t = time.perf_counter(); w = m.host_walk(plain=True); dt = time.perf_counter() - t
    print(f"threads {th}: {dt*1e3:.0f} ms")
