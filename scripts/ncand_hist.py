import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from harry_amd import codec as hc, meshgen as mg
mesh = mg.torus(708, 708, seed=2, sigma=1e-4)
m0 = hc.Mesh.from_arrays(mesh.verts, mesh.degrees, mesh.indices)
cx = hc.Codec(0)
cx.requant(m0, [(1, -1, 14)])
out = cx.write_hry(m0, profile=hc.PROFILE_CHUNKED)
cx.read_hry(out, keep_stages=True)
nc = cx.stage("ncand")
print("hist:", {int(k): int(v) for k, v in zip(*np.unique(nc, return_counts=True))})
