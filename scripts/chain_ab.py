"""Development: chain kernel time of the headline decode, median of N decodes (run with HRY_LIB=... for a variant build)."""
import sys, time, json
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from harry_amd import codec as hc, meshgen as mg
n = int(sys.argv[1]) if len(sys.argv) > 1 else 708
mesh = mg.torus(n, n, seed=2, sigma=1e-4)
m0 = hc.Mesh.from_arrays(mesh.verts, mesh.degrees, mesh.indices)
cx = hc.Codec(0)
cx.requant(m0, [(1, -1, 14)])
m = m0.clone(); cx.upload(m)
out = cx.write_hry(m, profile=hc.PROFILE_CHUNKED)
ch, tot = [], []
for it in range(12):
    t = time.time(); cx.read_hry(out); tot.append((time.time() - t) * 1e3); ch.append(cx.timing().get("k_chain_ms", 0))
print(os.environ.get("HRY_LIB", "default").split("_")[-1], "chain median %.3f min %.3f | decode median %.3f min %.3f" % (np.median(ch[2:]), min(ch[2:]), np.median(tot[2:]), min(tot[2:])))
