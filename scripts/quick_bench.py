"""Ad-hoc timing of the compat encode on one GPU (development aid; bench.py is the contract)."""
import sys, time, json, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from harry_amd import codec as hc, meshgen as mg

n = int(sys.argv[1]) if len(sys.argv) > 1 else 708
flags = int(sys.argv[2]) if len(sys.argv) > 2 else 0
mesh = mg.torus(n, n, seed=2, sigma=1e-4)
m0 = hc.Mesh.from_arrays(mesh.verts, mesh.degrees, mesh.indices)
cx = hc.Codec(0)
cx.requant(m0, [(1, -1, 14)])
for it in range(3):
    m = m0.clone()
    cx.upload(m)
    t = time.time()
    out = cx.write_hry(m, flags=flags)
    dt = time.time() - t
    tm = cx.timing()
    print(f"iter {it}: {dt*1e3:.1f} ms  {m.ntri/dt/1e6:.2f} Mtri/s  bytes {len(out)}  " + json.dumps({k: round(v, 3) if isinstance(v, float) else v for k, v in tm.items() if v}))

# reading the reference-format stream back (serial entropy decode + replay on the host, reconstruction on the device)
for it in range(2):
    t = time.time()
    back = cx.read_hry(out)
    dt = time.time() - t
    tm = cx.timing()
    print(f"read v0.1 {it}: {dt*1e3:.1f} ms  {back.ntri/dt/1e6:.2f} Mtri/s  " + json.dumps({k: round(v, 3) if isinstance(v, float) else v for k, v in tm.items() if v}))
