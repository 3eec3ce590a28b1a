"""Ad-hoc timing of lossless (float) encode/decode, chunked profile (development aid)."""
import sys, time, json, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from harry_amd import codec as hc, meshgen as mg
n = int(sys.argv[1]) if len(sys.argv) > 1 else 708
mesh = mg.torus(n, n, seed=2, sigma=1e-4, normals=len(sys.argv) > 2)
m0 = hc.Mesh.from_arrays(mesh.verts, mesh.degrees, mesh.indices)
cx = hc.Codec(0)
r = lambda tm: json.dumps({k: round(v, 3) if isinstance(v, float) else v for k, v in tm.items() if v})
for it in range(3):
    m = m0.clone(); cx.upload(m)
    t = time.time(); out = cx.write_hry(m, profile=hc.PROFILE_CHUNKED); dt = time.time() - t
    print(f"enc {it}: {dt*1e3:.1f} ms {m.ntri/dt/1e6:.2f} Mtri/s bytes {len(out)} " + r(cx.timing()))
    t = time.time(); d = cx.read_hry(out); dt = time.time() - t
    print(f"dec {it}: {dt*1e3:.1f} ms {m.ntri/dt/1e6:.2f} Mtri/s " + r(cx.timing()))
