"""Ad-hoc: chunked encode/decode of a multi-component mesh (development aid)."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from harry_amd import codec as hc, meshgen as mg
nc, nu = int(sys.argv[1]), int(sys.argv[2])
mesh = mg.multi_component(nc, nu, nu, polys=sys.argv[3] if len(sys.argv) > 3 else "tri")
m0 = hc.Mesh.from_arrays(mesh.verts, mesh.degrees, mesh.indices)
cx = hc.Codec(0)
r = lambda tm: json.dumps({k: round(v, 2) if isinstance(v, float) else v for k, v in tm.items() if v})
for it in range(2):
    m = m0.clone(); cx.upload(m)
    t = time.time(); out = cx.write_hry(m, profile=hc.PROFILE_CHUNKED); dt = time.time() - t
    print(f"enc: {dt*1e3:.1f} ms {mesh.ntri/dt/1e6:.2f} Mtri/s bytes {len(out)} " + r(cx.timing()))
    t = time.time(); d = cx.read_hry(out); dt = time.time() - t
    print(f"dec: {dt*1e3:.1f} ms {mesh.ntri/dt/1e6:.2f} Mtri/s " + r(cx.timing()))
from oracle import oracle_py as op
o = op.Mesh.from_ply(mesh.to_ply())
ref = op.Mesh.from_hry(o.encode().data)
print("check", np.array_equal(d.org(), ref.org()) and np.array_equal(d.list_data(1), ref.list_data(1)))
