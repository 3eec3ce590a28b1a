"""Histogram of prediction candidates per vertex on the decode side (stage "ncand" of a decode with keep_stages) and the time of
the reconstruction chain, for one component of the configs[3] stand-in and for a pure triangle torus.  Development aid."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from harry_amd import codec as hc, meshgen as mg
cx = hc.Codec(0)
for name, mesh in (("mixed 1 comp", mg.multi_component(1, 221, 222, seed=4, polys="mixed")), ("mixed 8 comp + nm", mg.with_nonmanifold(mg.multi_component(8, 221, 222, seed=4, polys="mixed"), 786, 393)),
                   ("quad torus", mg.torus(221, 222, polys="quad")), ("tri torus", mg.torus(221, 222))):
    m = hc.Mesh.from_arrays(mesh.verts, mesh.degrees, mesh.indices)
    out = cx.write_hry(m, profile=hc.PROFILE_CHUNKED)
    cx.read_hry(out)
    t = time.time(); d = cx.read_hry(out, keep_stages=True); dt = time.time() - t
    nc = cx.stage("ncand")
    h = np.bincount(np.minimum(nc, 9), minlength=10)
    tm = cx.timing()
    print(f"{name:20s} nv {d.nv:7d} ntri {mesh.ntri:8d} chain {tm['k_chain_ms']:7.2f} ms ({tm['k_chain_ms']*1e6/d.nv*1.0:6.0f} ns/vertex/chain) ncand hist {h.tolist()}")
