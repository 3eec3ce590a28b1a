"""In-process executor on a configs[3]-shaped mesh: python scripts/inprocess_time.py NCOMP NCTX [REPS]
(contexts share the box's devices round-robin).  Prints the stage clocks of hry_encode_sharded / hry_decode_sharded."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from harry_amd import codec as hc, meshgen as mg, _native as nat

ncomp, nctx = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
mesh = mg.multi_component(ncomp, 221, 222, seed=4, polys="mixed")
mesh = mg.with_nonmanifold(mesh, n_edges=max(1, mesh.ntri // 1000), n_vtx=max(1, mesh.ntri // 2000))
m0 = hc.Mesh.from_arrays(mesh.verts, mesh.degrees, mesh.indices)
ndev = nat.load().hry_device_count()
mc = hc.MultiCodec([i % ndev for i in range(nctx)])
r = lambda tm: json.dumps({k: round(v, 1) if isinstance(v, float) else v for k, v in tm.items() if v})
print(f"{mesh.ntri} triangles, {nctx} contexts on {ndev} device(s)", flush=True)
for it in range(reps):
    m = m0.clone()                       # twins pending, as a reader leaves them
    t = time.perf_counter(); out = mc.write_hry(m, keep_mesh=True); te = time.perf_counter() - t
    print(f"encode {te*1e3:.0f} ms {mesh.ntri/te/1e6:.1f} Mtri/s {len(out)} B " + r(mc.last), flush=True)
    t = time.perf_counter(); d = mc.read_hry(out); td = time.perf_counter() - t
    print(f"decode {td*1e3:.0f} ms {mesh.ntri/td/1e6:.1f} Mtri/s " + r(mc.last), flush=True)
