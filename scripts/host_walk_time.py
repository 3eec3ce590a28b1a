"""Development: the host's cut-border walk of a configs[3]-shaped mesh alone (no GPU needed), with HRY_TRACE's phases.
    HRY_TRACE=1 python scripts/host_walk_time.py [components] [passes]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from harry_amd import _native as nat
from harry_amd import codec as hc
from harry_amd import meshgen as mg

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 3
m0 = mg.multi_component(n, 221, 222, seed=4, polys="mixed")
m0 = mg.with_nonmanifold(m0, n_edges=max(1, m0.ntri // 1000), n_vtx=max(1, m0.ntri // 2000))
m = hc.Mesh.from_ply(m0.to_ply())
L = nat.load()
for rep in range(passes):
    a = m.clone()
    w = C.c_void_p()
    t = time.time()
    nat.check(L.hry_walk_run_plain(a.h, C.byref(w)))
    dt = time.time() - t
    L.hry_walk_free(w)
    print(f"pass {rep}: walk of {m0.ntri} triangles in {dt * 1e3:.1f} ms = {dt / m0.ntri * 1e9:.1f} ns per triangle", flush=True)
