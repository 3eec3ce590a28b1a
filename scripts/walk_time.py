import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from harry_amd import codec as hc, meshgen as mg, _native as nat
n = int(sys.argv[1]) if len(sys.argv) > 1 else 708
mesh = mg.torus(n, n, seed=2)
m0 = hc.Mesh.from_arrays(mesh.verts, mesh.degrees, mesh.indices)
L = nat.load()
best = 1e9
for i in range(7):
    m = m0.clone()
    w = C.c_void_p()
    t = time.perf_counter(); rc = L.hry_walk_run(m.h, C.byref(w)); dt = time.perf_counter() - t
    assert rc == 0
    L.hry_walk_free(w)
    best = min(best, dt)
print(f"walk {best*1e3:.1f} ms  {mesh.ntri/best/1e6:.1f} Mtri/s")
