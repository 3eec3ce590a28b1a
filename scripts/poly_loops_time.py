"""Development: the polygon loops of the host walk / replay on ONE mixed-polygon component (no GPU needed), lean against generic,
with hardware counters where the host allows them.   HRY_PERF=1 python scripts/poly_loops_time.py [SIDE]"""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from harry_amd import codec as hc, meshgen as mg, _native as nat
os.environ["HRY_HOST_THREADS"] = "1"
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
mesh = mg.torus(n, n, seed=2, polys="mixed")
m0 = hc.Mesh.from_arrays(mesh.verts, mesh.degrees, mesh.indices)
m0.twin()
L = nat.load()
for gen in (0, 1):
    if gen:
        os.environ["HRY_GENERIC_WALK"] = "1"; os.environ["HRY_GENERIC_REPLAY"] = "1"
    best = 1e9
    for i in range(5):
        m = m0.clone(); w = C.c_void_p()
        t = time.perf_counter(); rc = L.hry_walk_run_plain(m.h, C.byref(w)); dt = time.perf_counter() - t
        assert rc == 0
        if i < 4:
            L.hry_walk_free(w)
        best = min(best, dt)
    print("generic" if gen else "lean", f"walk {best*1e3:.1f} ms {best/mesh.ntri*1e9:.1f} ns/tri", flush=True)
    best = 1e9
    for i in range(5):
        mh, r = C.c_void_p(), C.c_void_p()
        t = time.perf_counter(); rc = L.hry_walk_replay(m.h, w, 0, C.byref(mh), C.byref(r)); dt = time.perf_counter() - t
        assert rc == 0
        L.hry_walk_free(r); L.hry_mesh_free(mh)
        best = min(best, dt)
    print("generic" if gen else "lean", f"replay {best*1e3:.1f} ms {best/mesh.ntri*1e9:.1f} ns/tri", flush=True)
    L.hry_walk_free(w)
