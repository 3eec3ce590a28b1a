"""Host-only timing of the cut-border walk (plain = what the chunked profile runs; model = with the operation model of the
reference stream) and of the decoder-side replay.  python scripts/walk_time.py [SIDE]"""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from harry_amd import codec as hc, meshgen as mg, _native as nat
n = int(sys.argv[1]) if len(sys.argv) > 1 else 708
mesh = mg.torus(n, n, seed=2)
m0 = hc.Mesh.from_arrays(mesh.verts, mesh.degrees, mesh.indices)
m0.twin()   # twin matching (left pending by the readers) is not part of the walk
L = nat.load()
for name, fn in (("plain", L.hry_walk_run_plain), ("model", L.hry_walk_run)):
    best = 1e9
    for i in range(9):
        m = m0.clone()
        w = C.c_void_p()
        t = time.perf_counter(); rc = fn(m.h, C.byref(w)); dt = time.perf_counter() - t
        assert rc == 0
        if i < 8:
            L.hry_walk_free(w)
        best = min(best, dt)
    print(f"walk ({name}) {best*1e3:.2f} ms  {mesh.ntri/best/1e6:.1f} Mtri/s  {best/mesh.ntri*1e9:.2f} ns/tri")
m = m0.clone()
w = C.c_void_p()
assert L.hry_walk_run_plain(m.h, C.byref(w)) == 0
best = 1e9
for i in range(9):
    mh, r = C.c_void_p(), C.c_void_p()
    t = time.perf_counter(); rc = L.hry_walk_replay(m.h, w, 0, C.byref(mh), C.byref(r)); dt = time.perf_counter() - t
    assert rc == 0
    L.hry_walk_free(r); L.hry_mesh_free(mh)
    best = min(best, dt)
print(f"replay (incl. plane split) {best*1e3:.2f} ms  {mesh.ntri/best/1e6:.1f} Mtri/s")
