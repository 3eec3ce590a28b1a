"""Development: the first (cold) and second chunked decode of an OBJ scene in one process, with HRY_TRACE's timeline.
    HRY_TRACE=1 python scripts/obj_cold.py [n] [flat|smooth]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from harry_amd import codec as hc
from harry_amd import meshgen as mg
from harry_amd import objgen as og

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
kind = sys.argv[2] if len(sys.argv) > 2 else "flat"
kw = dict(normals="flat") if kind == "flat" else dict(normals="smooth", tex="atlas", charts=7)
sc = og.scene(mg.torus(n, n, seed=2), **kw)
if os.environ.get("HRY_WARM_FILE"):   # the library's pages into the page cache first (is the cold cost file IO?)
    from harry_amd import _native
    t = time.time()
    with open(_native.LIB_PATH, "rb") as f:
        nbytes = len(f.read())
    print(f"read {nbytes >> 20} MB of {_native.LIB_PATH} in {(time.time() - t) * 1e3:.0f} ms", flush=True)
cx = hc.Codec(0)
m = hc.Mesh.from_obj(sc.obj, "")
data = cx.write_hry(m, profile=hc.PROFILE_CHUNKED)
for rep in range(3):
    print(f"---- decode {rep}", file=sys.stderr, flush=True)
    t = time.time()
    d = cx.read_hry(data)
    dt = time.time() - t
    print(f"decode {rep}: {dt * 1e3:.1f} ms  {cx.timing()}", flush=True)
cx.close()
