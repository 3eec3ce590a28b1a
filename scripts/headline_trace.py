"""Development: the headline mesh (BASELINE configs[1]: torus 708 x 708, -l1 -q14) encoded once and decoded a few times on one
context, the last decode with HRY_TRACE's time line on stderr (python scripts/headline_trace.py [side] [passes])."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
if not os.environ.get("NO_PIN"):
    bench.stay_on_memory_node()   # (as the benchmark does: the two host loops work out of recycled buffers)
from harry_amd import codec as hc
from harry_amd import meshgen as mg
side = int(sys.argv[1]) if len(sys.argv) > 1 else 708
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 6
g = mg.torus(side, side, seed=2, sigma=1e-4)
cx = hc.Codec(0)
m = hc.Mesh.from_arrays(g.verts, g.degrees, g.indices)
cx.requant(m, [(1, -1, 14)])
cx.upload(m)
data = None
for it in range(3):
    t = time.time(); data = cx.write_hry(m, profile=hc.PROFILE_CHUNKED); te = time.time() - t
print(f"encode {te * 1e3:.2f} ms, {len(data)} bytes", flush=True)
r = lambda tm: json.dumps({k: round(v, 2) if isinstance(v, float) else v for k, v in tm.items() if v})
for it in range(passes):
    if it == passes - 1:
        sys.stderr.write("---- last decode\n"); sys.stderr.flush()
    if os.environ.get("ALTERNATE"):
        cx.write_hry(m, profile=hc.PROFILE_CHUNKED)
    t = time.time(); d = cx.read_hry(data); td = time.time() - t
    print(f"decode {td * 1e3:.2f} ms  " + r(cx.timing()), flush=True)
    del d
