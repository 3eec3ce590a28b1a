"""Development: the configs[2] stand-in (28 M triangles, positions 14 / normals 10 bits) through bench.py's own leg, a few times;
HRY_TRACE=1 puts the time lines on stderr (python scripts/cfg3_time.py [passes])."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
bench.stay_on_memory_node()
from harry_amd import codec as hc
cx = hc.Codec(0)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 1):
    r = bench.cfg3_leg(cx)
    print(json.dumps({k: v for k, v in r.items() if k not in ("workload", "roofline")}), flush=True)
