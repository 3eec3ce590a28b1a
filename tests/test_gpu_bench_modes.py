"""bench.py with more than one GPU, rehearsed on this box's one device: the record the driver will read the first time it has an
8-GPU node must carry -- in ONE JSON line -- the collective of `north_star` (the gather of the segments, `rccl_ranks` == N), the
same mesh through one context (`n1_same_workload_value`), a CPU baseline, a roofline whose kernel is a measured maximum, and
the compact `summary` as its last key.  Both modes run as child processes: the ranks of the launcher mode share device 0 and
talk gloo (RCCL refuses two ranks on one device), the in-process executor's contexts share device 0."""
import json
import os
import subprocess
import sys

import pytest

from tests import util

pytestmark = pytest.mark.gpu
BENCH = os.path.join(util.ROOT, "bench.py")


def run_bench(*args, env=None):
    e = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    e.update(env or {})
    r = subprocess.run([sys.executable, BENCH] + [str(a) for a in args], capture_output=True, text=True, timeout=900, env=e, cwd=util.ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]          # ONE JSON line
    return json.loads(lines[0]), lines[0]


def check_common(line, text, n):
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline", "n1_same_workload_value", "rccl_ranks", "summary"):
        assert k in line, k
    assert line["n_gpus"] == n and line["scaling"] == "weak" and line["value"] > 0 and line["n1_same_workload_value"] > 0
    assert "error" not in line
    rf = line["roofline"]
    assert rf["bound"] == "hbm" and rf["frac"] > 0 and rf["kernel_ms"] > 0
    assert rf["kernel_ms"] == pytest.approx(max(line["kernel_ms"].values()), rel=1e-3)     # the measured maximum, not a fixed name
    cb = line["cpu_baseline"]
    assert cb["value"] > 0 and cb["cores"] == 1 and cb["kind"] in ("reference", "port") and "sample" in cb
    # the digest is the LAST key and fits the tail a driver keeps
    assert list(line)[-1] == "summary"
    tail = text[text.rindex('"summary"'):]
    assert len(tail) < 1600, len(tail)
    assert line["summary"]["n1_same_workload_value"] == line["n1_same_workload_value"]


def test_bench_launcher_mode_two_ranks_on_one_gpu():
    """what the driver starts for N > 1 (here: bench.py starts its own ranks); default mode, no flag"""
    line, text = run_bench("--gpus", 2, "--share-device", "--comps-per-gpu", 4, "--steps", 1, "--warmup", 0)
    check_common(line, text, 2)
    assert line["rccl_ranks"] == 2 and line["sharded"]["rccl_ranks"] == 2
    assert line["sharded"]["merged_decode_equals_single_gpu"] is True
    assert line["sharded"]["segments"] == 2 and "gather" in line["sharded"]["collectives"]
    assert "gloo" in line["sharded"]["backend"]          # the rehearsal says it is one
    assert line["inprocess"]["contexts"] == 2 and line["inprocess"]["value"] > 0
    assert line["summary"]["rccl_ranks"] == 2 and line["summary"]["merged_ok"] is True


def test_bench_inprocess_mode_two_contexts_on_one_gpu():
    line, text = run_bench("--gpus", 2, "--inprocess", "--share-device", "--comps-per-gpu", 4, "--steps", 1, "--warmup", 0)
    check_common(line, text, 2)
    assert line["rccl_ranks"] == 0 and "in-process" in line["mode"]
    assert line["sharded"]["merged_decode_equals_single_gpu"] is True and len(line["contexts"]) == 2


def test_bench_launcher_mode_four_ranks_share_one_generated_mesh():
    """Round 6: the whole mesh is generated ONCE -- rank 0 builds it and leaves its arrays in shared memory, the other ranks map
    them -- instead of once per rank (eight generations of 100 M triangles on one node's CPU share never fitted a driver's time
    limit); four ranks on this box's one device, the same record, nothing left behind in shared memory."""
    import glob
    before = set(glob.glob("/dev/shm/hry_bench_*"))
    line, text = run_bench("--gpus", 4, "--share-device", "--comps-per-gpu", 3, "--steps", 1, "--warmup", 0)
    check_common(line, text, 4)
    assert line["rccl_ranks"] == 4 and line["sharded"]["rccl_ranks"] == 4
    assert line["sharded"]["merged_decode_equals_single_gpu"] is True and line["sharded"]["segments"] == 4
    assert line["mesh_generated_on"] == "rank 0 (the other ranks map its arrays: shared memory)"
    assert set(glob.glob("/dev/shm/hry_bench_*")) == before
