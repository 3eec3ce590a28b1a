"""Multi-process path on CPU (gloo, world_size 2): the component sharding and the stream gather that bench.py uses on
N GPUs.  No compute entry points are called here (no GPU); the shard assignment, size exchange, padded gather and
concatenation are exercised with synthetic per-rank streams."""
import os
import socket
import subprocess
import sys
import textwrap

from tests import util

WORKER = textwrap.dedent('''
    import os, sys, hashlib
    sys.path.insert(0, os.environ["HRY_ROOT"])
    import torch, torch.distributed as dist
    from harry_amd import sharding

    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    # 7 components of different sizes -> greedy bin packing by triangle count, identical on every rank
    sizes = [900, 100, 400, 400, 250, 50, 700]
    parts = sharding.assign_components(sizes, world)
    assert sorted(c for p in parts for c in p) == list(range(len(sizes)))
    loads = [sum(sizes[c] for c in p) for p in parts]
    assert max(loads) - min(loads) <= max(sizes)
    mine = parts[rank]
    # stand-in for the per-component streams this rank produced
    streams = {c: hashlib.sha256(str(c).encode()).digest() * (1 + c) for c in mine}
    gathered = sharding.gather_streams(streams, len(sizes), torch.device("cpu"))
    if rank == 0:
        assert sorted(gathered) == list(range(len(sizes)))
        for c, b in gathered.items():
            assert b == hashlib.sha256(str(c).encode()).digest() * (1 + c)
        blob = sharding.concat_container(gathered)
        back = sharding.split_container(blob)
        assert back == [gathered[c] for c in range(len(sizes))]
        print("OK", len(blob))
    else:
        assert gathered is None
    # max-over-ranks timing reduction used by bench.py
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert t.item() == float(world)
    dist.barrier()
    dist.destroy_process_group()
''')


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_component_sharding_and_gather_world2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, HRY_ROOT=util.ROOT, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(free_port()), str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "OK" in r.stdout


def test_assign_components_is_deterministic_and_balanced():
    from harry_amd import sharding
    sizes = [97656] * 1024
    parts = sharding.assign_components(sizes, 8)
    assert all(len(p) == 128 for p in parts)
    assert sharding.assign_components([5, 1, 1, 1, 1, 1], 2) == [[0], [1, 2, 3, 4, 5]]
