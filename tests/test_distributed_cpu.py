"""Multi-process path on CPU (gloo, world_size 2): ONE mesh -> plan -> shards -> bounds exchange -> one segment per rank ->
gather on rank 0 -> merge -> ONE container, with the product's real split / scan / merge code (hry_shard_plan,
hry_shard_extract, the seeded host walk, hry_merge) and harry_amd.sharding's collectives.  There is no GPU here and the
product has no CPU codec, so the two device stages are stood in for by the checker: numpy computes each shard's bounds table
(what k_bounds returns) and the CPU oracle codes each shard's segment (oracle/hry_oracle.cc restates the shard container).
The merged container must decode (oracle) to exactly what the reference-format decode of the whole mesh gives."""
import os
import socket
import subprocess
import sys
import textwrap

from tests import util

WORKER = textwrap.dedent('''
    import os, sys
    sys.path.insert(0, os.environ["HRY_ROOT"])
    import numpy as np
    import torch, torch.distributed as dist
    from harry_amd import codec as hc, meshgen as mg, sharding
    from oracle import oracle_py as op          # stand-in for the device stages + checker (tests only)
    from tests import util

    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    cpu = torch.device("cpu")
    # mixed polygons, several components, non-manifold edges and vertices (components tied by shared vertices)
    gen = mg.with_nonmanifold(mg.multi_component(7, 11, 13, seed=4, polys="mixed"), 7, 4, seed=3)
    whole = hc.Mesh.from_arrays(gen.verts, gen.degrees, gen.indices)
    plan = hc.ShardPlan(whole, world)                         # same plan on every rank
    loads = [plan.triangles(r) for r in range(world)]
    assert sum(loads) == whole.ntri and min(loads) > 0
    shard = plan.extract(whole, rank)

    # bounds of the shard as k_bounds reports them: value bits + tie-breaking key (1 + index in the whole mesh)
    vof = shard.shard_elements(1)
    rows = []
    for c, (t, q, off) in enumerate(shard.list_fmt(1)):
        col = shard.component(1, c)
        imn, imx = int(np.argmin(col)), int(np.argmax(col))
        bits = lambda v: int(np.array([v], col.dtype).view(np.uint32)[0])
        mx_is_init = not (col[imx] > np.float32(1.175494351e-38))   # quant.h:33: the maximum starts at FLT_MIN
        rows.append([bits(col[imn]), bits(np.float32(1.175494351e-38)) if mx_is_init else bits(col[imx]), int(vof[imn]) + 1, 0 if mx_is_init else int(vof[imx]) + 1])
    table = np.array(rows, np.uint64).view(np.int64)
    sharding.allgather_combine(table, shard, cpu)             # all_gather + combination -> bounds of the WHOLE mesh
    wo = op.Mesh.from_ply(gen.to_ply())                       # (the oracle's own scan over the whole mesh)
    assert bytes(shard.list_min(1)) == bytes(wo.list_min(1)) and bytes(shard.list_max(1)) == bytes(wo.list_max(1))

    quant = [(1, -1, 12)]
    o = util.oracle_shard(shard, wo)
    o.set_bounds(1, bytes(shard.list_min(1)), bytes(shard.list_max(1)))
    o.requant(quant)
    seg = o.encode_chunked(1024).data                         # this rank's one-segment container
    merged = sharding.merge_on_rank0(seg, cpu)                # all_gather(sizes) + gather(payloads) + hry_merge on rank 0
    if rank == 0:
        wo.requant(quant)
        ref = op.Mesh.from_hry(wo.encode().data)              # the pin: reference-format decode of the whole mesh
        dec = op.Mesh.from_hry_chunked(merged)
        assert np.array_equal(dec.org(), ref.org()) and np.array_equal(dec.twin(), ref.twin())
        assert np.array_equal(dec.face_offsets(), ref.face_offsets())
        assert np.array_equal(dec.list_data(1), ref.list_data(1))
        print("OK", len(merged))
    else:
        assert merged is None
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


def test_one_mesh_two_ranks_one_container(tmp_path):
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
