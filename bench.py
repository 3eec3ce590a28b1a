#!/usr/bin/env python3
"""bench.py -- headline benchmark of the .hry hot path on MI355X (contract: see the task description / DESIGN.md).

One step = one pass of the hot path over one batch of synthetic input: encode the resident mesh to .hry
(host cut-border walk + every HIP kernel + D2H of the stream) and, where the profile supports it, decode it back.
Workload at N=1: BASELINE.json configs[1] -- 1 002 528-triangle closed torus, float32 xyz, `-l1 -q14`, one MI355X; the
larger configurations ride along as sub-records of the same line (`cfg3`: the 28 M-triangle torus with normals, positions 14 /
normals 10 bits; `cfg4_share`: one GPU's share of configs[3], 128 mixed-polygon non-manifold components, lossless float).
With N>1 GPUs the workload is ONE mesh shaped like configs[3]: N x 128 mixed-polygon components with non-manifold edges and
vertices, float32 xyz, lossless (12.6 M triangles per GPU: weak scaling; N = 8 is configs[3] itself, 100.6 M triangles).  Two
ways to run it:
  * under a launcher (`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`, what the driver does): one process
    per GPU.  Every rank plans the split (hry_shard_plan: components, coding order, global vertex / face / half-edge bases) and
    extracts its shard (timed once, reported as plan_ms / extract_ms); a step = k_bounds on the shard + all-gather of the
    shards' bounds (RCCL) + encode of the shard (one segment, no data-path collective) + gather of the segments on rank 0 and
    merge into ONE .hry v0.3 container + decode of the rank's own segment.  After the timed region rank 0 decodes the merged
    container and checks it against the single-GPU path on the whole mesh, and -- with the other ranks idle -- runs the same job
    through the in-process executor over all N devices (`inprocess` sub-record).
  * `python bench.py --gpus N` with no launcher around it starts the N ranks itself (the same thing, the same line).
  * `python bench.py --gpus N --inprocess`: ONE process, N device contexts (hry_encode_sharded / hry_decode_sharded: plan once, a
    worker thread per device, segments merged in host memory; no torch.distributed, no collective).  A step = the whole job from
    the host mesh: plan + upload + bounds + encode + merge, then the decode of the merged container on all devices into one mesh.
    The launcher run carries the same job as its `inprocess` sub-record.
Every N > 1 line carries `rccl_ranks` (launcher mode: == N, an all_reduce over the backend that gathers the segments),
`n1_same_workload_value` (the SAME mesh through one context on one GPU: what a scaling curve of this line is to be read against,
N = 1 of the benchmark being configs[1]), `cpu_baseline` (the reference binary on a bounded sample of the same-shaped mesh) and a
`roofline` whose kernel is the measured maximum of the step's kernels.

Prints ONE JSON line on rank 0; its LAST key, `summary`, is a compact digest (< 1.5 KB) of the sub-records.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec


def cpu_allowance():
    """CPUs this process may keep busy: affinity mask and control-group quota (what harry_amd/csrc/host/thread_pool.cpp reads)"""
    n = len(os.sched_getaffinity(0))
    try:
        a, b = open("/sys/fs/cgroup/cpu.max").read().split()
        if a != "max":
            n = max(1, min(n, int(int(a) / int(b))))
    except (OSError, ValueError):
        pass
    return n


def build_workload(n_side: int, seed: int):
    from harry_amd import meshgen as mg
    return mg.torus(n_side, n_side, seed=seed, sigma=1e-4)


def cpu_baseline(mesh, quant, budget_s=20.0):
    """The CPU oracle (a restatement pinned byte-for-byte to the reference, kind = "port") timed single-threaded on
    the same in-memory workload: encode (quantisation + .hry production) and decode, same phase boundaries."""
    from oracle import oracle_py as op
    ply = mesh.to_ply()
    base = op.Mesh.from_ply(ply)
    reps, t_enc, t_dec = 0, 0.0, 0.0
    t_start = time.perf_counter()
    hry = b""
    while reps < 1 or (time.perf_counter() - t_start < budget_s and reps < 8):
        m = base.clone()
        t0 = time.perf_counter()
        m.requant(quant)
        hry = m.encode().data
        t1 = time.perf_counter()
        op.Mesh.from_hry(hry)
        t2 = time.perf_counter()
        t_enc += t1 - t0
        t_dec += t2 - t1
        reps += 1
    ntri = mesh.ntri
    return {"value": round(ntri * reps / (t_enc + t_dec) / 1e6, 4), "unit": "Mtriangles/s", "cores": 1, "kind": "port",
            "encode_mtri_s": round(ntri * reps / t_enc / 1e6, 4), "decode_mtri_s": round(ntri * reps / t_dec / 1e6, 4),
            "hry_bytes": len(hry),
            "sample": f"full workload ({ntri} triangles, -l1 -q14), {reps} encode+decode repetitions, 1 thread"}, hry


def cpu_baseline_reference(mesh, budget_s=20.0):
    """The UNMODIFIED reference binary (oracle/_ref/harry_ref, built from /root/reference by oracle/Makefile; it travels with
    the snapshot), timed on this box's host cores on the same workload: its own phase clocks (main.cc:99-120) --
    encode = "Quantization" + "Writing output" of `harry in.ply out.hry -l1 -q14`, decode = "Reading input" of
    `harry out.hry back.ply`.  Single-threaded, like the reference.  Returns None where the binary is not available."""
    import re
    import subprocess
    import tempfile
    exe = os.path.join(ROOT, "oracle", "_ref", "harry_ref")
    if not os.path.exists(exe):
        return None
    try:
        with tempfile.TemporaryDirectory() as td:
            ply, hry, back = os.path.join(td, "in.ply"), os.path.join(td, "out.hry"), os.path.join(td, "back.ply")
            with open(ply, "wb") as f:
                f.write(mesh.to_ply())
            t_enc = t_dec = 0.0
            reps = 0
            t_start = time.perf_counter()
            ms = lambda text, what: float(re.search(what + r" took (\d+) ms", text).group(1))
            while reps < 1 or (time.perf_counter() - t_start < budget_s and reps < 8):
                e = subprocess.run([exe, ply, hry, "-l1", "-q14"], capture_output=True, text=True, timeout=300, check=True).stdout
                d = subprocess.run([exe, hry, back], capture_output=True, text=True, timeout=300, check=True).stdout
                t_enc += (ms(e, "Quantization") + ms(e, "Writing output")) * 1e-3
                t_dec += ms(d, "Reading input") * 1e-3
                reps += 1
            nbytes = os.path.getsize(hry)
        ntri = mesh.ntri
        return {"value": round(ntri * reps / (t_enc + t_dec) / 1e6, 4), "unit": "Mtriangles/s", "cores": 1, "kind": "reference",
                "encode_mtri_s": round(ntri * reps / t_enc / 1e6, 4), "decode_mtri_s": round(ntri * reps / t_dec / 1e6, 4), "hry_bytes": nbytes,
                "sample": f"full workload ({ntri} triangles, -l1 -q14), {reps} runs of the reference binary (encode = its Quantization + "
                          f"Writing output phases, decode = its Reading input phase of the .hry; 1 ms clock), 1 thread"}
    except Exception as exc:   # the checker must never take the benchmark down
        sys.stderr.write(f"reference baseline unavailable: {exc}\n")
        return None


def cpu_baseline_cfg4_sample(comps=12):
    """N > 1: the CPU baseline of the configs[3]-shaped workload on a BOUNDED sample -- `comps` of its components (the same generator,
    the same polygon mix and non-manifold rates, lossless float32 xyz) through the unmodified reference binary on this box's host
    cores (one thread, its own phase clocks: encode = "Quantization" + "Writing output", decode = "Reading input" of the .hry);
    the in-memory CPU port where the binary did not travel.  Components are independent units of this codec, so the rate per
    triangle of a sample is the rate of the whole mesh."""
    from harry_amd import meshgen as mg
    m = mg.multi_component(comps, CFG4_PER_GPU[1], CFG4_PER_GPU[2], seed=4, polys="mixed")
    m = mg.with_nonmanifold(m, n_edges=max(1, m.ntri // 1000), n_vtx=max(1, m.ntri // 2000))
    sample = (f"bounded sample of the same workload: {comps} of its mixed-polygon components ({m.ntri} triangles, 0.1 % non-manifold edges, "
              f"float32 xyz, lossless), one encode + decode")
    import re
    import subprocess
    import tempfile
    exe = os.path.join(ROOT, "oracle", "_ref", "harry_ref")
    try:
        if os.path.exists(exe):
            with tempfile.TemporaryDirectory() as td:
                ply, hry, back = os.path.join(td, "in.ply"), os.path.join(td, "out.hry"), os.path.join(td, "back.ply")
                with open(ply, "wb") as f:
                    f.write(m.to_ply())
                def ms(text, what):   # (no "Quantization" phase without -q: main.cc:104-111)
                    hit = re.search(what + r" took (\d+) ms", text)
                    return float(hit.group(1)) if hit else 0.0
                e = subprocess.run([exe, ply, hry], capture_output=True, text=True, timeout=600, check=True).stdout
                d = subprocess.run([exe, hry, back], capture_output=True, text=True, timeout=600, check=True).stdout
                t_enc = (ms(e, "Quantization") + ms(e, "Writing output")) * 1e-3
                t_dec = ms(d, "Reading input") * 1e-3
                nbytes = os.path.getsize(hry)
            kind = "reference"
            sample += " by the reference binary (its own phase clocks)"
        else:
            from oracle import oracle_py as op   # checker only, outside every timed region
            o = op.Mesh.from_ply(m.to_ply())
            t0 = time.perf_counter()
            data = o.encode().data
            t1 = time.perf_counter()
            op.Mesh.from_hry(data)
            t_enc, t_dec, nbytes = t1 - t0, time.perf_counter() - t1, len(data)
            kind = "port"
            sample += " by the in-memory CPU port (the reference binary did not travel)"
        return {"value": round(m.ntri / (t_enc + t_dec) / 1e6, 4), "unit": "Mtriangles/s", "cores": 1, "kind": kind,
                "encode_mtri_s": round(m.ntri / t_enc / 1e6, 4), "decode_mtri_s": round(m.ntri / t_dec / 1e6, 4), "hry_bytes": nbytes, "sample": sample}
    except Exception as exc:   # the checker must never take the benchmark down
        sys.stderr.write(f"cpu baseline (configs[3] sample) unavailable: {exc}\n")
        return None


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start N ranks (one process per GPU) and pass their output through.
    Nothing in this process has touched a GPU yet."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    if args.share_device:
        env["HRY_BENCH_SHARE_GPU"] = "1"               # rehearsal: every rank on device 0, gloo instead of RCCL
    raise SystemExit(subprocess.call(cmd, env=env))


CFG4_PER_GPU = (128, 221, 222)   # components per GPU, torus grid of each: 12.6 M triangles per GPU, x 8 = BASELINE configs[3]


def build_cfg4(n_gpus: int, comps_per_gpu: int = CFG4_PER_GPU[0]):
    """configs[3]-shaped: n_gpus x 128 components (40 % quads, 5 % pentagons, rest triangles), 0.1 % of the edges non-manifold,
    0.05 % non-manifold vertices, float32 xyz; deterministic"""
    from harry_amd import meshgen as mg
    m = mg.multi_component(comps_per_gpu * n_gpus, CFG4_PER_GPU[1], CFG4_PER_GPU[2], seed=4, polys="mixed")
    return mg.with_nonmanifold(m, n_edges=max(1, m.ntri // 1000), n_vtx=max(1, m.ntri // 2000))


def build_whole(n_side: int, world: int, comps_per_gpu: int = CFG4_PER_GPU[0]):
    """world == 1: BASELINE configs[1] itself; world > 1: ONE configs[3]-shaped mesh, 128 components per GPU"""
    if world == 1:
        return build_workload(n_side, seed=2)
    return build_cfg4(world, comps_per_gpu)


def _shared_mesh_paths(world):
    tag = f"hry_bench_{os.environ.get('MASTER_PORT', '0')}_{world}"
    base = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
    return [os.path.join(base, f"{tag}_{k}.npy") for k in ("verts", "degrees", "indices")]


def build_whole_once(n_side, world, rank, comps_per_gpu, dist):
    """The whole mesh on every rank of a launcher run, GENERATED ONCE: rank 0 builds it (14 - 21 s and a 33 GB peak of numpy
    temporaries for the 100 M triangles of configs[3]: eight such processes on one node's CPU share were never going to fit a driver's
    time limit) and leaves the three arrays in shared memory, the other ranks map them (the pages are shared, nothing is generated
    twice); every rank then builds its own native mesh from them, as before.  Outside every timed region."""
    if world == 1 or dist is None:
        return build_whole(n_side, world, comps_per_gpu)
    from harry_amd import meshgen as mg
    paths = _shared_mesh_paths(world)
    if rank == 0:
        mesh = build_whole(n_side, world, comps_per_gpu)
        for path, arr in zip(paths, (mesh.verts, mesh.degrees, mesh.indices)):
            np.save(path + ".tmp.npy", arr)
            os.replace(path + ".tmp.npy", path)
        dist.barrier()
        return mesh
    dist.barrier()
    verts, degrees, indices = (np.load(path, mmap_mode="r") for path in paths)
    m = mg.Mesh.__new__(mg.Mesh)          # (no copies, no sum over 78 M degrees: rank 0 built it)
    m.verts, m.degrees, m.indices, m.face_props = verts, degrees, indices, None
    return m


def release_shared_mesh(mesh, rank, dist):
    """every rank has its native copy: the files go (rank 0 keeps its arrays, the others drop their mappings)"""
    dist.barrier()
    if rank == 0:
        for path in _shared_mesh_paths(dist.get_world_size()):
            try:
                os.remove(path)
            except OSError:
                pass


def stay_on_memory_node():
    """The step is bound by two sequential host loops that work out of recycled buffers: a migration of this process to the other
    socket in mid-run leaves them in remote memory.  Confine the process to the CPUs of the memory node it is on (what `numactl
    --cpunodebind` would do); returns the node or None."""
    try:
        import ctypes
        cpu = ctypes.CDLL(None).sched_getcpu()
        allowed = os.sched_getaffinity(0)
        for node in range(64):
            path = f"/sys/devices/system/node/node{node}/cpulist"
            if not os.path.exists(path):
                break
            cpus = set()
            for part in open(path).read().strip().split(","):
                a, _, b = part.partition("-")
                cpus.update(range(int(a), int(b or a) + 1))
            if cpu in cpus and len(cpus & allowed) >= 8:
                os.sched_setaffinity(0, cpus & allowed)
                return node
    except Exception:
        pass
    return None


def _node_cpus(node: int):
    cpus = set()
    for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
        a, _, b = part.partition("-")
        cpus.update(range(int(a), int(b or a) + 1))
    return cpus


def stay_on_gpu_node(device: int, local_rank: int, local_world: int):
    """A rank of an N-GPU run works on the memory node its GPU hangs on (what the in-process executor does for its worker threads,
    device/sharded.cpp: device_cpus): the launcher starts the ranks wherever the scheduler puts them, six on one socket is as
    likely as four.  Where the GPU's PCI address does not show in sysfs (virtualised pools) the ranks are spread over the nodes
    in the order of their local ranks (GPUs 0 .. N/2 - 1 on the first socket is how these platforms are built).  Returns (node, how)."""
    n_nodes = 0
    while os.path.exists(f"/sys/devices/system/node/node{n_nodes}/cpulist"):
        n_nodes += 1
    if n_nodes < 2:
        return None, "one memory node"
    node, how = None, ""
    try:
        import torch
        pr = torch.cuda.get_device_properties(device)
        bus = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        node = int(open(f"/sys/bus/pci/devices/{bus}/numa_node").read())
        how = f"numa_node of {bus}"
    except Exception:
        node = None
    if node is None or node < 0 or node >= n_nodes:
        node = min(n_nodes - 1, local_rank * n_nodes // max(local_world, 1))
        how = "by local rank (the GPU's PCI address is not in sysfs)"
    try:
        cpus = _node_cpus(node)
        if len(cpus) >= 8:
            os.sched_setaffinity(0, cpus)
            return node, how
    except Exception:
        pass
    return None, "not bound"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)    # a step is ~17 ms of two sequential host loops, and on a shared host one step in twenty takes 23 - 28 (step_ms_each): thirty of them average that
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--side", type=int, default=708, help="torus grid side; 708 -> 1 002 528 triangles (configs[1])")
    ap.add_argument("--profile", default="auto", choices=["auto", "compat", "chunked"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-large", action="store_true", help="skip the cfg3 / cfg4_share sub-records")
    ap.add_argument("--launcher", action="store_true", help="(default since round 5) --gpus N without a launcher starts N ranks")
    ap.add_argument("--inprocess", action="store_true", help="--gpus N as ONE process with N device contexts (no torch.distributed, no collective)")
    ap.add_argument("--comps-per-gpu", type=int, default=CFG4_PER_GPU[0], help="N > 1: components per GPU of the configs[3]-shaped mesh")
    ap.add_argument("--share-device", action="store_true", help="rehearsal on a box with fewer GPUs: ranks / contexts share device 0 (ranks: gloo instead of RCCL)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        if args.inprocess:
            return inprocess_main(args)
        self_launch(args)
    stay_on_memory_node()
    # host threads of the multi-component walks / replays: the library's own choice -- the CPUs the process may keep busy (affinity
    # mask and the control group's CPU quota: the one-GPU boxes of this pool show 256 CPUs and grant 16), shared with the other
    # ranks of a launcher run (LOCAL_WORLD_SIZE).  Round 3 forced 64 threads here and ran into the quota's throttling.

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(args.gpus, 1):
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: harry_amd has no CPU path")
    # rehearsal on a one-GPU box (HRY_BENCH_SHARE_GPU=1): every rank on device 0, gloo instead of RCCL (RCCL refuses two ranks
    # on one device).  Never the measured configuration: the JSON line says so.
    share_gpu = os.environ.get("HRY_BENCH_SHARE_GPU") == "1"
    if not share_gpu and torch.cuda.device_count() < world:
        raise SystemExit(f"bench.py: {world} ranks need {world} GPUs, this node shows {torch.cuda.device_count()}")
    dev_index = 0 if share_gpu else local_rank
    backend = "gloo" if share_gpu else "nccl"
    gpu_node = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        gpu_node = stay_on_gpu_node(dev_index, local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", world)))   # before the runtime and the codec start their threads
        torch.cuda.set_device(dev_index)
        dist.init_process_group(backend)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    comm_dev = torch.device("cpu") if share_gpu else dev

    from harry_amd import codec as hc
    from harry_amd import sharding

    quant = [(1, -1, 14)] if world == 1 else []          # configs[1]: -l1 -q14; configs[3]: lossless
    mesh = build_whole_once(args.side, world, rank, args.comps_per_gpu, dist if world > 1 else None)   # the WHOLE mesh, identical on every rank
    whole = hc.Mesh.from_arrays(mesh.verts, mesh.degrees, mesh.indices)   # float32 positions, not yet quantised
    if world > 1:
        release_shared_mesh(mesh, rank, dist)
    cx = hc.Codec(dev_index)
    n_groups = n_comps = 1
    plan_ms = extract_ms = 0.0
    rccl_ranks = 1
    if world > 1:
        # proof that the collective library saw every rank: a sum of ones over the backend that carries the data
        ones = torch.ones(1, dtype=torch.int32, device=comm_dev)
        dist.all_reduce(ones)
        rccl_ranks = int(ones.item())
        t0 = time.perf_counter()
        plan = hc.ShardPlan(whole, world)                # deterministic: the same plan on every rank
        plan_ms = (time.perf_counter() - t0) * 1e3
        t0 = time.perf_counter()
        raw = plan.extract(whole, rank)                  # this rank's shard: whole groups of components, own numbering
        extract_ms = (time.perf_counter() - t0) * 1e3
        n_groups, n_comps = plan.ngroups, plan.ncomponents
        shard_tris = [plan.triangles(r) for r in range(world)]
        if rank != 0:
            del whole
    else:
        raw = whole
        shard_tris = [mesh.ntri]
    base = raw.clone()
    if world > 1:
        cx.upload(base)
        sharding.exchange_bounds(cx, base, comm_dev)
    if quant:
        cx.requant(base, quant)                          # used for the profile probe only; every timed step quantises itself

    profile = args.profile
    if profile == "auto":
        profile = "chunked"
        try:
            cx.write_hry(base.clone(), profile=hc.PROFILE_CHUNKED)
        except hc.HryError:
            profile = "compat"
    if world > 1 and profile != "chunked":
        raise SystemExit("the reference's single stream (compat) does not shard: replicas only (DESIGN.md section 5)")
    pid = hc.PROFILE_CHUNKED if profile == "chunked" else hc.PROFILE_COMPAT
    can_decode = True
    try:
        cx.read_hry(cx.write_hry(base.clone(), profile=pid), partial=world > 1)
    except hc.HryError:
        can_decode = False

    def one_step():
        """returns (container bytes, merged container on rank 0, timings); inputs are resident in HBM before the timed region"""
        m = raw.clone()
        cx.upload(m)                                     # float records + connectivity resident in HBM
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if world > 1:
            sharding.exchange_bounds(cx, m, comm_dev)    # k_bounds on the shard + all-gather (RCCL) -> bounds of the whole mesh
        if quant:
            cx.requant(m, quant)                         # encode = quantisation (bounds, float -> uint14) + .hry production
        t_q = time.perf_counter()
        out = cx.write_hry(m, profile=pid, as_buffer=True)   # the buffer hry_encode returns, as a C caller holds it (no copy into bytes)
        t1 = time.perf_counter()
        tm_e = cx.timing()
        tm_e["requant_ms"] = (t_q - t0) * 1e3
        gather = sharding.SegmentGather(out, comm_dev, as_buffer=True) if world > 1 else None    # segments -> rank 0, overlapping the decode below
        t_g = time.perf_counter()
        tm_d = {}
        if can_decode:
            cx.read_hry(out, partial=world > 1)          # every rank decodes its own segment
            tm_d = cx.timing()
        t2 = time.perf_counter()
        merged = gather.finish() if gather else None     # rank 0: hry_merge -> ONE container
        t3 = time.perf_counter()
        tm_e.update({"dec_" + k: v for k, v in tm_d.items()})
        tm_e["gather_merge_ms"] = (t_g - t1 + t3 - t2) * 1e3
        return out, merged, t1 - t0, t2 - t_g, t_g - t1 + t3 - t2, tm_e

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step()                                       # also warms the communicator up
    barrier()
    enc_s = dec_s = comm_s = 0.0
    timings = []
    out, merged = b"", None
    for _ in range(args.steps):
        out, merged, te, td, tg, tm = one_step()
        enc_s += te
        dec_s += td
        comm_s += tg
        tm["step_ms"] = (te + td + tg) * 1e3
        timings.append(tm)
    barrier()
    total = torch.tensor([enc_s + dec_s + comm_s, enc_s, dec_s, comm_s], dtype=torch.float64, device=comm_dev)
    if world > 1:
        dist.all_reduce(total, op=dist.ReduceOp.MAX)
    t_all, t_enc, t_dec, t_comm = (float(x) for x in total.tolist())

    if rank == 0:
        ntri = mesh.ntri                                  # the whole mesh
        step_ms = t_all / args.steps * 1e3
        value = ntri * args.steps / t_all / 1e6
        med = lambda k: float(np.median([t.get(k, 0.0) for t in timings]))
        # the dominant kernel of a step (HIP-event times taken inside the library on the codec stream)
        cands = {"k_rchain": med("k_rchain_ms"), "k_chunk_encode": med("k_entropy_ms") if profile == "chunked" else 0.0,
                 "k_predict_vtx": med("k_predict_ms"), "k_chunk_decode": med("dec_k_entropy_ms"),
                 ("k_unpredict3" if world == 1 else "k_unpredict2<float>"): med("dec_k_chain_ms")}
        dom_name = max(cands, key=cands.get)
        dom_ms = cands[dom_name]
        # algorithmic bytes per launch: every input array once + the stream once =
        # vertex records + 4 B per half-edge + |hry|  (same bytes in the opposite direction for decode); rank 0's shard
        alg_bytes = base.nv * base.list_stride(1) + 4 * base.ne + len(out)
        roof = {"bound": "hbm", "kernel": dom_name, "achieved": round(alg_bytes / (dom_ms * 1e-3) / 1e9, 3) if dom_ms > 0 else None,
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "traffic": None,
                "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms": round(dom_ms, 4)}
        roof["frac"] = round(roof["achieved"] / HBM_PEAK_GBS, 6) if roof["achieved"] else None
        # HBM bytes per launch of that kernel from the PMC passes committed under profiles/ (rocprofv3 cannot run inside the
        # timed program; scripts/collect_profiles.sh collects FETCH_SIZE and WRITE_SIZE in their own passes on this workload)
        for rnd in ("r6", "r5", "r4", "r3", "r2", "r1"):
            try:
                with open(os.path.join(ROOT, "profiles", rnd, "traffic.json")) as f:
                    tr = json.load(f)["kernels"]
                hit = [v for k, v in tr.items() if k.split("<")[0].split("_range")[0] == dom_name]
                if hit and args.side == 708 and profile == "chunked":
                    # MI355X_MICROARCH.md, HBM section: on gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes -> doubled;
                    # WRITE_SIZE is exact.  Per decode of the workload (all launches of the kernel together).
                    roof["traffic"] = 2 * hit[0]["fetch_bytes"] + hit[0]["write_bytes"]
                    roof["traffic_raw"] = {"FETCH_SIZE_bytes": hit[0]["fetch_bytes"], "WRITE_SIZE_bytes": hit[0]["write_bytes"]}
                    roof["traffic_source"] = f"profiles/{rnd}/traffic.json (rocprofv3 --pmc, separate passes; FETCH_SIZE x2 per the gfx950 note)"
                    break
            except (OSError, KeyError, ValueError):
                pass
        host_ms = med("host_walk_ms") + med("dec_host_walk_ms")
        per_gpu = (f"closed torus {args.side}x{args.side}, {mesh.ntri} triangles, float32 xyz, -l1 -q14 (BASELINE configs[1])" if world == 1 else
                   f"{args.comps_per_gpu} mixed-polygon components with non-manifold edges / vertices, {mesh.ntri // world} triangles, float32 xyz, lossless")
        k_chain_name = "k_unpredict3" if world == 1 else "k_unpredict2<float>"
        line = {
            "metric": "Mtriangles/s encode+decode", "value": round(value, 4), "unit": "Mtriangles/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(step_ms, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": "u8/u16 residual bytes, u32 range-coder registers (compat profile: u64)" if world == 1 else "u8 residual bytes of f32 values, u32 range-coder registers", "data": "synthetic",
            "config": {"workload": per_gpu if world == 1 else f"ONE mesh shaped like BASELINE configs[3] ({ntri} triangles, {n_comps} components, {n_groups} groups); per GPU: {per_gpu}; sharded by connected component",
                       "profile": profile, "decode_in_step": can_decode, "parallelism": f"component-sharded x{world}, one process per GPU", "inputs_resident": True,
                       "host_threads": os.environ.get("HRY_HOST_THREADS", "library default"), "cpus_allowed": cpu_allowance()},
            "encode_mtri_s": round(ntri * args.steps / t_enc / 1e6, 4),
            "decode_mtri_s": round(ntri * args.steps / t_dec / 1e6, 4) if can_decode and t_dec > 0 else None,
            "hry_bytes": len(out), "bits_per_vertex": round(8 * len(out) / max(base.nv, 1), 4),
            "host_fraction": round(host_ms / step_ms, 4) if step_ms > 0 else None,
            **({"mesh_generated_on": "rank 0 (the other ranks map its arrays: shared memory)"} if world > 1 else {}),
            "step_ms_each": [round(t["step_ms"], 2) for t in timings],      # (rank 0's steps: value is their mean, outliers included)
            "stage_ms": {k: round(med(k), 4) for k in ("requant_ms", "host_walk_ms", "h2d_ms", "device_ms", "k_predict_ms", "k_model_ms", "k_rchain_ms", "k_entropy_ms", "total_ms",
                                                        "gather_merge_ms", "dec_host_walk_ms", "dec_k_entropy_ms", "dec_k_chain_ms", "dec_total_ms")},
            "kernel_ms": {k: round(v, 4) for k, v in cands.items()},
            "roofline": roof,
        }
        if world > 1:
            # the merged container: decode it whole on this GPU and compare with the single-GPU path on the whole mesh
            # (outside the timed region)
            ok = None
            try:
                got = cx.read_hry(merged)
                one = whole.clone()
                cx.requant(one, quant)
                ref = cx.read_hry(cx.write_hry(one, profile=pid))
                ok = bool(got.nv == ref.nv and np.array_equal(got.face_offsets(), ref.face_offsets()) and np.array_equal(got.org(), ref.org())
                          and np.array_equal(got.twin(), ref.twin()) and np.array_equal(got.list_data(1), ref.list_data(1)))
            except Exception as exc:
                sys.stderr.write(f"merged-container check failed: {exc}\n")
                ok = False
            line["mode"] = "launcher: one process per GPU (torch.distributed, backend below)"
            line["rccl_ranks"] = rccl_ranks
            incl = t_all + args.steps * (plan_ms + extract_ms) * 1e-3
            line["value_including_plan"] = round(ntri * args.steps / incl / 1e6, 4)
            line["value_including_plan_note"] = "every step charged with rank 0's plan + extract of its shard (timed once before the steps: the mesh is static input)"
            line["sharded"] = {"memory_node_of_rank0": list(gpu_node) if gpu_node else None, "rccl_ranks": rccl_ranks, "rccl_ranks_how": "all_reduce(ones) over the backend below", "plan_ms": round(plan_ms, 2), "extract_ms": round(extract_ms, 2),
                               "plan_extract_note": "every rank plans the whole mesh and extracts its own shard, once, before the timed steps (the mesh is static input); "
                                                    "the in-process executor's numbers include both",
                               "backend": "nccl (RCCL over xGMI)" if backend == "nccl" else "gloo (one-GPU rehearsal, not a measurement)",
                               "components": n_comps, "groups": n_groups, "triangles_per_rank": shard_tris,
                               "merged_container_bytes": len(merged), "segments": world,
                               "merged_decode_equals_single_gpu": ok, "comm_ms_per_step": round(t_comm / args.steps * 1e3, 3),
                               "collectives": "all_gather(bounds, 96 B/rank) + all_gather(sizes) + gather(segments, asynchronous: overlaps the decode) per step"}
            if ok is False:
                line["error"] = "merged container does not decode to the single-GPU result"
            # the same workload through ONE context, unsharded (rank 0's GPU): what the driver's per-N efficiency should be read against,
            # since N = 1 of this benchmark is configs[1], another workload
            try:
                share = build_cfg4(1, args.comps_per_gpu)
                sm = hc.Mesh.from_arrays(share.verts, share.degrees, share.indices)
                ts = []
                for _ in range(3):
                    a = sm.clone(); cx.upload(a); torch.cuda.synchronize()
                    t0 = time.perf_counter(); ob = cx.write_hry(a, profile=pid); cx.read_hry(ob); ts.append(time.perf_counter() - t0)
                line["sharded"]["one_gpu_same_shape_mtri_s"] = round(share.ntri / min(ts[1:]) / 1e6, 3)
                line["sharded"]["weak_scaling_efficiency_vs_same_shape"] = round(value / (world * line["sharded"]["one_gpu_same_shape_mtri_s"]), 4)
                del sm, share
                # ... and the SAME mesh (all N x 128 components) through this one context: what N devices are to be compared with
                ts = []
                for _ in range(2):
                    a = whole.clone(); cx.upload(a); torch.cuda.synchronize()
                    t0 = time.perf_counter(); ob = cx.write_hry(a, profile=pid, as_buffer=True); cx.read_hry(ob); ts.append(time.perf_counter() - t0)
                line["sharded"]["one_context_same_mesh_mtri_s"] = round(ntri / ts[-1] / 1e6, 3)
                line["sharded"]["efficiency_vs_one_context"] = round(value / (world * line["sharded"]["one_context_same_mesh_mtri_s"]), 4)
                line["sharded"]["speedup_vs_one_context"] = round(value / line["sharded"]["one_context_same_mesh_mtri_s"], 3)
                line["n1_same_workload_value"] = line["sharded"]["one_context_same_mesh_mtri_s"]
                line["n1_same_workload_note"] = ("the SAME mesh (all N x components) through ONE context on rank 0's GPU, unsharded, inputs resident, encode + decode: "
                                                 "read this line's scaling against it -- N = 1 of this benchmark is configs[1], another workload")
            except Exception as exc:
                sys.stderr.write(f"same-shape single-GPU leg failed: {exc}\n")
        # end to end, as the `harry in.ply out.hry -l1 -q14` / `harry out.hry back.ply` command lines see it: PLY bytes ->
        # parse + twin matching -> upload -> quantisation -> .hry bytes, and .hry bytes -> mesh -> binary PLY bytes.  Outside the timed
        # region; not part of `value`.
        try:
            if world != 1:
                raise RuntimeError("skipped on multi-GPU runs (rank 0 only work would hold the other ranks at shutdown)")
            ply_bytes = mesh.to_ply()
            e2e_e, e2e_d = [], []
            for _ in range(3):
                t0 = time.perf_counter()
                mm = hc.Mesh.from_ply(ply_bytes)
                cx.requant(mm, quant)
                hb = cx.write_hry(mm, profile=pid)
                t1 = time.perf_counter()
                back = cx.read_hry(hb).to_ply() if can_decode else b""
                t2 = time.perf_counter()
                e2e_e.append(t1 - t0); e2e_d.append(t2 - t1)
            line["end_to_end"] = {"encode_ms": round(float(np.median(e2e_e)) * 1e3, 3), "decode_ms": round(float(np.median(e2e_d)) * 1e3, 3),
                                  "encode_mtri_s": round(ntri / float(np.median(e2e_e)) / 1e6, 3),
                                  "decode_mtri_s": round(ntri / float(np.median(e2e_d)) / 1e6, 3) if can_decode else None,
                                  "ply_bytes": len(ply_bytes), "what": "PLY bytes -> .hry bytes (parse, twin matching, upload, quantisation, encode); .hry bytes -> PLY bytes"}
        except Exception as exc:
            if world == 1:
                sys.stderr.write(f"end-to-end leg failed: {exc}\n")
        if world == 1:
            # the reference's "encode" starts from a host mesh (main.cc:104-117): the same step with the upload inside
            try:
                ts = []
                for _ in range(4):
                    m = raw.clone()
                    t0 = time.perf_counter()
                    cx.requant(m, quant)                 # uploads the records + connectivity first
                    cx.write_hry(m, profile=pid)
                    ts.append(time.perf_counter() - t0)
                line["encode_from_host_mtri_s"] = round(ntri / float(np.median(ts[1:])) / 1e6, 4)
                line["encode_from_host_ms"] = round(float(np.median(ts[1:])) * 1e3, 3)
            except Exception as exc:
                sys.stderr.write(f"encode-from-host leg failed: {exc}\n")
        if world == 1 and not args.no_large and args.side == 708:
            for name, leg in (("cfg3", cfg3_leg), ("cfg4_share", cfg4_share_leg), ("cfg4_end_to_end", cfg4_end_to_end_leg), ("cfg4_full", cfg4_full_leg)):
                try:
                    line[name] = leg(cx)
                except Exception as exc:
                    sys.stderr.write(f"{name} leg failed: {exc}\n")
        if not args.no_cpu_baseline and world == 1:
            cb, ref_hry = cpu_baseline(mesh, quant, budget_s=8.0)
            ref = cpu_baseline_reference(mesh, budget_s=12.0)
            if ref is not None:
                line["cpu_baseline"] = ref            # the reference itself
                line["cpu_baseline_port"] = cb        # the oracle (CPU restatement) in memory, same workload
                line["reference_bytes_equal_port"] = bool(ref["hry_bytes"] == len(ref_hry))
            else:
                line["cpu_baseline"] = cb
            line["bits_per_vertex_cpu_ref"] = round(8 * len(ref_hry) / base.nv, 4)
            if profile == "compat":
                line["byte_identical_to_cpu_ref"] = bool(out == ref_hry)
            # the drop-in number next to the headline: the reference's own single stream (compat profile, .hry v0.1) from the same
            # resident mesh -- byte-identical to the CPU reference's file, the serial range recurrence on a host core behind the kernels
            try:
                ts, tds, cb_out, tms = [], [], b"", []
                for _ in range(1 + 3):
                    m = raw.clone()
                    cx.upload(m)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    cx.requant(m, quant)
                    cb_out = cx.write_hry(m, profile=hc.PROFILE_COMPAT)
                    ts.append(time.perf_counter() - t0)
                    tms.append(cx.timing())
                    t0 = time.perf_counter()
                    cx.read_hry(cb_out)
                    tds.append(time.perf_counter() - t0)
                t_c, t_d = float(np.median(ts[1:])), float(np.median(tds[1:]))
                line["compat"] = {"encode_mtri_s": round(ntri / t_c / 1e6, 4), "encode_ms": round(t_c * 1e3, 3), "hry_bytes": len(cb_out),
                                  "decode_mtri_s": round(ntri / t_d / 1e6, 4), "decode_ms": round(t_d * 1e3, 3),
                                  "value": round(ntri / (t_c + t_d) / 1e6, 4),
                                  "vs_reference_binary_decode": round(ntri / t_d / 1e6 / line["cpu_baseline"]["decode_mtri_s"], 3) if line["cpu_baseline"].get("decode_mtri_s") else None,
                                  "byte_identical_to_cpu_ref": bool(cb_out == ref_hry),
                                  "vs_reference_binary_encode": round(ntri / t_c / 1e6 / line["cpu_baseline"]["encode_mtri_s"], 3),
                                  "stage_ms": {k: round(float(np.median([t[k] for t in tms[1:]])), 3) for k in ("host_walk_ms", "k_predict_ms", "k_model_ms", "k_rchain_ms", "device_ms", "total_ms")},
                                  "what": "quantisation + .hry v0.1 from the resident mesh; k_rchain_ms = the serial recurrence (host core, streamed behind the kernels)"}
            except Exception as exc:
                sys.stderr.write(f"compat leg failed: {exc}\n")
            # SURVEY.md section 8 row f3 next to it: an OBJ scene (positions per vertex, texture coordinates and normals per corner,
            # shared between corners) through the same reference stream; checked against the oracle, timed beside the reference binary
            try:
                line["obj"] = obj_leg(cx)
            except Exception as exc:
                sys.stderr.write(f"obj leg failed: {exc}\n")
        if world == 1 and "cfg4_full" in line:
            line["cfg4_full"] = line.pop("cfg4_full")     # (the named size last before the digest)
        if world > 1:
            # with the other ranks idle (they wait on the host-side barrier below): the same mesh through the in-process executor
            # over all N devices -- ONE process, plan + upload inside the timed call (rehearsal: the contexts share device 0 too)
            try:
                line["inprocess"] = inprocess_leg(hc, whole, [0] * world if share_gpu else list(range(world)), quant, mesh.ntri, reps=2 if share_gpu else 3)
            except Exception as exc:
                sys.stderr.write(f"in-process leg failed: {exc}\n")
            if not args.no_cpu_baseline:
                cb = cpu_baseline_cfg4_sample(min(12, args.comps_per_gpu * world))
                if cb is not None:
                    line["cpu_baseline"] = cb
        emit(line)
    cx.close()
    if world > 1:
        # rank 0 works alone after the timed region; the others wait on the HOST (a key in the rendezvous store: an RCCL barrier
        # would spin on their GPUs, which the in-process leg uses; a gloo group would print to stdout)
        try:
            from torch.distributed import distributed_c10d as c10d
            store = c10d._get_default_store()
            if rank == 0:
                store.set("hry_bench_done", "1")
            else:
                store.wait(["hry_bench_done"])
        except Exception:
            dist.barrier()
        dist.destroy_process_group()


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and d.get(k) is not None}


def build_summary(line):
    """A digest of the line's sub-records, small enough (< 1.5 KB) to sit whole in the last 2 000 characters of the output: value,
    encode / decode times, the host loops and the roofline fraction of every larger configuration."""
    sm = {"n_gpus": line.get("n_gpus"), "value": line.get("value"), "enc_mtri_s": line.get("encode_mtri_s"), "dec_mtri_s": line.get("decode_mtri_s"),
          "frac": (line.get("roofline") or {}).get("frac"), "kernel": (line.get("roofline") or {}).get("kernel")}
    steps = line.get("step_ms_each")
    if steps:   # (the mean is `ms_per_step`; on shared hosts one step in ten or twenty is an outlier of the two host loops)
        sm["step_ms_median"] = round(float(np.median(steps)), 2)
        sm["value_at_median_step"] = round(float(line["value"]) * float(line["ms_per_step"]) / float(np.median(steps)), 2) if line.get("ms_per_step") else None
    st = line.get("stage_ms") or {}
    if st.get("dec_host_walk_ms") is not None:
        sm["host_walk_ms"] = st.get("host_walk_ms"); sm["host_replay_ms"] = st.get("dec_host_walk_ms"); sm["dec_ms"] = st.get("dec_total_ms"); sm["enc_ms"] = st.get("total_ms")
    if line.get("n_gpus", 1) > 1:
        sm.update(_pick(line, ("rccl_ranks", "n1_same_workload_value")))
        sm["cpu_ref"] = (line.get("cpu_baseline") or {}).get("value")
        sm["merged_ok"] = (line.get("sharded") or {}).get("merged_decode_equals_single_gpu")
        ip = line.get("inprocess")
        if ip:
            sm["inprocess"] = _pick(ip, ("value", "encode_ms", "decode_ms"))
    rf = lambda r: (r.get("roofline") or {}).get("frac")
    c = line.get("cfg3")
    if c:
        sm["cfg3"] = dict(_pick(c, ("value", "encode_from_host_ms", "decode_ms", "host_walk_ms", "host_replay_ms")), frac=rf(c), ok=c.get("round_trip_invariants_ok"))
    c = line.get("cfg4_share")
    if c:
        sm["cfg4_share"] = dict(_pick(c, ("value", "encode_ms", "decode_ms", "host_walk_ms", "host_replay_ms")), frac=rf(c), ok=c.get("round_trip_invariants_ok"))
    c = line.get("cfg4_end_to_end")
    if c:
        sm["cfg4_e2e"] = _pick(c, ("encode_mtri_s", "decode_mtri_s", "parse_ms"))
    c = line.get("cfg4_full")
    if c and "one_context" in c:
        o = c["one_context"]
        sm["cfg4_full"] = dict(_pick(o, ("value", "encode_ms", "encode_from_host_ms", "decode_ms", "host_walk_ms", "host_replay_ms")), tri=c.get("triangles"),
                               frac=rf(c), kernel=(c.get("roofline") or {}).get("kernel"), kernel_ms=(c.get("roofline") or {}).get("kernel_ms"),
                               traffic=(c.get("roofline") or {}).get("traffic"), ok=c.get("round_trip_invariants_ok"))
        e8 = c.get("eight_contexts_one_device")
        if e8:
            sm["cfg4_full"]["x8ctx"] = dict(_pick(e8, ("value", "encode_ms", "decode_ms")), faster=e8.get("encode_faster_than_one_context_from_host"))
        cc = c.get("compat")
        if cc:
            sm["cfg4_full"]["compat"] = _pick(cc, ("encode_ms", "encode_mtri_s", "decode_ms", "decode_mtri_s", "byte_identical_to_chunked_decode"))
    elif c:
        sm["cfg4_full"] = c
    c = line.get("compat")
    if c:
        sm["compat"] = _pick(c, ("encode_ms", "decode_ms", "encode_mtri_s", "decode_mtri_s", "byte_identical_to_cpu_ref"))
    c = line.get("obj")
    if c:
        sm["obj"] = {"enc_mtri_s": c.get("encode_mtri_s"), "dec_mtri_s": c.get("decode_mtri_s"), "chunked_enc": (c.get("chunked") or {}).get("encode_mtri_s"),
                     "chunked_dec": (c.get("chunked") or {}).get("decode_mtri_s"), "ok": bool(c.get("byte_identical_to_cpu_ref") and c.get("decode_equals_cpu_ref"))}
    cb = line.get("cpu_baseline")
    if cb and line.get("n_gpus", 1) == 1:
        sm["cpu_ref"] = _pick(cb, ("value", "kind", "cores"))
    return sm


def emit(line):
    """prints the ONE line; `summary` is its last key"""
    line.pop("summary", None)
    sm = build_summary(line)
    txt = json.dumps(sm, separators=(",", ":"))
    while len(txt) > 1500 and isinstance(sm, dict) and len(sm) > 4:   # never expected; drop the last sections rather than outgrow the tail
        sm.pop(next(reversed(sm)))
        txt = json.dumps(sm, separators=(",", ":"))
    line["summary"] = sm
    body = json.dumps({k: v for k, v in line.items() if k != "summary"})
    print(body[:-1] + ', "summary": ' + txt + "}")
    sys.stdout.flush()


def inprocess_leg(hc, whole, devices, quant, ntri, reps=3):
    """ONE process, one context per entry of `devices`: the whole job from the host mesh (hry_encode_sharded: plan, extract,
    upload, bounds, quantisation, encode, merge) and back (hry_decode_sharded: every segment on its device, one mesh out)"""
    mc = hc.MultiCodec(devices)
    try:
        enc, dec, te_l, td_l, merged = [], [], [], [], b""
        for _ in range(1 + reps):
            t0 = time.perf_counter()
            merged = mc.write_hry(whole, quant, keep_mesh=True, as_buffer=True)   # (the whole mesh stays as it is: every step computes the bounds again)
            t1 = time.perf_counter()
            te_l.append(dict(mc.last))
            mc.read_hry(merged)
            t2 = time.perf_counter()
            td_l.append(dict(mc.last))
            enc.append(t1 - t0); dec.append(t2 - t1)
        e, d = float(np.median(enc[1:])), float(np.median(dec[1:]))
        medk = lambda L, k: round(float(np.median([x[k] for x in L[1:]])), 2)
        return {"contexts": len(devices), "devices": sorted(set(int(x) for x in devices)), "value": round(ntri / (e + d) / 1e6, 3), "unit": "Mtriangles/s encode+decode",
                "encode_mtri_s": round(ntri / e / 1e6, 3), "decode_mtri_s": round(ntri / d / 1e6, 3), "encode_ms": round(e * 1e3, 2), "decode_ms": round(d * 1e3, 2),
                "hry_bytes": len(merged), "inputs_resident": False,
                "encode_stage_ms": {k: medk(te_l, k) for k in ("plan_ms", "extract_ms", "bounds_ms", "combine_ms", "quant_ms", "encode_ms", "merge_ms", "phase_a_ms", "phase_b_ms", "host_walk_ms", "total_ms")},
                "decode_stage_ms": {"directory_ms": medk(td_l, "plan_ms"), "decode_ms": medk(td_l, "encode_ms"), "place_ms": medk(td_l, "extract_ms"), "filler_ms": medk(td_l, "merge_ms"),
                                    "host_replay_ms": medk(td_l, "host_walk_ms"), "total_ms": medk(td_l, "total_ms")},
                "shards": int(te_l[-1]["n_shards"]), "components": int(te_l[-1]["n_components"]), "groups": int(te_l[-1]["n_groups"]),
                "what": "hry_encode_sharded + hry_decode_sharded from / to ONE host mesh: plan once, a worker thread per context, segments merged in host memory; no torch.distributed"}
    finally:
        mc.close()


def inprocess_main(args):
    """`python bench.py --gpus N` without a launcher: one process, N device contexts, no torch.distributed."""
    from harry_amd import _native as nat
    from harry_amd import codec as hc
    n_dev = nat.load().hry_device_count()
    if n_dev <= 0:
        raise SystemExit("bench.py needs a GPU: harry_amd has no CPU path")
    if n_dev < args.gpus and not args.share_device:
        raise SystemExit(f"bench.py: {args.gpus} contexts need {args.gpus} GPUs, this node shows {n_dev} (--share-device rehearses on fewer)")
    devices = [i % n_dev for i in range(args.gpus)]
    world = args.gpus
    quant = []
    mesh = build_cfg4(world, args.comps_per_gpu)
    whole = hc.Mesh.from_arrays(mesh.verts, mesh.degrees, mesh.indices)
    t0 = time.perf_counter()
    whole.twin()                                          # half-edge twins matched before the timed region (the reader's job; reported as twins_ms)
    twins_ms = (time.perf_counter() - t0) * 1e3
    mc = hc.MultiCodec(devices)
    ntri = mesh.ntri
    enc_s = dec_s = 0.0
    te_l, td_l, merged = [], [], b""

    def one_step():
        t0 = time.perf_counter()
        out = mc.write_hry(whole, quant, keep_mesh=True, as_buffer=True)   # (the whole mesh stays as it is: every step computes the bounds again)
        t1 = time.perf_counter()
        te = dict(mc.last)
        tim_e = mc.timings()
        dec = mc.read_hry(out)
        t2 = time.perf_counter()
        return out, dec, t1 - t0, t2 - t1, te, dict(mc.last), tim_e, mc.timings()

    for _ in range(args.warmup):
        one_step()
    t_begin = time.perf_counter()
    k_chain, k_entropy, k_dentropy = [], [], []
    dec = None
    per_ctx = []
    for _ in range(args.steps):
        merged, dec, te, td, a, b, tim_e, tim_d = one_step()
        enc_s += te; dec_s += td
        te_l.append(a); td_l.append(b)
        k_chain.append(max(t["k_chain_ms"] for t in tim_d)); k_entropy.append(max(t["k_entropy_ms"] for t in tim_e)); k_dentropy.append(max(t["k_entropy_ms"] for t in tim_d))
        per_ctx.append([(e["host_walk_ms"], e["k_entropy_ms"], d["host_walk_ms"], d["k_chain_ms"], d["k_entropy_ms"]) for e, d in zip(tim_e, tim_d)])
    t_all = time.perf_counter() - t_begin
    medk = lambda L, k: round(float(np.median([x[k] for x in L])), 2)
    value = ntri * args.steps / t_all / 1e6
    # dominant kernel: the measured maximum over the step's kernels (HIP events inside the library, the slowest context's); algorithmic
    # bytes of one context's share
    cands = {"k_unpredict2<float>": float(np.median(k_chain)), "k_chunk_encode": float(np.median(k_entropy)), "k_chunk_decode": float(np.median(k_dentropy))}
    dom_name = max(cands, key=cands.get)
    dom_ms = cands[dom_name]
    alg_bytes = (whole.nv * whole.list_stride(1) + 4 * whole.ne + len(merged)) // world
    roof = {"bound": "hbm", "kernel": dom_name, "achieved": round(alg_bytes / (dom_ms * 1e-3) / 1e9, 3) if dom_ms > 0 else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "traffic": None, "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms": round(dom_ms, 4),
            "note": "per context: its share of the mesh, its own launches (k_unpredict2<float>: summed over the batches of a decode; k_chunk_decode: both decoder kernels' launches)"}
    roof["frac"] = round(roof["achieved"] / HBM_PEAK_GBS, 6) if roof["achieved"] else None
    # check: the merged container decodes like the single-context path on the whole mesh
    ok = None
    try:
        cx = hc.Codec(devices[0])
        ref = cx.read_hry(cx.write_hry(whole.clone(), profile=hc.PROFILE_CHUNKED))
        ok = bool(np.array_equal(dec.face_offsets(), ref.face_offsets()) and np.array_equal(dec.org(), ref.org()) and np.array_equal(dec.twin(), ref.twin())
                  and np.array_equal(dec.list_data(1), ref.list_data(1)))
        cx.close()
    except Exception as exc:
        sys.stderr.write(f"merged-container check failed: {exc}\n")
        ok = False
    # every context's own numbers (medians over the steps): the record shows that N devices worked
    pc = np.median(np.array(per_ctx, dtype=np.float64), axis=0)
    contexts = [{"context": i, "device": int(devices[i]), "encode_host_walk_ms": round(float(pc[i][0]), 2), "k_chunk_encode_ms": round(float(pc[i][1]), 3),
                 "decode_host_replay_ms": round(float(pc[i][2]), 2), "k_unpredict2_float_ms": round(float(pc[i][3]), 3), "k_chunk_decode_ms": round(float(pc[i][4]), 3)} for i in range(world)]
    # the same mesh through ONE context on the first device: what the N contexts are to be compared with
    one_ctx = None
    try:
        cx1 = hc.Codec(devices[0])
        ts = []
        for _ in range(2):
            a = whole.clone()
            t0 = time.perf_counter(); ob = cx1.write_hry(a, profile=hc.PROFILE_CHUNKED, as_buffer=True); t1 = time.perf_counter(); cx1.read_hry(ob); ts.append((t1 - t0, time.perf_counter() - t1))
        cx1.close()
        one_ctx = {"value": round(ntri / sum(ts[-1]) / 1e6, 3), "encode_from_host_mtri_s": round(ntri / ts[-1][0] / 1e6, 3), "decode_mtri_s": round(ntri / ts[-1][1] / 1e6, 3)}
    except Exception as exc:
        sys.stderr.write(f"one-context leg failed: {exc}\n")
    line = {"metric": "Mtriangles/s encode+decode", "value": round(value, 4), "unit": "Mtriangles/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(t_all / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "mode": "in-process (--inprocess; NOT the default for N > 1): ONE process, one worker thread + device context per GPU (hry_encode_sharded / hry_decode_sharded); no torch.distributed, no collective -- rccl_ranks 0",
            "twins_ms": round(twins_ms, 1), "twins_note": "half-edge twin matching of the freshly built mesh, on the host, once, before the steps (a reader's job)",
            "contexts": contexts, "one_context_same_mesh": one_ctx, "rccl_ranks": 0,
            "n1_same_workload_value": one_ctx["value"] if one_ctx else None,
            "n1_same_workload_note": "the SAME mesh through ONE context on the first device, from the host mesh, encode + decode: read this line's scaling against it",
            "efficiency_vs_one_context": round(value / (world * one_ctx["value"]), 4) if one_ctx else None,
            "speedup_vs_one_context": round(value / one_ctx["value"], 3) if one_ctx else None,
            "cpus_allowed": cpu_allowance(),
            "dtype": "u8 residual bytes of f32 values, u32 range-coder registers", "data": "synthetic",
            "config": {"workload": f"ONE mesh shaped like BASELINE configs[3] ({ntri} triangles, {int(te_l[-1]['n_components'])} components); per GPU: {args.comps_per_gpu} mixed-polygon "
                                   f"components with non-manifold edges / vertices, {ntri // world} triangles, float32 xyz, lossless; sharded by connected component",
                       "profile": "chunked", "decode_in_step": True, "parallelism": f"component-sharded x{world}, ONE process, one worker thread + context per device",
                       "inputs_resident": False, "step_includes": "plan, extract, upload, bounds, encode, merge; decode of every segment, placement into one mesh",
                       "devices": devices, "shared_devices": len(set(devices)) < len(devices)},
            "encode_mtri_s": round(ntri * args.steps / enc_s / 1e6, 4), "decode_mtri_s": round(ntri * args.steps / dec_s / 1e6, 4),
            "hry_bytes": len(merged), "bits_per_vertex": round(8 * len(merged) / max(whole.nv, 1), 4),
            "stage_ms": {"encode": {k: medk(te_l, k) for k in ("plan_ms", "extract_ms", "bounds_ms", "combine_ms", "quant_ms", "encode_ms", "merge_ms", "phase_a_ms", "phase_b_ms", "host_walk_ms", "total_ms")},
                         "decode": {"directory_ms": medk(td_l, "plan_ms"), "decode_ms": medk(td_l, "encode_ms"), "place_ms": medk(td_l, "extract_ms"), "filler_ms": medk(td_l, "merge_ms"),
                                    "host_replay_ms": medk(td_l, "host_walk_ms"), "total_ms": medk(td_l, "total_ms")}},
            "kernel_ms": {k: round(v, 4) for k, v in cands.items()},
            "roofline": roof,
            "sharded": {"executor": "in-process (hry_encode_sharded / hry_decode_sharded)", "contexts": world, "segments": int(td_l[-1]["n_segments"]),
                        "groups": int(te_l[-1]["n_groups"]), "merged_decode_equals_single_gpu": ok, "collectives": "none: the segments meet in host memory"}}
    if ok is False:
        line["error"] = "merged container does not decode to the single-GPU result"
    if not args.no_cpu_baseline:
        cb = cpu_baseline_cfg4_sample(min(12, args.comps_per_gpu * world))
        if cb is not None:
            line["cpu_baseline"] = cb
    emit(line)
    mc.close()


def cfg3_leg(cx):
    """BASELINE configs[2] stand-in at full size: closed torus 3742 x 3742 = 28 005 128 triangles with analytic normals, positions
    14 bits / normals 10 bits, chunked profile, one GPU.  Encode is timed from the HOST mesh (upload inside)."""
    from harry_amd import codec as hc
    from harry_amd import meshgen as mg
    mesh = mg.torus(3742, 3742, seed=3, sigma=1e-4, normals=True)
    m0 = hc.Mesh.from_arrays(mesh.verts, mesh.degrees, mesh.indices)
    m0.twin()
    quant = [(1, 0, 14), (1, 1, 14), (1, 2, 14), (1, 3, 10), (1, 4, 10), (1, 5, 10)]
    enc, dec, tms = [], [], []
    out, qrec, d = b"", None, None
    for _ in range(3):
        m = m0.clone()
        t0 = time.perf_counter()
        cx.requant(m, quant)
        out = cx.write_hry(m, profile=hc.PROFILE_CHUNKED, as_buffer=True)
        t1 = time.perf_counter()
        te = cx.timing()
        d = cx.read_hry(out)
        t2 = time.perf_counter()
        td = cx.timing()
        enc.append(t1 - t0); dec.append(t2 - t1); tms.append((te, td))
        qrec = m
    # size-independent check (the decoder renumbers vertices in coding order): the same multiset of quantised records
    a = np.ascontiguousarray(qrec.list_data(1).reshape(m0.nv, -1).view(np.uint16)[:, ::2])
    b = np.ascontiguousarray(d.list_data(1).reshape(m0.nv, -1).view(np.uint16)[:, ::2])
    key = lambda x: np.sort((x[:, 0].astype(np.uint64) << 50) | (x[:, 1].astype(np.uint64) << 36) | (x[:, 2].astype(np.uint64) << 22) ^ (x[:, 3].astype(np.uint64) << 20)
                            ^ (x[:, 4].astype(np.uint64) << 10) ^ x[:, 5].astype(np.uint64))
    ok = bool((d.nv, d.nf, d.ne) == (m0.nv, m0.nf, m0.ne) and np.array_equal(key(a), key(b)))
    e, dd = min(enc[1:]), min(dec[1:])
    te, td = tms[enc.index(e)][0], tms[dec.index(dd)][1]   # (the stage times of the passes the record quotes)
    ntri = mesh.ntri
    alg = m0.nv * m0.list_stride(1) + 4 * m0.ne + len(out)
    return {"workload": "closed torus 3742 x 3742, 28 005 128 triangles, float32 xyz + analytic normals, -l1 -a0 -q14 -a1 -q14 -a2 -q14 -a3 -q10 -a4 -q10 -a5 -q10 (BASELINE configs[2] stand-in)",
            "triangles": int(ntri), "value": round(ntri / (e + dd) / 1e6, 3), "encode_from_host_mtri_s": round(ntri / e / 1e6, 3), "decode_mtri_s": round(ntri / dd / 1e6, 3),
            "encode_from_host_ms": round(e * 1e3, 2), "decode_ms": round(dd * 1e3, 2), "hry_bytes": len(out), "bits_per_vertex": round(8 * len(out) / m0.nv, 3),
            "host_walk_ms": round(te["host_walk_ms"], 2), "host_replay_ms": round(td["host_walk_ms"], 2), "k_unpredict3_ms": round(td["k_chain_ms"], 2),
            "k_chunk_encode_ms": round(te["k_entropy_ms"], 2), "k_chunk_decode_ms": round(td["k_entropy_ms"], 2), "k_predict_ms": round(te["k_predict_ms"], 2),
            "roofline": {"bound": "hbm", "kernel": "k_unpredict3", "algorithmic_bytes_per_launch": alg, "kernel_ms": round(td["k_chain_ms"], 3),
                         "achieved": round(alg / (td["k_chain_ms"] * 1e-3) / 1e9, 3) if td["k_chain_ms"] > 0 else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(alg / (td["k_chain_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 6) if td["k_chain_ms"] > 0 else None},
            "round_trip_invariants_ok": ok, "passes": 2}


def cfg4_share_leg(cx):
    """One GPU's share of BASELINE configs[3]: 128 mixed-polygon components, 0.1 % non-manifold edges, lossless float32 -- the
    lossless-float reconstruction chain (k_unpredict2<float>), all components in one launch."""
    from harry_amd import codec as hc
    mesh = build_cfg4(1)
    m0 = hc.Mesh.from_arrays(mesh.verts, mesh.degrees, mesh.indices)
    m0.twin()
    enc, enc_res, dec, tms = [], [], [], []
    out, d = b"", None
    for _ in range(3):
        m = m0.clone()
        t0 = time.perf_counter()
        out = cx.write_hry(m, profile=hc.PROFILE_CHUNKED, as_buffer=True)        # from the host mesh: the upload is inside
        t1 = time.perf_counter()
        te = cx.timing()
        d = cx.read_hry(out)
        t2 = time.perf_counter()
        td = cx.timing()
        m = m0.clone(); cx.upload(m)
        t3 = time.perf_counter()
        cx.write_hry(m, profile=hc.PROFILE_CHUNKED, as_buffer=True)              # resident inputs
        t4 = time.perf_counter()
        enc.append(t1 - t0); dec.append(t2 - t1); enc_res.append(t4 - t3); tms.append((te, td))
    # size-independent check: the decoded vertex records are a permutation of the input's (lossless)
    rec = lambda mm: np.sort(np.ascontiguousarray(mm.list_data(1)).view(np.dtype((np.void, 12))).reshape(-1))
    nref = int(np.unique(m0.org()).size)                          # vertices no face references are not coded
    ok = bool((d.nv, d.nf, d.ne) == (m0.nv, m0.nf, m0.ne) and (nref < m0.nv or np.array_equal(rec(d), rec(m0))))
    e, er, dd = min(enc[1:]), min(enc_res[1:]), min(dec[1:])
    te, td = tms[enc.index(e)][0], tms[dec.index(dd)][1]   # (the stage times of the passes the record quotes)
    ntri = mesh.ntri
    alg = m0.nv * m0.list_stride(1) + 4 * m0.ne + len(out)
    traffic = traffic_raw = traffic_source = None
    for rnd in ("r6", "r5", "r4", "r3"):   # the PMC passes of this very workload (scripts/collect_profiles.sh: leg cfg4share)
        if traffic is not None:
            break
        try:
            with open(os.path.join(ROOT, "profiles", rnd, "cfg4share", "traffic.json")) as f:
                hit = json.load(f)["kernels"]["k_unpredict2<float>"]
            traffic = 2 * hit["fetch_bytes"] + hit["write_bytes"]
            traffic_raw = {"FETCH_SIZE_bytes": hit["fetch_bytes"], "WRITE_SIZE_bytes": hit["write_bytes"]}
            traffic_source = f"profiles/{rnd}/cfg4share/traffic.json (rocprofv3 --pmc, separate passes; FETCH_SIZE x2 per the gfx950 note; the three chains of a component write 4 bytes each into the same 12-byte records)"
        except (OSError, KeyError, ValueError):
            pass
    return {"workload": "128 mixed-polygon components (40 % quads, 5 % pentagons) with 0.1 % non-manifold edges and 0.05 % non-manifold vertices, float32 xyz, lossless: one GPU's share of BASELINE configs[3]",
            "triangles": int(ntri), "value": round(ntri / (er + dd) / 1e6, 3), "encode_mtri_s": round(ntri / er / 1e6, 3), "encode_from_host_mtri_s": round(ntri / e / 1e6, 3),
            "decode_mtri_s": round(ntri / dd / 1e6, 3), "encode_ms": round(er * 1e3, 2), "encode_from_host_ms": round(e * 1e3, 2), "decode_ms": round(dd * 1e3, 2),
            "hry_bytes": len(out), "bits_per_vertex": round(8 * len(out) / m0.nv, 3),
            "host_walk_ms": round(te["host_walk_ms"], 2), "host_replay_ms": round(td["host_walk_ms"], 2), "k_unpredict2_float_ms": round(td["k_chain_ms"], 2),
            "k_chunk_encode_ms": round(te["k_entropy_ms"], 2), "k_chunk_decode_ms": round(td["k_entropy_ms"], 2), "k_predict_ms": round(te["k_predict_ms"], 2),
            "roofline": {"bound": "hbm", "kernel": "k_unpredict2<float>", "algorithmic_bytes_per_launch": alg, "algorithmic_bytes_per_triangle": round(alg / ntri, 2),
                         "kernel_ms": round(td["k_chain_ms"], 3), "achieved": round(alg / (td["k_chain_ms"] * 1e-3) / 1e9, 3) if td["k_chain_ms"] > 0 else None,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(alg / (td["k_chain_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 6) if td["k_chain_ms"] > 0 else None,
                         "traffic": traffic, "traffic_raw": traffic_raw, "traffic_source": traffic_source},
            "round_trip_invariants_ok": ok, "passes": 2}


def cfg4_end_to_end_leg(cx):
    """One GPU's share of BASELINE configs[3] from file bytes to file bytes: a binary PLY with polygons of several degrees -> mesh
    (parse on the host threads, twin matching on the device) -> .hry (chunked) -> mesh -> binary PLY.  What `harry in.ply out.hry`
    and `harry out.hry back.ply` do between reading and writing their files (the 100 M-triangle mesh through the command itself:
    profiles/r4/cfg4_e2e_100M.txt)."""
    from harry_amd import codec as hc
    mesh = build_cfg4(1)
    ply = mesh.to_ply()
    parse, enc, dec, wr = [], [], [], []
    out, back = b"", b""
    for _ in range(3):
        t0 = time.perf_counter()
        m = hc.Mesh.from_ply(ply)
        t1 = time.perf_counter()
        out = cx.write_hry(m, profile=hc.PROFILE_CHUNKED, as_buffer=True)
        t2 = time.perf_counter()
        d = cx.read_hry(out)
        t3 = time.perf_counter()
        back = d.to_ply()
        t4 = time.perf_counter()
        parse.append(t1 - t0); enc.append(t2 - t1); dec.append(t3 - t2); wr.append(t4 - t3)
    ntri = mesh.ntri
    # size-independent check: the PLY that comes back holds the same multiset of vertex records and as many faces of every degree
    m0 = hc.Mesh.from_ply(ply)
    mb = hc.Mesh.from_ply(back)
    rec = lambda mm: np.sort(np.ascontiguousarray(mm.list_data(1)).view(np.dtype((np.void, 12))).reshape(-1))
    nref = int(np.unique(m0.org()).size)
    ok = bool((mb.nv, mb.nf, mb.ne) == (m0.nv, m0.nf, m0.ne) and np.array_equal(np.sort(np.diff(mb.face_offsets())), np.sort(np.diff(m0.face_offsets())))
              and (nref < m0.nv or np.array_equal(rec(mb), rec(m0))))
    p, e, dd, w = min(parse[1:]), min(enc[1:]), min(dec[1:]), min(wr[1:])
    return {"workload": "one GPU's share of BASELINE configs[3] (128 mixed-polygon components, 0.1 % non-manifold edges, float32 xyz, lossless), file bytes to file bytes",
            "triangles": int(ntri), "ply_bytes": len(ply), "hry_bytes": len(out),
            "encode_mtri_s": round(ntri / (p + e) / 1e6, 3), "decode_mtri_s": round(ntri / (dd + w) / 1e6, 3),
            "parse_ms": round(p * 1e3, 2), "parse_ms_per_mtri": round(p * 1e3 / (ntri / 1e6), 3), "encode_ms": round(e * 1e3, 2), "decode_ms": round(dd * 1e3, 2),
            "ply_write_ms": round(w * 1e3, 2), "encode_includes": "twin matching on the device, upload, walk, kernels, container", "round_trip_invariants_ok": ok, "passes": 2}


def cfg4_full_leg(cx):
    """BASELINE configs[3] / [4] at the named size on ONE MI355X: 1 024 mixed-polygon components + 150 000 non-manifold slivers,
    100.6 M triangles, float32 xyz, lossless.  One context (resident inputs, and from the host mesh), then eight contexts on this
    one device through the in-process executor (hry_encode_sharded / hry_decode_sharded: what `harry --gpus 8` runs) -- the code
    path of an 8-GPU node with one GPU's worth of hardware.  Size-independent checks only (the oracle needs two minutes:
    tests/test_gpu_configs.py holds that check)."""
    import resource
    from harry_amd import codec as hc
    try:
        avail_gb = os.sysconf("SC_AVPHYS_PAGES") * os.sysconf("SC_PAGE_SIZE") / 2**30
    except (ValueError, OSError):
        avail_gb = 0
    if avail_gb < 48:
        return {"skipped": f"needs about 40 GB of host memory, {avail_gb:.0f} GB are free"}
    t0 = time.perf_counter()
    mesh = build_cfg4(8)
    m0 = hc.Mesh.from_arrays(mesh.verts, mesh.degrees, mesh.indices)
    cx.upload(m0)                                         # (twin matching on the device; every clone below has its twins)
    build_s = time.perf_counter() - t0
    ntri = mesh.ntri
    enc, dec, host, tms = [], [], [], []
    out, d = b"", None
    for _ in range(2):
        m = m0.clone(); cx.upload(m)
        t0 = time.perf_counter()
        out = cx.write_hry(m, profile=hc.PROFILE_CHUNKED, as_buffer=True)
        t1 = time.perf_counter()
        te = cx.timing()
        d = cx.read_hry(out)
        t2 = time.perf_counter()
        td = cx.timing()
        m = m0.clone()
        t3 = time.perf_counter()
        cx.write_hry(m, profile=hc.PROFILE_CHUNKED, as_buffer=True)   # from the host mesh: the upload inside
        t4 = time.perf_counter()
        enc.append(t1 - t0); dec.append(t2 - t1); host.append(t4 - t3); tms.append((te, td))
    rec = lambda mm: np.sort(np.ascontiguousarray(mm.list_data(1)).view(np.dtype((np.void, 12))).reshape(-1))
    ok = bool((d.nv, d.nf, d.ne) == (m0.nv, m0.nf, m0.ne) and np.array_equal(np.sort(np.diff(d.face_offsets())), np.sort(np.diff(m0.face_offsets()))))
    nref = int(np.unique(m0.org()).size)
    ok = ok and (nref < m0.nv or bool(np.array_equal(rec(d), rec(m0))))
    one = d
    e, dd, eh = enc[-1], dec[-1], host[-1]
    te, td = tms[-1]
    alg = m0.nv * m0.list_stride(1) + 4 * m0.ne + len(out)
    kc = td["k_chain_ms"]
    # the dominant kernel is the MEASURED maximum (HIP events inside the library): k_chunk_encode of the encode, the float chains of
    # the decode summed over their batches (ChainBatches: three beside the replay + the last), the decode's entropy kernels
    cands = {"k_chunk_model+k_chunk_ranges": te["k_entropy_ms"], "k_unpredict2<float>": kc, "k_chunk_decode_lanes": td["k_entropy_ms"]}
    dom = max(cands, key=cands.get)
    dom_ms = cands[dom]
    traffic = traffic_raw = traffic_source = None
    for rnd in ("r6", "r5"):   # the PMC passes of this very workload (scripts/collect_profiles.sh PART=2: leg cfg4full)
        try:
            with open(os.path.join(ROOT, "profiles", rnd, "cfg4full", "traffic.json")) as f:
                hit = json.load(f)["kernels"][dom]
            traffic = 2 * hit["fetch_bytes"] + hit["write_bytes"]
            traffic_raw = {"FETCH_SIZE_bytes": hit["fetch_bytes"], "WRITE_SIZE_bytes": hit["write_bytes"]}
            traffic_source = f"profiles/{rnd}/cfg4full/traffic.json (rocprofv3 --pmc, separate passes, per pass of the workload; FETCH_SIZE x2 per the gfx950 note)"
            break
        except (OSError, KeyError, ValueError):
            pass
    res = {"workload": "BASELINE configs[3] / [4] at the named size: 1 024 mixed-polygon components (40 % quads, 5 % pentagons) + 150 000 non-manifold slivers, float32 xyz, lossless, ONE MI355X",
           "triangles": int(ntri), "components": None, "build_s": round(build_s, 1), "cpus_allowed": cpu_allowance(),
           "one_context": {"value": round(ntri / (e + dd) / 1e6, 3), "encode_mtri_s": round(ntri / e / 1e6, 3), "decode_mtri_s": round(ntri / dd / 1e6, 3),
                           "encode_from_host_mtri_s": round(ntri / eh / 1e6, 3), "encode_ms": round(e * 1e3, 1), "encode_from_host_ms": round(eh * 1e3, 1), "decode_ms": round(dd * 1e3, 1),
                           "host_walk_ms": round(te["host_walk_ms"], 1), "host_replay_ms": round(td["host_walk_ms"], 1), "k_unpredict2_float_ms": round(kc, 2),
                           "k_chunk_encode_ms": round(te["k_entropy_ms"], 2), "k_chunk_decode_ms": round(td["k_entropy_ms"], 2), "hry_bytes": len(out),
                           "bits_per_vertex": round(8 * len(out) / m0.nv, 3)},
           "roofline": {"bound": "hbm", "kernel": dom, "algorithmic_bytes_per_launch": alg, "kernel_ms": round(dom_ms, 3),
                        "kernel_ms_candidates": {k: round(v, 3) for k, v in cands.items()},
                        "kernel_ms_note": "k_unpredict2<float>: sum over the batches of one decode; k_chunk_decode_lanes: the decode's entropy kernels (lanes + a few wave-per-stream launches); "
                                          "k_chunk_model+k_chunk_ranges: the encoder's two kernels together (round 5: the model a wavefront per stream, the range registers a lane per stream; k_chunk_encode was 28 ms)",
                        "achieved": round(alg / (dom_ms * 1e-3) / 1e9, 3) if dom_ms > 0 else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(alg / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 6) if dom_ms > 0 else None,
                        "traffic": traffic, "traffic_raw": traffic_raw, "traffic_source": traffic_source},
           "round_trip_invariants_ok": ok, "passes": 2}
    # the reference's single stream (compat, .hry v0.1: the drop-in default) of the same mesh: encode from resident inputs, decode
    try:
        m = m0.clone(); cx.upload(m)
        t0 = time.perf_counter()
        cdat = cx.write_hry(m, profile=hc.PROFILE_COMPAT, as_buffer=True)
        t1 = time.perf_counter()
        tce = cx.timing()
        cdec = cx.read_hry(cdat)
        t2 = time.perf_counter()
        same = bool(np.array_equal(cdec.org(), one.org()) and np.array_equal(cdec.list_data(1), one.list_data(1)))
        res["compat"] = {"encode_ms": round((t1 - t0) * 1e3, 1), "encode_mtri_s": round(ntri / (t1 - t0) / 1e6, 3), "decode_ms": round((t2 - t1) * 1e3, 1),
                         "decode_mtri_s": round(ntri / (t2 - t1) / 1e6, 3), "hry_bytes": len(cdat), "host_walk_ms": round(tce["host_walk_ms"], 1),
                         "recurrence_ms": round(tce["k_rchain_ms"], 1), "byte_identical_to_chunked_decode": same, "passes": 1,
                         "what": ".hry v0.1 (bit-identical to the CPU reference: tests/test_gpu_configs.py holds the oracle check at this size); one serial range recurrence on a host core"}
        del cdec, cdat
    except Exception as exc:
        sys.stderr.write(f"cfg4_full: compat leg failed: {exc}\n")
    # eight contexts on this one device: plan, walk in place, interval uploads, one container -- and back
    try:
        mc = hc.MultiCodec([cx.device] * 8)
        try:
            se, sd, last_e, last_d, merged, md = [], [], None, None, b"", None
            for _ in range(2):
                m = m0.clone()
                t0 = time.perf_counter()
                merged = mc.write_hry(m, as_buffer=True)
                t1 = time.perf_counter()
                last_e = dict(mc.last)
                md = mc.read_hry(merged)
                t2 = time.perf_counter()
                last_d = dict(mc.last)
                se.append(t1 - t0); sd.append(t2 - t1)
            same = bool(np.array_equal(md.org(), one.org()) and np.array_equal(md.list_data(1), one.list_data(1)) and np.array_equal(md.face_offsets(), one.face_offsets()))
            res["eight_contexts_one_device"] = {"value": round(ntri / (se[-1] + sd[-1]) / 1e6, 3), "encode_mtri_s": round(ntri / se[-1] / 1e6, 3), "decode_mtri_s": round(ntri / sd[-1] / 1e6, 3),
                                                "encode_ms": round(se[-1] * 1e3, 1), "decode_ms": round(sd[-1] * 1e3, 1), "hry_bytes": len(merged),
                                                "encode_stage_ms": {k: round(last_e[k], 1) for k in ("twins_ms", "plan_ms", "extract_ms", "bounds_ms", "encode_ms", "merge_ms", "host_walk_ms", "total_ms")},
                                                "decode_stage_ms": {"decode_ms": round(last_d["encode_ms"], 1), "place_ms": round(last_d["extract_ms"], 1), "host_replay_ms": round(last_d["host_walk_ms"], 1), "total_ms": round(last_d["total_ms"], 1)},
                                                "encode_faster_than_one_context_from_host": bool(se[-1] < eh), "decodes_to_the_one_context_mesh": same,
                                                "what": "hry_encode_sharded / hry_decode_sharded, from / to ONE host mesh (inputs not resident); extract_ms = the interval uploads beside the walks"}
        finally:
            mc.close()
    except Exception as exc:
        sys.stderr.write(f"cfg4_full: in-process leg failed: {exc}\n")
    res["peak_rss_gb"] = round(resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 2**20, 1)
    return res


def _obj_scene_record(cx, sc, workload, with_reference_binary=True):
    """One OBJ scene through both profiles: reference stream (bytes against the CPU port) and the parallel container (bytes against the
    port's restatement, decode against the reference-format decode); best of the passes after the first."""
    import subprocess
    import tempfile
    from harry_amd import codec as hc
    from oracle import oracle_py as op   # checker only
    t0 = time.perf_counter()
    m = hc.Mesh.from_obj(sc.obj, "")
    t_parse = time.perf_counter() - t0
    ntri = m.ntri
    enc, dec, data = [], [], b""
    for _ in range(3):
        a = m.clone(); cx.upload(a)                      # records, connectivity and binding tables resident in HBM, as for the PLY legs
        t0 = time.perf_counter()
        data = cx.write_hry(a, profile=hc.PROFILE_COMPAT)
        t1 = time.perf_counter()
        d = cx.read_hry(data)
        t2 = time.perf_counter()
        enc.append(t1 - t0); dec.append(t2 - t1)
    want = op.Mesh.from_obj(sc.obj, "").encode().data
    # the same scene in the parallel container (.hry v0.2 holds general bindings too)
    cenc, cdec, cdata, ctm_e, ctm_d = [], [], b"", {}, {}
    for _ in range(3):
        a = m.clone(); cx.upload(a)
        t0 = time.perf_counter()
        cdata = cx.write_hry(a, profile=hc.PROFILE_CHUNKED)
        t1 = time.perf_counter()
        ctm_e = cx.timing()
        t1b = time.perf_counter()
        cd = cx.read_hry(cdata)
        t2 = time.perf_counter()
        ctm_d = cx.timing()
        cenc.append(t1 - t0); cdec.append(t2 - t1b)
    chunked = {"encode_ms": round(min(cenc[1:]) * 1e3, 2), "decode_ms": round(min(cdec[1:]) * 1e3, 2),
               "encode_mtri_s": round(ntri / min(cenc[1:]) / 1e6, 3), "decode_mtri_s": round(ntri / min(cdec[1:]) / 1e6, 3), "hry_bytes": len(cdata),
               "encode_host_ms": round(ctm_e.get("host_walk_ms", 0.0), 2), "decode_chain_ms": round(ctm_d.get("k_chain_ms", 0.0), 2),
               "container_equals_cpu_port": bool(cdata == op.Mesh.from_obj(sc.obj, "").encode_chunked(hc.container_info(cdata)["chunk_syms"]).data),
               "decode_equals_reference_format_decode": bool(all(np.array_equal(cd.list_data(l), x) for l, x in enumerate(_oracle_lists(op, want))))}
    rec = {"workload": workload, "triangles": int(ntri), "obj_bytes": len(sc.obj), "inputs_resident": True, "chunked": chunked,
           "parse_ms": round(t_parse * 1e3, 2), "encode_ms": round(min(enc[1:]) * 1e3, 2), "decode_ms": round(min(dec[1:]) * 1e3, 2),
           "encode_mtri_s": round(ntri / min(enc[1:]) / 1e6, 3), "decode_mtri_s": round(ntri / min(dec[1:]) / 1e6, 3),
           "hry_bytes": len(data), "byte_identical_to_cpu_ref": bool(data == want),
           "decode_equals_cpu_ref": bool(d.to_obj() is not None and all(np.array_equal(d.list_data(l), x) for l, x in enumerate(_oracle_lists(op, want))))}
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "harry_ref")
    if with_reference_binary and os.path.exists(ref_bin):
        with tempfile.TemporaryDirectory() as tmp:
            src, hry, back = os.path.join(tmp, "s.obj"), os.path.join(tmp, "s.hry"), os.path.join(tmp, "b.obj")
            with open(src, "wb") as f:
                f.write(sc.obj)
            t0 = time.perf_counter()
            subprocess.run([ref_bin, src, hry], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
            t1 = time.perf_counter()
            subprocess.run([ref_bin, hry, back], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
            t2 = time.perf_counter()
            rec["reference_binary"] = {"obj_to_hry_ms": round((t1 - t0) * 1e3, 1), "hry_to_obj_ms": round((t2 - t1) * 1e3, 1),
                                       "bytes_equal": bool(open(hry, "rb").read() == data), "what": "whole process, file to file, one core"}
    return rec


def obj_leg(cx):
    """OBJ scenes: smooth normals + atlas (shared records, short chains), one normal per face (every record
    depends on the one before it: k_gen_chain<1, float>), and a many-part scene over two device contexts of this process."""
    from harry_amd import codec as hc
    from harry_amd import meshgen as mg
    from harry_amd import objgen as og
    from oracle import oracle_py as op   # checker only
    rec = _obj_scene_record(cx, og.scene(mg.torus(200, 200, seed=2), normals="smooth", tex="atlas", charts=7),
                            "torus 200 x 200 as OBJ: smooth normals + 7-chart texture atlas (v / vt / vn, f v/t/n)")
    try:
        rec["flat_normals"] = _obj_scene_record(cx, og.scene(mg.torus(200, 200, seed=2), normals="flat", tex="atlas", charts=7),
                                                "the same torus with ONE normal per face (f v/t/n, 80 000 vn lines): the serial record chain")
    except Exception as exc:   # noqa: BLE001
        rec["flat_normals"] = {"error": str(exc)}
    try:
        # a scene of many parts in the sharded container (.hry v0.3 with general bindings): two contexts of this process on this device
        base = mg.with_nonmanifold(mg.multi_component(32, 40, 41, seed=5, polys="mixed"), 30, 20, seed=4)
        sc = og.scene(base, normals="flat", tex="corner")
        whole = hc.Mesh.from_obj(sc.obj, "")
        one = cx.read_hry(cx.write_hry(whole.clone(), profile=hc.PROFILE_CHUNKED))
        mc = hc.MultiCodec([cx.device, cx.device])
        try:
            te, td, merged = [], [], b""
            for _ in range(3):
                w = hc.Mesh.from_obj(sc.obj, "")
                t0 = time.perf_counter()
                merged = mc.write_hry(w, [])
                t1 = time.perf_counter()
                got = mc.read_hry(merged)
                t2 = time.perf_counter()
                te.append(t1 - t0); td.append(t2 - t1)
            info = hc.container_info(merged)
            rec["sharded"] = {"workload": "32 mixed-polygon parts with non-manifold edges as OBJ (one normal per face, texture coordinates per corner), 2 contexts on one device",
                              "triangles": int(whole.ntri), "segments": int(info.get("segments", 0)), "hry_bytes": len(merged),
                              "encode_ms": round(min(te[1:]) * 1e3, 2), "decode_ms": round(min(td[1:]) * 1e3, 2),
                              "encode_mtri_s": round(whole.ntri / min(te[1:]) / 1e6, 3), "decode_mtri_s": round(whole.ntri / min(td[1:]) / 1e6, 3),
                              "text_equals_single_context": bool(got.to_obj() == one.to_obj())}
        finally:
            mc.close()
    except Exception as exc:   # noqa: BLE001
        rec["sharded"] = {"error": str(exc)}
    return rec


def _oracle_lists(op, data):
    o = op.Mesh.from_hry(data)
    return [o.list_data(l) for l in range(o.nlists)]


if __name__ == "__main__":
    main()
