"""Multi-GPU sharding of the .hry path: meshes shard by independent connected component; there is
no data-path collective.  Ranks exchange only the finished streams, for concatenation on rank 0 (RCCL over xGMI when the
process group is `nccl`, gloo in the CPU tests)."""
from __future__ import annotations

import struct

import torch
import torch.distributed as dist


def assign_components(tri_counts, world: int):
    """Greedy bin packing by triangle count (largest first, ties by index); identical on every rank."""
    order = sorted(range(len(tri_counts)), key=lambda c: (-tri_counts[c], c))
    parts = [[] for _ in range(world)]
    load = [0] * world
    for c in order:
        r = min(range(world), key=lambda k: (load[k], k))
        parts[r].append(c)
        load[r] += tri_counts[c]
    return [sorted(p) for p in parts]


def gather_streams(streams: dict, n_components: int, device: torch.device):
    """streams: {component index: bytes} produced by this rank.  Returns {component: bytes} on rank 0, None elsewhere.
    Variable-length gather: all_gather of the per-component sizes, then one padded gather of the payloads."""
    world, rank = dist.get_world_size(), dist.get_rank()
    sizes = torch.zeros(n_components, dtype=torch.int64, device=device)
    for c, b in streams.items():
        sizes[c] = len(b)
    all_sizes = [torch.zeros_like(sizes) for _ in range(world)]
    dist.all_gather(all_sizes, sizes)
    per_rank = [int(s.sum().item()) for s in all_sizes]
    cap = max(max(per_rank), 1)
    buf = torch.zeros(cap, dtype=torch.uint8, device=device)
    off = 0
    for c in sorted(streams):
        b = streams[c]
        buf[off:off + len(b)] = torch.frombuffer(bytearray(b), dtype=torch.uint8).to(device)
        off += len(b)
    out = [torch.zeros(cap, dtype=torch.uint8, device=device) for _ in range(world)] if rank == 0 else None
    dist.gather(buf, out, dst=0)
    if rank != 0:
        return None
    result = {}
    for r in range(world):
        raw = out[r].cpu().numpy().tobytes()
        off = 0
        for c in range(n_components):
            n = int(all_sizes[r][c].item())
            if n:
                result[c] = raw[off:off + n]
                off += n
    return result


MAGIC = b"HRYS"


def concat_container(streams: dict) -> bytes:
    """Concatenation of per-component .hry streams: magic, u32 count, u64 sizes, payloads in component order."""
    keys = sorted(streams)
    head = MAGIC + struct.pack("<I", len(keys)) + b"".join(struct.pack("<Q", len(streams[k])) for k in keys)
    return head + b"".join(streams[k] for k in keys)


def split_container(blob: bytes):
    if blob[:4] != MAGIC:
        raise ValueError("not a multi-component container")
    (n,) = struct.unpack_from("<I", blob, 4)
    sizes = struct.unpack_from(f"<{n}Q", blob, 8)
    off = 8 + 8 * n
    out = []
    for s in sizes:
        out.append(blob[off:off + s])
        off += s
    return out
