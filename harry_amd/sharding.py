"""One mesh over N GPUs: the mesh shards by groups of connected components (libharry_amd:
hry_shard_plan / hry_shard_extract), every rank codes its shard with no data-path collective, and the ranks exchange only
  * the per-component bounds of their shards (a few hundred bytes, all-gather) -- quantisation and the container header use
    the bounds of the WHOLE mesh (structs/quant.h:30-96), and
  * the finished segments, gathered on rank 0 for concatenation into ONE .hry v0.3 container (hry_merge).
torch.distributed is plumbing here: backend `nccl` = RCCL over xGMI on the GPU box, `gloo` in the CPU tests."""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

from . import codec as hc


def assign_components(tri_counts, world: int):
    """Greedy bin packing by triangle count (largest first, ties by index); identical on every rank.  The library applies
    the same rule to groups of components when there are few of them (host/shard.cpp)."""
    order = sorted(range(len(tri_counts)), key=lambda c: (-tri_counts[c], c))
    parts = [[] for _ in range(world)]
    load = [0] * world
    for c in order:
        r = min(range(world), key=lambda k: (load[k], k))
        parts[r].append(c)
        load[r] += tri_counts[c]
    return [sorted(p) for p in parts]


# ---- bounds of the whole mesh from the bounds of its shards ------------------------------------------------------------
def shard_bounds(cx: "hc.Codec", shard: "hc.Mesh") -> np.ndarray:
    """k_bounds on the shard's resident records -> int64[n_components_total, 4]: min bits, max bits, and for each the key that
    breaks ties like ONE sequential scan over the whole mesh would (structs/quant.h:30-44: strict comparisons, first wins):
    0 for the scan's initial value, else 1 + the element's index in the whole mesh."""
    cx.bounds(shard)
    rows = []
    for l in range(shard.nlists):
        fmt = shard.list_fmt(l)
        if not fmt:
            continue
        mn, mx = shard.list_min(l), shard.list_max(l)
        at = shard.bounds_at(l)
        whole_index = shard.shard_elements(16 + l) if shard.general else shard.shard_elements(1 if l == 1 else 0)
        for c, (t, _q, off) in enumerate(fmt):
            sz = hc.TYPE_SIZE[t]
            bits = lambda rec: int.from_bytes(bytes(rec[off:off + sz]), "little")
            key = lambda a: 0 if a == 0 else (int(whole_index[a - 1]) + 1 if len(whole_index) else a)
            rows.append([bits(mn), bits(mx), key(at[c][0]), key(at[c][1])])
    return np.array(rows, dtype=np.uint64).reshape(-1, 4).view(np.int64)


def combine_bounds(per_shard, shard: "hc.Mesh"):
    """per_shard: the shard_bounds arrays of every shard (same list formats).  Sets the whole mesh's bounds on `shard`."""
    tabs = [np.asarray(p).view(np.uint64).reshape(-1, 4) for p in per_shard]
    row = 0
    for l in range(shard.nlists):
        fmt = shard.list_fmt(l)
        if not fmt:
            continue
        stride = shard.list_stride(l)
        mn, mx = bytearray(stride), bytearray(stride)
        for t, _q, off in fmt:
            sz, dt = hc.TYPE_SIZE[t], np.dtype(hc.TYPE_NP[t])
            val = lambda b: np.frombuffer(int(b).to_bytes(8, "little")[:sz], dtype=dt)[0]
            best_mn = best_mx = None
            for tab in tabs:
                vmn, vmx, kmn, kmx = val(tab[row, 0]), val(tab[row, 1]), int(tab[row, 2]), int(tab[row, 3])
                if best_mn is None or vmn < best_mn[0] or (not best_mn[0] < vmn and kmn < best_mn[1]):
                    best_mn = (vmn, kmn, int(tab[row, 0]))
                if best_mx is None or vmx > best_mx[0] or (not best_mx[0] > vmx and kmx < best_mx[1]):
                    best_mx = (vmx, kmx, int(tab[row, 1]))
            mn[off:off + sz] = best_mn[2].to_bytes(8, "little")[:sz]
            mx[off:off + sz] = best_mx[2].to_bytes(8, "little")[:sz]
            row += 1
        shard.set_bounds(l, bytes(mn), bytes(mx))


def exchange_bounds(cx: "hc.Codec", shard: "hc.Mesh", device: torch.device):
    """k_bounds on this rank's shard, all-gather (RCCL over xGMI under `nccl`), the same combination on every rank"""
    allgather_combine(shard_bounds(cx, shard), shard, device)


def allgather_combine(table: np.ndarray, shard: "hc.Mesh", device: torch.device):
    """table: this rank's shard_bounds array"""
    mine = torch.from_numpy(np.ascontiguousarray(table).copy()).to(device)
    if not dist.is_initialized() or dist.get_world_size() == 1:
        combine_bounds([mine.cpu().numpy()], shard)
        return
    out = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    combine_bounds([o.cpu().numpy() for o in out], shard)


# ---- the finished segments -> rank 0 -------------------------------------------------------------------------------
_host_stage = {}   # rank 0: pinned landing memory of the gathered segments, kept between steps (pinning costs milliseconds per MB)


def _as_u8_tensor(container):
    """the container's bytes as a CPU uint8 tensor WITHOUT a copy (bytes are read-only: torch warns, nothing writes)"""
    import warnings
    src = container.view() if hasattr(container, "view") and not isinstance(container, (bytes, bytearray, memoryview)) else container
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return torch.frombuffer(src, dtype=torch.uint8) if len(container) else torch.zeros(0, dtype=torch.uint8)


class SegmentGather:
    """Variable-length gather of every rank's one-segment container on rank 0: all_gather of the sizes, then one padded gather
    of the payloads, started asynchronously so that it overlaps whatever the rank does next (the collective runs on RCCL's
    own stream); finish() waits for it and, on rank 0, merges the segments into ONE .hry v0.3 (hry_merge).  No copy of a segment
    on the way: the rank's container goes up from where the encoder wrote it, rank 0 lands all of them in one pinned block and
    hry_merge reads them there (eight 39 MB segments through bytes objects were 100 ms of rank 0's step)."""

    def __init__(self, container, device: torch.device, as_buffer: bool = False):
        self.single = not dist.is_initialized() or dist.get_world_size() == 1
        self.container = container
        self.as_buffer = as_buffer
        if self.single:
            return
        world, self.rank = dist.get_world_size(), dist.get_rank()
        size = torch.tensor([len(container)], dtype=torch.int64, device=device)
        sizes = [torch.zeros_like(size) for _ in range(world)]
        dist.all_gather(sizes, size)
        self.sizes = [int(s.item()) for s in sizes]
        cap = max(max(self.sizes), 1)
        self.buf = torch.empty(cap, dtype=torch.uint8, device=device)
        self.buf[:len(container)].copy_(_as_u8_tensor(container), non_blocking=False)
        self.out = [torch.empty(cap, dtype=torch.uint8, device=device) for _ in range(world)] if self.rank == 0 else None
        # under RCCL the gather is asynchronous (its own stream, device buffers); gloo would run it on a host thread next to the
        # rank's own host work (the replay of the decode is host-bound), so there it completes right here
        self.work = dist.gather(self.buf, self.out, dst=0, async_op=dist.get_backend() == "nccl")

    def finish(self):
        """rank 0: the merged container; other ranks: None"""
        if self.single:
            return hc.merge([self.container], as_buffer=self.as_buffer)
        if self.work is not None:
            self.work.wait()
        if self.rank != 0:
            return None
        total = sum(self.sizes)
        on_gpu = self.out[0].is_cuda
        key = "pinned" if on_gpu else "plain"
        stage = _host_stage.get(key)
        if stage is None or stage.numel() < total:
            stage = torch.empty(total + total // 8 + 4096, dtype=torch.uint8, pin_memory=on_gpu)
            _host_stage[key] = stage
        parts, at = [], 0
        for r, n in enumerate(self.sizes):
            stage[at:at + n].copy_(self.out[r][:n], non_blocking=on_gpu)
            parts.append(stage[at:at + n].numpy())
            at += n
        if on_gpu:
            torch.cuda.current_stream().synchronize()
        return hc.merge(parts, as_buffer=self.as_buffer)


def merge_on_rank0(container: bytes, device: torch.device):
    """rank 0: ONE .hry v0.3 container holding every rank's segment; other ranks: None"""
    return SegmentGather(container, device).finish()
