"""What the reconstruction chain (k_unpredict3) meets on the headline mesh: per 64-vertex tile, how many runs start inside it
(a vertex needs a source of its own tile other than its predecessor), how many vertices have more than two candidates, how many
tiles can be prepared before their predecessor is finished.  From the candidate table of a decode (stages "cand" / "ncand")."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from harry_amd import codec as hc, meshgen as mg
n = int(sys.argv[1]) if len(sys.argv) > 1 else 708
mesh = mg.torus(n, n, seed=2, sigma=1e-4)
m = hc.Mesh.from_arrays(mesh.verts, mesh.degrees, mesh.indices)
cx = hc.Codec(0)
cx.requant(m, [(1, -1, 14)])
out = cx.write_hry(m, profile=hc.PROFILE_CHUNKED)
d = cx.read_hry(out, keep_stages=True)
nc = cx.stage("ncand").astype(np.int64)[:d.nv]
cand = cx.stage("cand", np.uint32).reshape(-1, 24)[:d.nv].astype(np.int64)
nv = d.nv
v = np.arange(nv)
print("ncand histogram", np.bincount(np.minimum(nc, 9), minlength=10).tolist())
big = nc > 2
# sources of vertices with <= 2 candidates
ids = cand[:, :6].copy()
valid = np.zeros((nv, 6), bool)
valid[:, :3] = (nc >= 1)[:, None] & ~big[:, None]
valid[:, 3:] = (nc >= 2)[:, None] & ~big[:, None]
is_pred = valid & (ids + 1 == v[:, None]) & (np.arange(6)[None, :] % 3 != 2)
# the chained source: the first predecessor slot
first_pred = np.where(is_pred.any(1), is_pred.argmax(1), 7)
chained = np.zeros((nv, 6), bool)
rows = np.where(first_pred < 7)[0]
chained[rows, first_pred[rows]] = True
other = valid & ~chained
need = np.where(other, ids + 1, 0).max(1)           # every other source must be final before the run starts
tile = v & ~63
need_rel = np.maximum(need - tile, 0)
gap = np.where(need > 0, v + 1 - need, 1 << 30)
ntiles = (nv + 63) // 64
cuts = bigs = unsettled = 0
runs_hist = np.zeros(8, np.int64)
for t in range(ntiles):
    lo, hi = 64 * t, min(nv, 64 * t + 64)
    s = 0
    r = 0
    nr, bg = need_rel[lo:hi], big[lo:hi]
    while s < hi - lo:
        if bg[s]:
            bigs += 1; s += 1; r += 1; continue
        j = s + 1
        while j < hi - lo and not bg[j] and nr[j] <= s:
            j += 1
        r += 1
        s = j
    cuts += r - 1
    runs_hist[min(r, 7)] += 1
    if (gap[lo:hi] <= np.arange(hi - lo) + 64 * 3).any():
        unsettled += 1
print(f"tiles {ntiles}; serial pieces per tile beyond the first: {cuts / ntiles:.3f}; vertices with > 2 candidates per tile: {bigs / ntiles:.3f}; "
      f"tiles with a recent source (not preparable early with 4 wavefronts): {unsettled / ntiles:.3f}")
print("pieces per tile histogram (1..7+):", runs_hist[1:].tolist())
dist = v - np.where(other, ids, v[:, None]).min(1)
near = other.any(1) & (np.where(other, v[:, None] - ids, 1 << 30).min(1) < 64)
print("vertices with a non-predecessor source less than 64 back:", int(near.sum()), f"({near.mean():.4f})")
print("distance of the nearest non-predecessor source, percentiles 1/5/50:", np.percentile(np.where(other, v[:, None] - ids, 1 << 30).min(1)[other.any(1)], [1, 5, 50]).tolist())
