"""How the reconstruction chain of the 1 M-triangle workload decomposes into tiles / runs (development aid):
candidate tables of a decode -> per-vertex 'need' and 'gap' as k_chain_records computes them -> run and early-tile counts."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from harry_amd import codec as hc, meshgen as mg
n = int(sys.argv[1]) if len(sys.argv) > 1 else 708
W = int(sys.argv[2]) if len(sys.argv) > 2 else 4
mesh = mg.torus(n, n, seed=2, sigma=1e-4)
m0 = hc.Mesh.from_arrays(mesh.verts, mesh.degrees, mesh.indices)
cx = hc.Codec(0)
cx.requant(m0, [(1, -1, 14)])
out = cx.write_hry(m0.clone(), profile=hc.PROFILE_CHUNKED)
cx.read_hry(out, keep_stages=True)
nc = cx.stage("ncand").astype(np.int64)
cand = cx.stage("cand", np.uint32).reshape(-1, 24).astype(np.int64)
nv = len(nc)
v = np.arange(nv)
ids = cand[:, :6].copy()
valid = (np.arange(6)[None, :] < (3 * np.minimum(nc, 2))[:, None])
plus = (np.arange(6) % 3 != 2)[None, :]
is_pred = valid & plus & (ids == (v - 1)[:, None])
first_pred = is_pred & (np.cumsum(is_pred, axis=1) == 1)
other = valid & ~first_pred
need = np.where(other, ids + 1, 0).max(axis=1)
gap = np.where(need > 0, v + 1 - need, 65535)
big = nc > 2
tile = v // 64
lane = v % 64
need_rel = np.maximum(0, need - tile * 64)
runs = 0; early = 0; tiles = (nv + 63) // 64; bigs = int(big.sum())
for t in range(tiles):
    lo, hi = t * 64, min(nv, t * 64 + 64)
    s = lo
    first = True
    while s < hi:
        if big[s]:
            s += 1; first = False; continue
        e = s + 1
        while e < hi and not big[e] and need_rel[e] <= s - lo:
            e += 1
        runs += 1
        if first and np.all(gap[s:e] > lane[s:e] + 64 * (W - 1)):
            early += 1
        first = False
        s = e
print(f"{nv} vertices, {tiles} tiles, {runs} runs ({runs/tiles:.2f} per tile), {bigs} vertices with > 2 candidates, early tiles {early} ({100*early/tiles:.1f} %) at W = {W}")
print("gap histogram (vertices):", {k: int(((gap >= a) & (gap < b)).sum()) for k, (a, b) in {"1-2": (1, 3), "3-63": (3, 64), "64-255": (64, 256), "256-1023": (256, 1024), ">=1024": (1024, 1 << 30)}.items()})
