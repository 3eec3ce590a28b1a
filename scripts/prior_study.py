"""Size study (development aid): ideal adaptive code length of the vertex / operation planes of the 1 M-triangle workload for
different chunk sizes and initial tables (flat = the reference's all-ones, prior = the plane's own histogram scaled to K).
The adaptive code length depends on the per-chunk symbol counts only:
  bits = sum_j log2(T0 + j) - sum_s sum_{k < n_s} log2(a_s + k)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from scipy.special import gammaln
from harry_amd import codec as hc, meshgen as mg

def bits(counts, init):
    counts = counts.astype(np.float64); init = init.astype(np.float64)
    n, t0 = counts.sum(), init.sum()
    m = (counts > 0)
    num = (gammaln(init[m] + counts[m]) - gammaln(init[m])).sum()
    den = gammaln(t0 + n) - gammaln(t0)
    return (den - num) / np.log(2)

def study(name, plane, chunks, Ks):
    N = len(plane)
    hist = np.bincount(plane, minlength=256)
    present = int((hist > 0).sum())
    out = []
    for ch in chunks:
        row = {}
        for K in Ks:
            if K == 0:
                init = (hist > 0).astype(np.int64) if name.startswith("op") else np.ones(256, np.int64)
                extra = 0
            else:
                init = np.where(hist > 0, np.maximum(1, np.round(K * hist / N)), 0).astype(np.int64)
                extra = 16 * present   # bits to store the table (2 bytes per present symbol, generous)
            tot = extra
            for f in range(0, N, ch):
                c = np.bincount(plane[f:f + ch], minlength=256)
                tot += bits(c, init) + 32 + 32   # flush + directory entry
            row[K] = tot / 8
        out.append((ch, row))
    return out

mesh = mg.torus(708, 708, seed=2, sigma=1e-4)
m0 = hc.Mesh.from_arrays(mesh.verts, mesh.degrees, mesh.indices)
cx = hc.Codec(0)
cx.requant(m0, [(1, -1, 14)])
m = m0.clone()
cx.write_hry(m, profile=hc.PROFILE_CHUNKED, keep_stages=True)
vp = cx.stage("vplanes").reshape(6, -1)
w = m0.clone().host_walk(plain=True)
ops = [w["op_sym"][w["op_class"] == k] for k in range(8)]
chunks = [8192, 16384, 32768, 131072]
Ks = [0, 256, 1024, 4096]
total = {(ch, K): 0.0 for ch in chunks for K in Ks}
for p in range(6):
    for ch, row in study(f"v{p}", vp[p], chunks, Ks):
        for K, b in row.items(): total[(ch, K)] += b
for k in range(8):
    if len(ops[k]):
        for ch, row in study(f"op{k}", ops[k], [max(512, c // 8) for c in chunks], Ks):
            for K, b in row.items(): total[(ch * 8, K)] += b
print("bytes (vertex planes + operation planes), rows = chunk size, columns = K of the prior (0 = flat)")
for ch in chunks:
    print(ch, {K: int(total[(ch, K)]) for K in Ks})
