"""Dependency pattern of the reconstruction chain inside 64-vertex batches (development aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from harry_amd import codec as hc, meshgen as mg

kind = sys.argv[1] if len(sys.argv) > 1 else "torus"
if kind == "torus":
    mesh = mg.torus(708, 708, seed=2, sigma=1e-4)
elif kind == "ico":
    mesh = mg.icosphere(7)
elif kind == "mixed":
    mesh = mg.torus(500, 500, polys="mixed", seed=3)
else:
    mesh = mg.grid(700, 700)
m0 = hc.Mesh.from_arrays(mesh.verts, mesh.degrees, mesh.indices)
cx = hc.Codec(0)
cx.requant(m0, [(1, -1, 14)])
out = cx.write_hry(m0, profile=hc.PROFILE_CHUNKED)
cx.read_hry(out, keep_stages=True)
nc = cx.stage("ncand").astype(np.int64)
cand = cx.stage("cand", np.uint32).reshape(-1, 24).astype(np.int64)
n = len(nc)
v = np.arange(n)
base = v & ~63
print("vertices", n, "ncand hist", {int(k): int(c) for k, c in zip(*np.unique(nc, return_counts=True))})
ids = cand[:, :6].copy()
valid = np.zeros((n, 6), bool)
valid[:, :3] = (nc >= 1)[:, None]
valid[:, 3:] = (nc >= 2)[:, None]
pend = valid & (ids >= base[:, None])
dist = v[:, None] - ids
npend = pend.sum(1)
print("pending sources per vertex:", {int(k): int(c) for k, c in zip(*np.unique(npend, return_counts=True))})
for j, name in enumerate(["a0", "b0", "o0", "a1", "b1", "o1"]):
    pj = pend[:, j]
    d = dist[pj, j]
    print(f"slot {name}: pending {pj.sum():8d}  dist==1 {int((d == 1).sum()):8d}  dist==2 {int((d == 2).sum()):8d}  other {int((d > 2).sum()):8d}")
# pure chain: <= 1 pending, role a or b, distance 1
one = npend == 1
plus_role = (pend[:, [0, 1, 3, 4]].sum(1) == 1)
d1 = ((pend & (dist == 1)).sum(1) == 1)
pure = (npend == 0) | (one & plus_role & d1)
print("pure-chain vertices:", int(pure.sum()), f"({pure.mean()*100:.2f} %)")
two_same = (npend == 2) & ((pend & (dist == 1)).sum(1) == 2)
print("two pending, both the previous vertex:", int(two_same.sum()))
small = (nc <= 2)
bad = ~(pure) | ~small
nb = (n + 63) // 64
badb = np.zeros(nb, bool)
np.logical_or.at(badb, v[bad] // 64, True)
print("batches:", nb, "with a non-pure vertex:", int(badb.sum()), f"({badb.mean()*100:.1f} %)")
