"""Development: the hand-over times of every 64-vertex tile of the reconstruction chain (component 0) of the headline mesh,
from a library built with -DHRY_CHAIN_LOG (scripts/build_variant.sh chainlog -DHRY_CHAIN_LOG; run with
HRY_LIB=harry_amd/variants/libharry_amd_chainlog.so python scripts/chain_log.py [side]): ticks per tile by the tile's kind, along
the chain, by the late tiles' overlap.  FAST_TAIL=1: the slow tail of the tiles without heads (percentiles, by their place in a row of
such tiles, by what their owner did before, the late ones' indices mod 64 -- how the 64-tile flush and the preparation rate were
found); ROWHEADS=1: the candidate counts of the heads evaluated from rows; MARKS=1 with a -DHRY_CHAIN_MARKS build: ticks from the
value's arrival to the end of the first run, of the first head, of the tile."""
import sys, os, ctypes
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
bench.stay_on_memory_node()
from harry_amd import codec as hc
from harry_amd import meshgen as mg
side = int(sys.argv[1]) if len(sys.argv) > 1 else 708
g = mg.torus(side, side, seed=2, sigma=1e-4)
cx = hc.Codec(0)
m = hc.Mesh.from_arrays(g.verts, g.degrees, g.indices)
cx.requant(m, [(1, -1, 14)])
cx.upload(m)
data = cx.write_hry(m, profile=hc.PROFILE_CHUNKED)
for it in range(4):
    d = cx.read_hry(data); del d
lib = ctypes.CDLL(os.environ["HRY_LIB"])
nv = side * side
nt = (nv + 63) // 64
buf = np.zeros(nt, dtype=np.uint64)
assert lib.hry_debug_chain_log(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint(nt)) == 0
t = (buf >> np.uint64(16)).astype(np.int64)
kind = (buf & np.uint64(0xf)).astype(np.int64)
nh = ((buf >> np.uint64(4)) & np.uint64(0x3f)).astype(np.int64)
nrow = ((buf >> np.uint64(10)) & np.uint64(0x3f)).astype(np.int64)
dt = np.diff(t)
k1 = kind[1:]
print(f"tiles {nt}; ticks first -> last {t[-1] - t[0]} (with the gaps between launches)")
gap = (dt > 20000) | (dt < 0)          # between launches (pieces)
print(f"launch gaps: {gap.sum()} of {dt[gap].sum()} ticks; inside launches {dt[~gap].sum()} ticks, {dt[~gap].mean():.0f} per tile")
names = {0: "unset", 1: "fast", 2: "prepared, heads / out of range", 3: "prepared late", 4: "dense", 5: "prepared late, fast", 6: "one head between two runs"}
for k in sorted(set(k1.tolist())):
    sel = (k1 == k) & ~gap
    if sel.any():
        print(f"  {names.get(k, k):32s} n {sel.sum():6d}  mean {dt[sel].mean():8.0f}  median {np.median(dt[sel]):8.0f}  p90 {np.percentile(dt[sel], 90):8.0f}  sum {dt[sel].sum():9d} ({100.0 * dt[sel].sum() / dt[~gap].sum():.1f} %)")
# fast tiles by what came before them
for prev in (1, 2):
    sel = (k1[1:] == 1) & (k1[:-1] == prev) & ~gap[1:]
    if sel.any(): print(f"  fast after kind {prev}: n {sel.sum()} mean {dt[1:][sel].mean():.0f}")
sel2 = (k1 == 2) & ~gap
for h in range(0, 9):
    s = sel2 & (nh[1:] == h)
    if s.any(): print(f"  prepared with {h} heads ({nrow[1:][s].mean():.2f} from rows): n {s.sum():6d} mean {dt[s].mean():8.0f}")
# along the chain: per 1/16 of the mesh
step = max(1, (nt - 1) // 16)
for i in range(0, nt - 1, step):
    s = slice(i, min(nt - 1, i + step))
    g_ = ~gap[s]
    kk = k1[s]
    print(f"  tiles {i:6d}..: {dt[s][g_].mean():7.0f} per tile; fast {np.mean(kk == 1):.2f} heads {np.mean(kk == 2):.2f} late {np.mean(kk == 3):.2f} late-fast {np.mean(kk == 5):.2f} dense {np.mean(kk == 4):.2f}")
sel3 = ((k1 == 3) | (k1 == 5)) & ~gap
for d in range(0, 8):
    s = sel3 & (nrow[1:] == d)
    if s.any(): print(f"  prepared late, {d} tiles of overlap: n {s.sum():6d} median {np.median(dt[s]):8.0f}")

if hasattr(lib, "hry_debug_chain_marks") and os.environ.get("MARKS"):
    mk = np.zeros(nt, dtype=np.uint64)
    assert lib.hry_debug_chain_marks(mk.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint(nt)) == 0
    a = (mk & np.uint64(0xffff)).astype(np.int64); b = ((mk >> np.uint64(16)) & np.uint64(0xffff)).astype(np.int64); c = ((mk >> np.uint64(32)) & np.uint64(0xffff)).astype(np.int64)
    print("ticks from the arrival of the value to: the end of the first run | of the first head | of the tile (medians)")
    for k in (1, 2, 3, 5):
        for h in range(0, 4):
            sel = (kind == k) & ((nh == h) if k in (2, 3) else True) & (c > 0)
            if sel.any() and (h == 0 or k in (2, 3)):
                print(f"  kind {k} heads {h if k in (2, 3) else '-'}: n {sel.sum():6d}  run {np.median(a[sel]):6.0f}  head {np.median(b[sel]):6.0f}  end {np.median(c[sel]):6.0f}   (tile-to-tile median {np.median(dt[sel[1:] & ~gap]) if (sel[1:] & ~gap).any() else 0:.0f})")

sel = (kind == 1)
if sel.any() and nh[sel].max() > 0:
    print("fast tiles by the chain's polls for the descriptor:", {int(k): int((nh[sel] == k).sum()) for k in sorted(set(nh[sel].tolist()))})

if hasattr(lib, "hry_debug_chain_marks") and os.environ.get("ROWHEADS"):
    mk = np.zeros(nt, dtype=np.uint64)
    assert lib.hry_debug_chain_marks(mk.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint(nt)) == 0
    from collections import Counter
    c = Counter()
    for w in mk[mk != 0].tolist():
        while w:
            c[w & 0xff] += 1; w >>= 8
    print("heads evaluated from candidate rows, by their number of candidates (accumulated over the decodes of this run):", dict(sorted(c.items())))

if os.environ.get("FAST_TAIL"):
    W = int(os.environ.get("HRY_CHAIN_WAVES", "7"))
    f = (k1 == 1) & ~gap
    d = dt[f]
    print("fast tiles: ticks to the tile before, percentiles 10/25/50/75/90/95/99:", [int(np.percentile(d, p)) for p in (10, 25, 50, 75, 90, 95, 99)])
    idx = np.nonzero(f)[0] + 1
    # by the length of the run of fast tiles that ends here
    runlen = np.zeros(nt, dtype=np.int64)
    for i in range(1, nt):
        runlen[i] = runlen[i - 1] + 1 if kind[i] == 1 else 0
    for r in (1, 2, 3, 4, 5, 6, 7, 8, 10, 14, 20):
        s = f & (runlen[1:] == r)
        if s.any(): print(f"  {r:3d}th fast tile in a row: n {s.sum():5d} mean {dt[s].mean():7.0f} median {np.median(dt[s]):7.0f}")
    s = f & (runlen[1:] > 20)
    if s.any(): print(f"  beyond the 20th: n {s.sum():5d} mean {dt[s].mean():7.0f} median {np.median(dt[s]):7.0f}")
    # by what the same wavefront did W tiles earlier (it prepared this tile right after that one)
    for kk in (1, 2, 6, 3):
        s = f.copy(); s[:W] = False
        s[W:] &= (kind[1:][:-W] == kk) if W < nt - 1 else False
        if s.any(): print(f"  fast tiles whose owner's previous tile was of kind {kk}: n {s.sum():5d} mean {dt[s].mean():7.0f} median {np.median(dt[s]):7.0f}")
    big = np.nonzero((dt > 4000) & ~gap & (k1 == 1))[0] + 1
    from collections import Counter
    print("fast tiles above 4000 ticks:", len(big), "their index mod 64:", dict(sorted(Counter((big % 64).tolist()).items())))
