"""Random quantised meshes through the chunked profile on the GPU against the CPU oracle, aimed at the reconstruction chain:
heavy noise and few bits (clamped parallelograms, far residual codes: the scan's speculation fails and is repaired), short rings
(every vertex a head: the cut-and-scan path), non-manifold and multi-component meshes, forced small slices of the pipelined decode.
Development aid; the committed tests hold the fixed cases.   python tests/tools/chain_stress.py [n_meshes] [seed]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from harry_amd import codec as hc
from harry_amd import meshgen as mg
from oracle import oracle_py as op   # checker only

n_meshes = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
cx = hc.Codec(0)


def same(a, o):
    assert (a.nv, a.nf, a.ne) == (o.nv, o.nf, o.ne)
    assert np.array_equal(a.org(), o.org())
    for l in (0, 1):
        assert np.array_equal(a.list_data(l), o.list_data(l)), l


t0 = time.time()
only = os.environ.get("STRESS_ONLY")
failed = 0
for it in range(n_meshes):
    kind = int(rng.integers(0, 7 if os.environ.get("STRESS_BIG") else 6))
    if os.environ.get("STRESS_SLIVERS") and rng.integers(0, 2):
        kind = 7   # many components with slivers coded before and after the components they hang on (work lists of several tiny components, owners remembered by the chains)
    sigma = float(10.0 ** rng.uniform(-5, -0.5))
    seed = int(rng.integers(1, 99))
    polys = ["tri", "quad", "mixed"][int(rng.integers(0, 3))]
    dims = [int(x) for x in rng.integers(0, 1 << 30, 4)]
    q = int(rng.integers(2, 17))
    lossless = bool(rng.integers(0, 4) == 0) or bool(os.environ.get("STRESS_LOSSLESS"))
    scale_pow = float(rng.uniform(-43, 37)) if rng.integers(0, 3) == 0 else 0.0   # lossless: magnitudes where float sums overflow or go denormal
    flat_axis = int(rng.integers(0, 6))                                            # lossless: 0 - 2: that coordinate is 0 everywhere (every prediction is 0)
    from_compat = bool(rng.integers(0, 4) == 0)     # decode the reference-format stream (host entropy decoder + device reconstruction)
    chunk = [0, 0, 1024, 4096, 20000][int(rng.integers(0, 5))]
    mode = int(rng.integers(0, 3))
    faces, slice_ = int(rng.integers(64, 3000)), 64 * int(rng.integers(1, 200))
    if only is not None and it != int(only):
        continue
    r = lambda k, lo, hi: lo + dims[k] % (hi - lo)
    if kind == 0:
        base = mg.torus(r(0, 20, 260), r(1, 20, 260), polys=polys, seed=seed, sigma=sigma)
    elif kind == 1:
        base = mg.grid(r(0, 20, 250), r(1, 20, 250), seed=seed, sigma=sigma, quads=bool(dims[2] & 1))
    elif kind == 2:
        base = mg.icosphere(r(0, 3, 7), seed=seed, sigma=sigma)
    elif kind == 3:
        base = mg.multi_component(r(0, 2, 30), r(1, 8, 60), r(2, 8, 60), polys=polys, seed=seed)
    elif kind == 4:
        base = mg.with_nonmanifold(mg.torus(r(0, 30, 120), r(1, 30, 120), polys=polys, seed=seed, sigma=sigma), r(2, 1, 40), r(3, 1, 20), seed=seed)
    elif kind == 5:
        base = mg.with_colors(mg.torus(r(0, 30, 160), r(1, 30, 160), normals=True, seed=seed, sigma=sigma))
    elif kind == 7:
        base = mg.with_nonmanifold(mg.multi_component(r(0, 2, 40), r(1, 6, 40), r(2, 6, 40), polys=polys, seed=seed), r(2, 20, 400), r(3, 10, 200), seed=seed)
    else:   # large enough for the pipelined decode with its production parameters (STRESS_BIG=1)
        base = mg.torus(r(0, 370, 520), r(1, 370, 520), polys="tri", seed=seed, sigma=sigma) if dims[2] & 1 else mg.grid(r(0, 370, 520), r(1, 370, 520), seed=seed, sigma=sigma)
        lossless, mode = False, 2
    if lossless and kind != 5:
        v = base.verts.copy()
        with np.errstate(over="ignore", under="ignore"):
            if scale_pow:
                for k in "xyz":
                    v[k] = (v[k].astype(np.float64) * 10.0 ** scale_pow).astype(np.float32)
            if flat_axis < 3:
                v["xyz"[flat_axis]][:] = np.float32(0.0) if seed & 1 else np.float32(-0.0)
        v2 = np.nan_to_num(np.stack([v["x"], v["y"], v["z"]]), nan=0.0, posinf=3e38, neginf=-3e38)
        v["x"], v["y"], v["z"] = v2[0], v2[1], v2[2]
        base = mg.Mesh(v, base.degrees, base.indices, base.face_props)
    ply = base.to_ply()
    a, o = hc.Mesh.from_ply(ply), op.Mesh.from_ply(ply)
    quant = [] if lossless else [(1, -1, q)] if kind != 5 else [(1, c, q) for c in range(6)]   # (the colours are bytes already)
    if quant:
        cx.requant(a, quant)
        o.requant(quant)
    compat = o.clone().encode().data
    ref_dec = op.Mesh.from_hry(compat)
    got = cx.write_hry(a.clone(), profile=hc.PROFILE_CHUNKED, chunk_syms=chunk)
    assert got == o.clone().encode_chunked(chunk).data, "container"
    if from_compat:
        assert cx.write_hry(a.clone(), profile=hc.PROFILE_COMPAT) == compat, "reference-format stream"
        got = compat
    for k in ("HRY_NO_PIPELINE", "HRY_PIPELINE_MIN_VERTICES", "HRY_PIPELINE_FACES", "HRY_PIPELINE_SLICE"):
        os.environ.pop(k, None)
    if mode == 0:
        os.environ["HRY_NO_PIPELINE"] = "1"
    elif mode == 1:
        os.environ.update(HRY_PIPELINE_MIN_VERTICES="0", HRY_PIPELINE_FACES=str(faces), HRY_PIPELINE_SLICE=str(slice_))
    def diff(d):
        msg = []
        if not np.array_equal(d.org(), ref_dec.org()): msg.append("org")
        for l in (0, 1):
            x, y = d.list_data(l), ref_dec.list_data(l)
            if not np.array_equal(x, y):
                st = d.list_stride(l)
                neq = x.reshape(-1, st) != y.reshape(-1, st)
                bad = np.flatnonzero(neq.any(axis=1))
                msg.append(f"list{l}: {len(bad)} records, first {bad[:6].tolist()} last {bad[-3:].tolist()} byte columns {np.flatnonzero(neq.any(axis=0)).tolist()}")
        return "; ".join(msg)
    verdict = "ok"
    again = int(os.environ.get("STRESS_REPEAT", "1"))
    stages = []
    for rep in range(again):
        dmsg = diff(cx.read_hry(got, keep_stages=bool(os.environ.get("STRESS_STAGES"))))
        if os.environ.get("STRESS_STAGES"):
            stages.append((bool(dmsg), cx.stage("ncand").copy(), cx.stage("cand", np.uint32).copy()))
        if dmsg:
            failed += 1
            verdict = "MISMATCH"
            print(f"    decode {rep}: {dmsg}", flush=True)
    if stages and any(b for b, _, _ in stages) and not all(b for b, _, _ in stages):
        bad = next(x for x in stages if x[0]); good = next(x for x in stages if not x[0])
        nc_b, nc_g = bad[1][:a.nv], good[1][:a.nv]
        cd_b, cd_g = bad[2].reshape(-1, 6)[:a.nv], good[2].reshape(-1, 6)[:a.nv]
        dn = np.flatnonzero(nc_b != nc_g)
        print(f"    candidate counts differ at {len(dn)} vertices: {dn[:10].tolist()}; bad {nc_b[dn[:10]].tolist()} good {nc_g[dn[:10]].tolist()}")
        for v in dn[:4]:
            print(f"      v {v}: bad rows {cd_b[v][:3 * max(1, min(2, int(nc_b[v])))].tolist()}  good rows {cd_g[v][:3 * max(1, min(2, int(nc_g[v])))].tolist()}")
    src = "compat" if from_compat else "chunk%d" % chunk
    print(f"{it:3d} kind {kind} {polys:5s} nv {a.nv:6d} q{0 if lossless else q:<2d} {src} sigma {sigma:.1e} seed {seed} mode {mode} faces {faces} slice {slice_}  {verdict}  ({time.time() - t0:.0f} s)", flush=True)
print("all equal" if not failed else f"{failed} MISMATCHES")
sys.exit(1 if failed else 0)
