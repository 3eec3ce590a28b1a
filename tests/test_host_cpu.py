"""CPU-side tests of the product's host code (no GPU, no compute entry points):
 - the C-ABI library loads and exports every symbol include/harry_amd.h declares
 - PLY reader / half-edge twins / PLY writer agree with the oracle
 - the host cut-border walk (order, repaired twins, connectivity symbols, op model) agrees with the oracle's trace
"""
import ctypes
import os
import re

import numpy as np
import pytest

from harry_amd import _native as nat
from harry_amd import codec as hc
from harry_amd import meshgen as mg
from oracle import oracle_py as op
from tests import util

GOLD = os.path.join(util.ROOT, "tests", "golden")
PLYS = sorted(f[:-4] for f in os.listdir(GOLD) if f.endswith(".ply") and ".dec." not in f)


def test_abi_exports_every_declared_symbol():
    hdr = open(os.path.join(util.ROOT, "include", "harry_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = sorted(set(re.findall(r"\b(hry_[a-z0-9_]+)\s*\(", hdr)))
    assert len(names) > 30
    L = nat.load()
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing
    assert L.hry_abi_version() == 6


def test_no_device_fails_loudly():
    """Without a HIP device the codec must refuse to exist (no CPU fallback)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(hc.HryError) as e:
        hc.Codec(0)
    assert e.value.code == nat.E_NODEVICE


@pytest.mark.parametrize("name", PLYS)
def test_ply_reader_matches_oracle(name):
    data = open(os.path.join(GOLD, name + ".ply"), "rb").read()
    a = hc.Mesh.from_ply(data)
    b = op.Mesh.from_ply(data)
    assert (a.nv, a.nf, a.ne, a.ntri) == (b.nv, b.nf, b.ne, b.ntri)
    assert np.array_equal(a.face_offsets(), b.face_offsets())
    assert np.array_equal(a.org(), b.org())
    assert np.array_equal(a.twin(), b.twin())
    for l in range(2):
        assert a.list_fmt(l) == b.list_fmt(l)
        assert np.array_equal(a.list_data(l), b.list_data(l))


def test_ply_writer_roundtrip_and_from_arrays():
    m = mg.with_face_props(mg.with_colors(mg.torus(9, 8, polys="mixed")))
    a = hc.Mesh.from_ply(m.to_ply())
    again = hc.Mesh.from_ply(a.to_ply(ascii=False))
    assert np.array_equal(a.org(), again.org()) and np.array_equal(a.twin(), again.twin())
    for l in range(2):
        assert np.array_equal(a.list_data(l), again.list_data(l))
    txt = a.to_ply(ascii=True)
    assert txt.startswith(b"ply\nformat ascii 1.0\n")
    b = hc.Mesh.from_arrays(m.verts, m.degrees, m.indices, m.face_props)
    assert np.array_equal(a.org(), b.org()) and np.array_equal(a.twin(), b.twin())
    for l in range(2):
        assert a.list_fmt(l) == b.list_fmt(l)
        assert np.array_equal(a.list_data(l), b.list_data(l))


def test_reader_rejects_garbage():
    with pytest.raises(hc.HryError):
        hc.Mesh.from_ply(b"not a ply file")
    with pytest.raises(hc.HryError):
        hc.Mesh.from_ply(mg.grid(4).to_ply()[:-7])


# ---------------------------------------------------------------- the walk
CTX_IOP, CTX_OP, CTX_ELEM, CTX_PART, CTX_VERT, CTX_NUMTRI, CTX_REGFACE, CTX_REGVTX = 0, 1, 2, 6, 8, 12, 14, 16
GROUPS = [(CTX_IOP, 1), (CTX_ELEM, 4), (CTX_PART, 2), (CTX_VERT, 4), (CTX_NUMTRI, 2)]


def conn_part_of_trace(tr, numtri_coded):
    """Oracle trace restricted to the connectivity part, with the symbols the product never materialises removed:
    reg_face/reg_vtx (single region: l = 0, h = t) and, for single-degree meshes, numtri (same reason)."""
    keep = (tr["ctx"] < CTX_REGFACE) | (tr["ctx"] >= 18)
    if not numtri_coded:
        keep &= ~((tr["ctx"] >= CTX_NUMTRI) & (tr["ctx"] < CTX_REGFACE))
    return tr[keep]


def check_walk_against_oracle(ply: bytes):
    a = hc.Mesh.from_ply(ply)
    o = op.Mesh.from_ply(ply)
    res = o.encode(trace=True)
    w = a.host_walk()
    assert np.array_equal(w["order_v"], res.order_vtx())
    assert np.array_equal(w["order_f"], res.order_face())
    assert np.array_equal(a.twin(), o.twin()), "repaired twins differ"
    n_conn, numtri_coded = int(w["info"][0]), bool(w["info"][1])
    tr = conn_part_of_trace(res.trace(), numtri_coded)
    conn = tr[:n_conn]
    assert len(tr) >= n_conn and (n_conn == len(tr) or tr["ctx"][n_conn] >= 18)
    assert np.all(conn["ctx"] < 18)
    # operations: symbol and evaluated model
    ops = conn[conn["ctx"] == CTX_OP]
    assert np.array_equal(w["op_sym"], ops["sym"].astype(np.uint8))
    assert np.array_equal(w["op_l"], ops["l"]) and np.array_equal(w["op_h"], ops["h"]) and np.array_equal(w["op_t"], ops["t"])
    assert np.array_equal(np.nonzero(conn["ctx"] == CTX_OP)[0], w["op_pos"])
    # ... and the table the device's operation model places its records with (op_position_table): operation i sits at
    # i + cum[j], j = the last group with thr[j] <= i
    j = np.searchsorted(w["op_thr"], np.arange(len(w["op_pos"])), side="right")
    assert np.array_equal(np.arange(len(w["op_pos"])) + np.concatenate([[0], w["op_cum"]])[j], w["op_pos"])
    # byte groups
    for g, (ctx, nb) in enumerate(GROUPS):
        val, pos = w[f"grp{g}_val"], w[f"grp{g}_pos"]
        for b in range(nb):
            sel = np.nonzero(conn["ctx"] == ctx + b)[0]
            assert np.array_equal(sel, pos + b), (g, b)
            assert np.array_equal(conn["sym"][sel], (val >> (8 * b)) & 0xff)
    return a, o, res, w


@pytest.mark.parametrize("name", [p for p in PLYS if p != "grid_double"])   # (lossless doubles: unspecified in the reference itself)
def test_walk_matches_oracle_on_golden_inputs(name):
    check_walk_against_oracle(open(os.path.join(GOLD, name + ".ply"), "rb").read())


@pytest.mark.parametrize("case", ["multi", "nonmanifold", "mixed_big", "open_quads", "ico"])
def test_walk_matches_oracle_on_generated(case):
    m = {"multi": lambda: mg.multi_component(12, 14, 16),
         "nonmanifold": lambda: mg.with_nonmanifold(mg.concat([mg.torus(20, 22, polys="mixed"), mg.torus(9, 10, polys="mixed", center=(4, 0, 0))]), 12, 6),
         "mixed_big": lambda: mg.torus(70, 64, polys="mixed"),
         "open_quads": lambda: mg.grid(23, 17, quads=True),
         "ico": lambda: mg.icosphere(4)}[case]()
    check_walk_against_oracle(m.to_ply())


def test_parse_quant_flags():
    assert hc.parse_quant_flags(["-l1", "-q14"]) == ([(1, -1, 14)], False)
    assert hc.parse_quant_flags(["-l", "1", "-a", "0", "-q", "14", "-a3", "-q10", "-c"]) == ([(1, 0, 14), (1, 3, 10)], True)
    with pytest.raises(ValueError):
        hc.parse_quant_flags(["-q14"])


@pytest.mark.parametrize("ncomp,nu", [(3, 4), (40, 6), (300, 4), (1500, 3), (90, 12)])
def test_start_face_order_many_components(ncomp, nu):
    """Every component after the first starts at the face std::unordered_set iteration would yield (App. B-1); the
    product derives that order analytically, the oracle uses the real container.  Crosses several rehash points."""
    m = mg.multi_component(ncomp, nu, nu + 1, polys="tri")
    check_walk_against_oracle(m.to_ply())


# ---------------------------------------------------------------- reading the reference's stream: the serial host half
def golden_variants(kind="small"):
    import json
    with open(os.path.join(GOLD, "manifest.json")) as f:
        man = json.load(f)
    return [(n, t, e["variants"][t]["flags"]) for n, e in sorted(man[kind].items()) for t in sorted(e["variants"])]


@pytest.mark.parametrize("name,tag,flags", golden_variants(), ids=[f"{n}.{t}" for n, t, _ in golden_variants()])
def test_host_stream_reader_on_reference_files(name, tag, flags):
    """Entropy decode + cut-border replay of files written by the reference binary: connectivity equals the oracle's
    decode of the same file (which is pinned to the reference's own decode), and the residual byte planes equal the
    data symbols of the stream (taken from the oracle's trace when it encodes the source PLY to this very file)."""
    data = open(os.path.join(GOLD, f"{name}.{tag}.hry"), "rb").read()
    m, order_v, vpl, fpl = hc.read_stream_host(data)
    o = op.Mesh.from_hry(data)
    assert (m.nv, m.nf, m.ne) == (o.nv, o.nf, o.ne)
    assert np.array_equal(m.face_offsets(), o.face_offsets())
    assert np.array_equal(m.org(), o.org())
    assert np.array_equal(m.twin(), o.twin())
    src = op.Mesh.from_ply(open(os.path.join(GOLD, name + ".ply"), "rb").read())
    quant, clear = util.flags_to_quant(flags)
    if quant or clear:
        src.requant(quant, clear)
    res = src.encode(trace=True)
    assert res.data == data
    tr = res.trace()
    vc, fc = len(order_v), m.nf
    data_syms = tr[tr["ctx"] >= 18]
    # per record: type symbol + data bytes; vertices first, then faces
    sv = (len(vpl) // vc if vc else 0) + 1
    sf = (len(fpl) // fc if fc else 0) + 1
    assert len(data_syms) == vc * sv + fc * sf
    vt = data_syms[:vc * sv].reshape(vc, sv)["sym"][:, 1:].astype(np.uint8)
    ft = data_syms[vc * sv:].reshape(fc, sf)["sym"][:, 1:].astype(np.uint8)
    assert np.array_equal(vpl.reshape(sv - 1, vc).T, vt)
    assert np.array_equal(fpl.reshape(sf - 1, fc).T, ft)


def test_host_stream_reader_rejects_corrupt_input():
    data = open(os.path.join(GOLD, "torus_tri.q14.hry"), "rb").read() if os.path.exists(os.path.join(GOLD, "torus_tri.q14.hry")) else None
    if data is None:
        n, t, _ = golden_variants()[0]
        data = open(os.path.join(GOLD, f"{n}.{t}.hry"), "rb").read()
    with pytest.raises(hc.HryError):
        hc.read_stream_host(data[:20])
    rng = np.random.default_rng(3)
    for trial in range(20):
        bad = bytearray(data)
        for k in rng.integers(len(data) // 2, len(data), 8):
            bad[k] ^= int(rng.integers(1, 256))
        try:
            hc.read_stream_host(bytes(bad))     # must either decode something or fail cleanly, never crash
        except hc.HryError:
            pass


# ---------------------------------------------------------------- several host threads (components after the first)
def walk_plain(ply: bytes, threads: int, min_faces: int = 0):
    os.environ["HRY_HOST_THREADS"] = str(threads)
    os.environ["HRY_PARALLEL_MIN_FACES"] = str(min_faces)
    try:
        m = hc.Mesh.from_ply(ply)
        return m.host_walk(plain=True), m.twin().copy()
    finally:
        del os.environ["HRY_HOST_THREADS"], os.environ["HRY_PARALLEL_MIN_FACES"]


def shuffled_faces(m, seed=5):
    """the same triangles in random file order (components no longer contiguous: the labelling's cross-range path)"""
    perm = np.random.default_rng(seed).permutation(m.nf)
    return mg.Mesh(m.verts, m.degrees[perm], m.indices.reshape(-1, 3)[perm].reshape(-1))


@pytest.mark.parametrize("case", ["multi_tri", "multi_mixed_nm", "tied_by_vertices", "many_small", "single", "shuffled", "soups"])
def test_threaded_walk_equals_sequential_walk(case):
    """The walk of the components after the first on several threads (component discovery, start-face order, vertex index
    bases computed up front) must reproduce the sequential walk array for array, including the repaired twins."""
    m = {"multi_tri": lambda: mg.multi_component(23, 14, 16, polys="tri"),
         "multi_mixed_nm": lambda: mg.with_nonmanifold(mg.multi_component(17, 12, 13, polys="mixed", seed=3), 40, 25),
         "tied_by_vertices": lambda: mg.with_nonmanifold(mg.concat([mg.torus(9, 10, center=(3.0 * i, 0, 0), seed=i) for i in range(12)]), 30, 60),
         "many_small": lambda: mg.multi_component(700, 3, 4, polys="tri"),
         "single": lambda: mg.torus(40, 44),
         "shuffled": lambda: shuffled_faces(mg.multi_component(12, 40, 42, polys="tri")),
         # edges shared by any number of faces: the walk's repairs of half-edge twins cut components in two -- the walk on several
         # threads notices and ends on one (host.hpp WalkMismatch): same outputs, same repaired twins
         "soups": lambda: mg.concat([mg.torus(12, 13, seed=1)] + [mg.soup(seed=s) for s in (5, 6, 13, 14, 16, 22)] + [mg.torus(10, 11, seed=2)])}[case]()
    ply = m.to_ply()
    seq, twin_seq = walk_plain(ply, 1)
    for threads in (2, 5):
        par, twin_par = walk_plain(ply, threads)
        assert np.array_equal(twin_seq, twin_par)
        for k in seq:
            assert np.array_equal(seq[k], par[k]), (k, threads)
    # and the plain walk agrees with the modelled one on everything both produce
    full = hc.Mesh.from_ply(ply).host_walk()
    for k in ("order_v", "order_f", "op_sym", "op_class", "grp0_val", "grp1_val", "grp2_val", "grp3_val", "grp4_val", "grp3_pos"):
        assert np.array_equal(full[k], seq[k]), k


# ---------------------------------------------------------------- decoder-side replay, cut at the directory's restart points
def replay(ply: bytes, threads: int, restarts: bool, min_faces: int = 0):
    os.environ["HRY_HOST_THREADS"] = str(threads)
    os.environ["HRY_PARALLEL_MIN_FACES"] = str(min_faces)
    try:
        m = hc.Mesh.from_ply(ply)
        dec, order_v, seg_start, seg_level, nrs = hc.walk_and_replay(m, restarts)
        return m, dec, order_v, seg_start, seg_level, nrs
    finally:
        del os.environ["HRY_HOST_THREADS"], os.environ["HRY_PARALLEL_MIN_FACES"]


@pytest.mark.parametrize("case", ["multi_tri", "multi_mixed", "shared_vertices", "many_small", "single"])
def test_replay_from_restart_points_equals_sequential_replay(case):
    """The replay of the connectivity planes started independently at every restart point of the chunked directory (spans on
    several host threads; spans that name older vertices afterwards, in order) rebuilds exactly what one sequential replay
    rebuilds: face offsets, origins, twins, decode order, component table and levels."""
    m = {"multi_tri": lambda: mg.multi_component(60, 30, 32, polys="tri"),
         "multi_mixed": lambda: mg.with_nonmanifold(mg.multi_component(40, 24, 26, polys="mixed", seed=3), 40, 25),
         "shared_vertices": lambda: mg.with_nonmanifold(mg.concat([mg.torus(40, 41, center=(3.0 * i, 0, 0), seed=i) for i in range(20)]), 30, 200),
         "many_small": lambda: mg.multi_component(3000, 3, 4, polys="tri"),
         "single": lambda: mg.torus(80, 84)}[case]()
    ply = m.to_ply()
    src, ref, ov, ss, sl, _ = replay(ply, 1, False)
    # the sequential replay inverts the walk: same polygon degrees in coding order, same vertex count
    assert ref.nf == src.nf and ref.ne == src.ne and len(ov) == src.nv
    for threads in (2, 5):
        _, dec, ov2, ss2, sl2, nrs = replay(ply, threads, True)
        if case != "single":
            assert nrs > 0, "the case must exercise restart points"
        assert np.array_equal(ref.face_offsets(), dec.face_offsets())
        assert np.array_equal(ref.org(), dec.org())
        assert np.array_equal(ref.twin(), dec.twin())
        assert np.array_equal(ov, ov2) and np.array_equal(ss, ss2) and np.array_equal(sl, sl2)
    # the lean loops (replay_triangles, replay_polygons: the ones above) against the generic replay_span, span by span
    os.environ["HRY_GENERIC_REPLAY"] = "1"
    try:
        _, dec, ov2, ss2, sl2, nrs = replay(ply, 3, True)
    finally:
        del os.environ["HRY_GENERIC_REPLAY"]
    assert np.array_equal(ref.face_offsets(), dec.face_offsets()) and np.array_equal(ref.org(), dec.org()) and np.array_equal(ref.twin(), dec.twin())
    assert np.array_equal(ov, ov2) and np.array_equal(ss, ss2) and np.array_equal(sl, sl2)


# ---------------------------------------------------------------- restart points INSIDE a component (round 6: border snapshots)
def _snapshot_case(case):
    return {"torus_tri": lambda: mg.torus(80, 84),
            "torus_quads": lambda: mg.torus(40, 44, polys="quad"),
            "torus_mixed": lambda: mg.torus(50, 52, polys="mixed"),
            "multi_mixed_nm": lambda: mg.with_nonmanifold(mg.multi_component(6, 24, 26, polys="mixed", seed=3), 40, 25),
            "shared_vertices": lambda: mg.with_nonmanifold(mg.concat([mg.torus(40, 41, center=(3.0 * i, 0, 0), seed=i) for i in range(12)]), 30, 200),
            "open_grid": lambda: mg.grid(60, 70)}[case]()


def _with_env(env, fn):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        return fn()
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v


@pytest.mark.parametrize("case", ["torus_tri", "torus_quads", "torus_mixed", "multi_mixed_nm", "shared_vertices", "open_grid"])
@pytest.mark.parametrize("spacing", [64, 257, 1000])
def test_replay_from_border_snapshots_equals_sequential_replay(case, spacing):
    """The walk notes the cut-border every `spacing` faces of a component (parts, vertices, triangle counts: no half-edges); the
    replay starts a span at every snapshot -- placeholders for the border's half-edges, counters of its own -- on several host
    threads and joins the spans afterwards.  Same arrays as ONE sequential replay (cbm/decoder.h:27-211): face offsets, origins,
    twins, decode order, component table and levels -- with and without the restart points at component starts, with the
    snapshots written by the one-thread walk, the two-core walk and the walk of the components on several threads."""
    gen = _snapshot_case(case)
    ply = gen.to_ply()
    src, ref, ov, ss, sl, _ = replay(ply, 1, False)
    seen_points = 0
    for threads, mode, split in ((2, 3, 0), (5, 3, 1), (3, 3, 0)):
        env = {"HRY_SNAPSHOT_FACES": spacing, "HRY_WALK_SPLIT": split, "HRY_WALK_RING": 1024}
        _, dec, ov2, ss2, sl2, (nrs, nsn) = _with_env(env, lambda: replay(ply, threads, mode))
        seen_points += nsn
        assert np.array_equal(ref.face_offsets(), dec.face_offsets())
        assert np.array_equal(ref.org(), dec.org())
        assert np.array_equal(ref.twin(), dec.twin())
        assert np.array_equal(ov, ov2) and np.array_equal(ss, ss2) and np.array_equal(sl, sl2)
    if not (case == "multi_mixed_nm" and spacing == 1000):
        assert seen_points > 0, "the case must exercise border snapshots"


@pytest.mark.parametrize("case", ["torus_tri", "torus_mixed", "multi_mixed_nm", "shared_vertices"])
def test_border_snapshots_are_the_same_whichever_loop_walks(case):
    """one-thread triangle / polygon loops, the generic loop, the two-core walk, components on several threads: one directory section"""
    gen = _snapshot_case(case)
    ply = gen.to_ply()
    want = _with_env({"HRY_SNAPSHOT_FACES": 100, "HRY_WALK_SPLIT": 0}, lambda: walk_plain(ply, 1))[0]["snap_section"]
    assert len(want) > 8
    for env, threads in (({"HRY_WALK_SPLIT": 1, "HRY_WALK_RING": 1024}, 2), ({"HRY_GENERIC_WALK": 1}, 1), ({"HRY_WALK_SPLIT": 0}, 5), ({"HRY_GENERIC_WALK": 1}, 4)):
        env = dict(env, HRY_SNAPSHOT_FACES=100)
        got = _with_env(env, lambda: walk_plain(ply, threads))[0]["snap_section"]
        assert np.array_equal(got, want), env
    # no snapshots asked for: no section
    assert len(_with_env({"HRY_SNAPSHOT_FACES": 0}, lambda: walk_plain(ply, 2))[0]["snap_section"]) == 0
    assert len(_with_env({"HRY_NO_SNAPSHOTS": 1, "HRY_SNAPSHOT_FACES": 100}, lambda: walk_plain(ply, 2))[0]["snap_section"]) == 0


@pytest.mark.parametrize("case", ["torus_tri", "torus_quads", "torus_mixed", "multi_mixed_nm", "shared_vertices", "open_grid"])
def test_border_snapshot_section_equals_the_oracles(case):
    """The directory section the product's walk writes is, byte for byte, the one in the oracle's restatement of the container --
    whose own sequential decode checks every snapshot against the state its replay is in at that moment."""
    gen = _snapshot_case(case)
    ply = gen.to_ply()
    for spacing in (90, 700):
        got = _with_env({"HRY_SNAPSHOT_FACES": spacing}, lambda: walk_plain(ply, 3))[0]["snap_section"].tobytes()
        o = op.Mesh.from_ply(ply)
        ref = o.encode_chunked(1024, spacing).data
        if not got:
            assert case == "multi_mixed_nm" and spacing == 700
            continue
        assert got in ref
        back = op.Mesh.from_hry_chunked(ref)      # (raises if a snapshot does not describe the replay's state)
        plain = op.Mesh.from_hry(op.Mesh.from_ply(ply).encode().data)
        assert np.array_equal(back.org(), plain.org()) and np.array_equal(back.twin(), plain.twin())
    # a damaged snapshot is refused by the oracle's decode (it is looked at, not skipped)
    bad = bytearray(ref)
    at = ref.index(got) + 12 + 13             # the first snapshot's next vertex (behind the section's three words and thirteen one-byte cursors)
    bad[at] ^= 1
    with pytest.raises(Exception):
        op.Mesh.from_hry_chunked(bytes(bad))


# ---------------------------------------------------------------- half-edge twin matching on several threads
@pytest.mark.parametrize("case", ["torus_tri", "mixed_nonmanifold", "open_grid", "duplicates"])
def test_threaded_twin_matching_equals_sequential(case):
    """structs/conn.h:201-214 pairs half-edges greedily in face order; the product deals the edges out to buckets and pairs
    every bucket with the same rule on its own thread.  Twins must be identical to the one-thread pass, also where an edge
    carries more than two faces or the same face occurs twice (order-dependent pairing)."""
    m = {"torus_tri": lambda: mg.torus(310, 300, seed=3),
         "mixed_nonmanifold": lambda: mg.with_nonmanifold(mg.torus(260, 250, polys="mixed", seed=5), 400, 150),
         "open_grid": lambda: mg.grid(400, 350),
         "duplicates": lambda: mg.concat([mg.torus(300, 300, seed=1), mg.torus(20, 20, seed=1)])}[case]()
    if case == "duplicates":   # the small torus a second time on the SAME vertices: every one of its edges carries four faces
        small = mg.torus(20, 20, seed=1)
        idx = np.concatenate([m.indices, small.indices.astype(np.uint32) + np.uint32(300 * 300)])
        m = mg.Mesh(m.verts, np.concatenate([m.degrees, small.degrees]), idx)
    def twins(threads):
        os.environ["HRY_HOST_THREADS"] = str(threads)
        try:
            return hc.Mesh.from_arrays(m.verts, m.degrees, m.indices).twin().copy()
        finally:
            del os.environ["HRY_HOST_THREADS"]
    ref = twins(1)
    assert len(ref) >= 1 << 18, "the case must be large enough for the threaded path"
    for t in (2, 7):
        assert np.array_equal(ref, twins(t)), t


POLY_CASES = {"mixed_torus": lambda: mg.torus(40, 36, polys="mixed"), "quad_torus": lambda: mg.torus(17, 20, polys="quad"),
              "mixed_multi_nm": lambda: mg.with_nonmanifold(mg.multi_component(9, 12, 14, polys="mixed", seed=6), 30, 20),
              "mixed_tiny": lambda: mg.torus(3, 4, polys="mixed")}


@pytest.mark.parametrize("case", ["torus", "open_grid", "ico", "multi_nm", "tiny"] + list(POLY_CASES))
@pytest.mark.parametrize("threads", [1, 3])
def test_lean_triangle_replay_equals_generic_replay(case, threads, monkeypatch):
    """cbm_replay.hpp: replay_triangles (few instructions per triangle, the headline decode's loop) and cbm_unwalk.cpp:
    replay_polygons (the same for polygon meshes, one span or spans on several threads) against replay_span on the same
    connectivity planes: borders, splits / unions (torus), several components, shared non-manifold vertices."""
    mesh = {"torus": lambda: mg.torus(40, 36), "open_grid": lambda: mg.grid(31, 17), "ico": lambda: mg.icosphere(4),
            "multi_nm": lambda: mg.with_nonmanifold(mg.multi_component(6, 9, 10, polys="tri"), 7, 4), "tiny": lambda: mg.grid(2), **POLY_CASES}[case]()
    monkeypatch.setenv("HRY_HOST_THREADS", str(threads))
    monkeypatch.setenv("HRY_PARALLEL_MIN_FACES", "1")
    out = []
    for generic in (False, True):
        if generic:
            monkeypatch.setenv("HRY_GENERIC_REPLAY", "1")
        m = hc.Mesh.from_arrays(mesh.verts, mesh.degrees, mesh.indices)
        dec, order_v, seg_start, seg_level, _ = hc.walk_and_replay(m, threads > 1)
        out.append((dec.face_offsets(), dec.org(), dec.twin(), order_v, seg_start, seg_level))
    for a, b in zip(*out):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("case", ["torus", "open_grid", "ico", "multi_nm", "tiny"] + list(POLY_CASES))
@pytest.mark.parametrize("threads", [1, 4])
def test_lean_triangle_walk_equals_generic_walk(case, threads, monkeypatch):
    """cbm_walk.cpp: walk_component_tri (the headline encode's loop) against the generic walk_component: same order, same
    symbols, same repaired twins."""
    mesh = {"torus": lambda: mg.torus(40, 36), "open_grid": lambda: mg.grid(31, 17), "ico": lambda: mg.icosphere(4),
            "multi_nm": lambda: mg.with_nonmanifold(mg.multi_component(6, 9, 10, polys="tri"), 7, 4), "tiny": lambda: mg.grid(2), **POLY_CASES}[case]()
    monkeypatch.setenv("HRY_HOST_THREADS", str(threads))
    monkeypatch.setenv("HRY_PARALLEL_MIN_FACES", "1")
    out = []
    for generic in (False, True):
        if generic:
            monkeypatch.setenv("HRY_GENERIC_WALK", "1")
        m = hc.Mesh.from_arrays(mesh.verts, mesh.degrees, mesh.indices)
        w = m.host_walk(plain=True)
        out.append((w, m.twin()))
    for k in out[0][0]:
        assert np.array_equal(out[0][0][k], out[1][0][k]), k
    assert np.array_equal(out[0][1], out[1][1])


@pytest.mark.parametrize("case", ["torus", "open_grid", "ico", "multi_nm", "multi_shared", "tiny", "torus_big"])
def test_triangle_walk_on_two_cores_equals_the_one_thread_loop(case, monkeypatch):
    """cbm_walk.cpp, round 5: the walking thread writes a trace of its decisions, a second thread expands it into the operation
    bytes, order_v / order_f, the marks and the explicitly named vertices -- entry for entry what the one-thread loop writes
    (meshes with borders, non-manifold edges and vertices, several components one after the other, components that name each
    other's vertices, splits and unions of the cut-border)."""
    def shared():
        parts = [mg.grid(5, 6, seed=s) for s in range(7)]
        b = mg.concat(parts)
        idx = b.indices.copy()
        nvp = parts[0].nv
        for k in range(1, 7):
            idx[idx == k * nvp + 3] = 3          # one vertex of every part becomes vertex 3 of the first
        return mg.Mesh(b.verts, b.degrees, idx, None)
    mesh = {"torus": lambda: mg.torus(40, 36), "open_grid": lambda: mg.grid(31, 17), "ico": lambda: mg.icosphere(4),
            "multi_nm": lambda: mg.with_nonmanifold(mg.multi_component(6, 9, 10, polys="tri"), 7, 4), "multi_shared": shared, "tiny": lambda: mg.grid(2),
            "torus_big": lambda: mg.torus(300, 280)}[case]()
    monkeypatch.setenv("HRY_HOST_THREADS", "2")
    monkeypatch.setenv("HRY_PARALLEL_MIN_FACES", "4000000000")   # (every component in the sequential loop)
    out = []
    for split, ring in (("0", None), ("1", None), ("1", "1024")):   # (the trace is a ring: the last round goes round a small one many times)
        monkeypatch.setenv("HRY_WALK_SPLIT", split)
        if ring:
            monkeypatch.setenv("HRY_WALK_RING", ring)
        m = hc.Mesh.from_arrays(mesh.verts, mesh.degrees, mesh.indices)
        w = m.host_walk(plain=True)
        out.append((w, m.twin()))
    for other in out[1:]:
        for k in out[0][0]:
            assert np.array_equal(out[0][0][k], other[0][k]), k
        assert np.array_equal(out[0][1], other[1])
    assert len(out[0][0]["order_f"]) == mesh.nf


@pytest.mark.timeout(120)
def test_parallel_walk_in_a_forked_child(monkeypatch):
    """the helper threads of parallel_for (host/thread_pool.cpp) are kept between calls; a forked child has none of them and must
    start its own instead of waiting for the parent's"""
    import ctypes as C
    from harry_amd import _native as nat
    monkeypatch.setenv("HRY_PARALLEL_MIN_FACES", "1000")
    m0 = mg.multi_component(6, 60, 62, seed=4, polys="mixed")
    m = hc.Mesh.from_ply(m0.to_ply())
    L = nat.load()

    def walk():
        a, w = m.clone(), C.c_void_p()
        nat.check(L.hry_walk_run_plain(a.h, C.byref(w)))
        L.hry_walk_free(w)

    walk()                      # the parent's helpers exist now
    pid = os.fork()
    if pid == 0:
        try:
            walk()
            os._exit(0)
        except BaseException:
            os._exit(1)
    _, status = os.waitpid(pid, 0)
    assert os.WIFEXITED(status) and os.WEXITSTATUS(status) == 0
    walk()


# ---------------------------------------------------------------- PLY reader / writer on several threads, polygons of several degrees
@pytest.mark.parametrize("threads", [1, 5])
@pytest.mark.parametrize("face_props", [False, True])
def test_polygon_ply_reader_and_writer_on_threads(threads, face_props, monkeypatch):
    """host/ply_io.cpp: a binary little-endian file with polygons of several degrees is read block by block behind one pass over
    the count bytes, and written with every thread knowing where its faces go (formats/ply/reader.cc:323-429, writer.cc:60-104):
    same arrays as the generator holds, same bytes behind the header as the generator's own writer."""
    monkeypatch.setenv("HRY_HOST_THREADS", str(threads))
    gen = mg.with_nonmanifold(mg.torus(260, 250, polys="mixed", seed=11), 30, 20)
    assert gen.nf >= 1 << 16
    if face_props:
        fp = np.zeros(gen.nf, dtype=[("quality", "<f4"), ("flags", "u1")])
        fp["quality"] = np.arange(gen.nf, dtype=np.float32) * 0.25
        fp["flags"] = np.arange(gen.nf) % 251
        gen = mg.Mesh(gen.verts, gen.degrees, gen.indices, fp)
    ply = gen.to_ply()
    m = hc.Mesh.from_ply(ply)
    assert np.array_equal(m.org(), gen.indices)
    assert np.array_equal(np.diff(m.face_offsets()), gen.degrees)
    body = lambda b: b[b.index(b"end_header\n") + 11:]
    assert body(m.to_ply()) == body(ply)
    # damaged input: an index out of range, a degree below 3, a truncated face element
    bad = bytearray(ply)
    at = len(ply) - len(body(ply)) + gen.nv * gen.verts.dtype.itemsize
    if not face_props:
        bad[at + 1:at + 5] = (gen.nv + 7).to_bytes(4, "little")
        with pytest.raises(hc.HryError):
            hc.Mesh.from_ply(bytes(bad))
        with pytest.raises(hc.HryError):
            hc.Mesh.from_ply(ply[:-9])


def test_range_reciprocal_equals_division(tmp_path):
    """r = floor(range / total) of the compat recurrence (coder.h:69-70) through the precomputed reciprocal: exact for every range <= 2^63."""
    import shutil
    import subprocess
    cxx = shutil.which("g++") or shutil.which("c++")
    if not cxx:
        pytest.skip("no host compiler")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "magic_check")
    subprocess.run([cxx, "-O2", "-std=c++17", "-I", os.path.join(root, "harry_amd", "csrc", "device"), os.path.join(root, "tests", "tools", "magic_check.cpp"), "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and " 0 bad" in r.stdout, r.stdout + r.stderr
