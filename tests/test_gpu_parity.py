"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI, against
 (1) the reference's own outputs committed under tests/golden/ and (2) the CPU oracle on the same seeded inputs.
Bar: bit-exact (bytes, integers, IEEE bit patterns)."""
import json
import os

import numpy as np
import pytest

from harry_amd import codec as hc
from harry_amd import meshgen as mg
from oracle import oracle_py as op
from tests import util
from tests.test_host_cpu import conn_part_of_trace

pytestmark = pytest.mark.gpu

GOLD = os.path.join(util.ROOT, "tests", "golden")
with open(os.path.join(GOLD, "manifest.json")) as _f:
    MANIFEST = json.load(_f)
SMALL = [(n, t, v) for n, e in sorted(MANIFEST["small"].items()) for t, v in sorted(e["variants"].items())]


@pytest.fixture(scope="module")
def cx():
    c = hc.Codec(0)
    yield c
    c.close()


def magic_of(t: int):
    """Reference implementation (Python ints) of the reciprocal the device precomputes for a context total: floor(R / t) ==
    (R * magic) >> (64 + shift) for every R <= 2^63 (codec_math.hpp: make_magic)."""
    if t < 2:
        return 0, 0
    k = t.bit_length() - 1
    if t & (t - 1) == 0:
        return 1 << 63, k - 1
    return ((1 << (64 + k)) // t + 1) & (2 ** 64 - 1), k


def test_reciprocal_reference_is_exact_on_the_coders_range():
    rng = np.random.default_rng(3)
    for t in [2, 3, 5, 7, 255, 256, 257, 65535, 65537, 2 ** 31 - 1, 2 ** 31 + 1, 2 ** 32 - 1] + [int(x) for x in rng.integers(2, 2 ** 32, 200)]:
        m, sh = magic_of(t)
        for R in [1 << 63, (1 << 63) - 1, (1 << 62) + 1, t * ((1 << 63) // t), t * ((1 << 63) // t) - 1] + [int(x) for x in rng.integers(1 << 62, 1 << 63, 20)]:
            assert (R * m) >> (64 + sh) == R // t, (t, R)


# ---------------------------------------------------------------- end to end vs the reference's bytes
@pytest.mark.parametrize("name,tag,v", SMALL, ids=[f"{n}.{t}" for n, t, _ in SMALL])
def test_encode_compat_byte_identical_to_reference(cx, name, tag, v):
    ply = open(os.path.join(GOLD, name + ".ply"), "rb").read()
    ref = open(os.path.join(GOLD, f"{name}.{tag}.hry"), "rb").read()
    m = hc.Mesh.from_ply(ply)
    quant, clear = util.flags_to_quant(v["flags"])
    if quant or clear:
        cx.requant(m, quant, clear)
    got = cx.write_hry(m)
    assert got == ref


def test_requant_integers_identical(cx):
    """-q: quantised integers identical to the oracle (which is pinned to the reference), incl. uchar sources."""
    for mesh, quant in ((mg.with_colors(mg.torus(30, 34, normals=True)), [(1, -1, 6)]),
                        (mg.torus(40, 42, normals=True), [(1, 0, 14), (1, 1, 14), (1, 2, 14), (1, 3, 10), (1, 4, 10), (1, 5, 10)]),
                        (mg.with_face_props(mg.grid(31, 17)), [(0, -1, 5), (1, -1, 11)]),
                        (mg.negated(mg.grid(25)), [(1, -1, 14)]),
                        (mg.icosphere(3), [(1, -1, 20)])):
        ply = mesh.to_ply()
        a = hc.Mesh.from_ply(ply)
        o = op.Mesh.from_ply(ply)
        cx.requant(a, quant)
        o.requant(quant)
        for l in range(2):
            assert a.list_fmt(l) == o.list_fmt(l)
            assert np.array_equal(a.list_data(l), o.list_data(l))
            if a.list_stride(l):
                assert np.array_equal(a.list_min(l), o.list_min(l)) and np.array_equal(a.list_max(l), o.list_max(l))


def test_bounds_kernel_bitexact(cx):
    m = mg.negated(mg.grid(40))          # all-negative coordinates: max stays FLT_MIN
    v = m.verts.copy()
    v["x"][5] = np.float32(-0.0) ; v["x"][9] = np.float32(0.0)   # +-0 tie: the first one wins
    v["y"][:] = np.float32(-3.5)
    m = mg.Mesh(v, m.degrees, m.indices)
    ply = m.to_ply()
    a = hc.Mesh.from_ply(ply)
    o = op.Mesh.from_ply(ply)
    cx.bounds(a)
    assert np.array_equal(a.list_min(1), o.list_min(1))
    assert np.array_equal(a.list_max(1), o.list_max(1))
    big = mg.with_colors(mg.torus(300, 310, normals=True))
    ply = big.to_ply()
    a, o = hc.Mesh.from_ply(ply), op.Mesh.from_ply(ply)
    cx.bounds(a)
    assert np.array_equal(a.list_min(1), o.list_min(1)) and np.array_equal(a.list_max(1), o.list_max(1))


# ---------------------------------------------------------------- stage by stage vs the oracle's trace
def check_stages(cx, mesh: mg.Mesh, quant=()):
    ply = mesh.to_ply()
    a = hc.Mesh.from_ply(ply)
    o = op.Mesh.from_ply(ply)
    if quant:
        cx.requant(a, quant)
        o.requant(quant)
    res = o.encode(trace=True)
    got = cx.write_hry(a, keep_stages=True)
    assert np.array_equal(cx.stage("order_v", np.uint32), res.order_vtx())
    assert np.array_equal(cx.stage("order_f", np.uint32), res.order_face())
    assert np.array_equal(cx.stage("twin", np.uint32), o.twin())
    n_conn, vc, fc, sv, sf, ns, numtri_coded, _ = [int(x) for x in cx.stage("layout", np.uint32)]
    tr = conn_part_of_trace(res.trace(), bool(numtri_coded))
    assert len(tr) == ns
    # residual byte planes == the data symbols the oracle coded
    vplanes = cx.stage("vplanes").reshape(sv - 1, vc) if sv > 1 else np.zeros((0, vc), np.uint8)
    vt = tr[n_conn:n_conn + vc * sv].reshape(vc, sv)
    assert np.array_equal(vplanes.T, vt["sym"][:, 1:].astype(np.uint8))
    fplanes = cx.stage("fplanes").reshape(sf - 1, fc) if sf > 1 else np.zeros((0, fc), np.uint8)
    ft = tr[n_conn + vc * sv:].reshape(fc, sf)
    assert np.array_equal(fplanes.T, ft["sym"][:, 1:].astype(np.uint8))
    # exact model evaluation: (l, h - l | l, flags, reciprocal of t)
    rec = cx.stage("rec", np.dtype([("magic", "<u8"), ("x", "<u4"), ("meta", "<u4")]))
    sym_l = cx.stage("sym_l", np.uint32)
    assert np.array_equal(sym_l, tr["l"].astype(np.uint32))
    sub = tr["h"] == tr["t"]
    noop = sub & (tr["l"] == 0)
    assert np.array_equal((rec["meta"] >> 6) & 1, sub.astype(np.uint32))
    assert np.array_equal((rec["meta"] >> 7) & 1, noop.astype(np.uint32))
    assert np.array_equal(rec["x"], np.where(sub, tr["l"], tr["h"] - tr["l"]).astype(np.uint32))
    ts = np.unique(tr["t"])
    table = {int(t): magic_of(int(t)) for t in ts}
    exp_magic = np.array([table[int(t)][0] for t in tr["t"]], np.uint64)
    exp_shift = np.array([table[int(t)][1] for t in tr["t"]], np.uint32)
    assert np.array_equal(rec["magic"], exp_magic)
    assert np.array_equal(rec["meta"] & 63, exp_shift)
    # serial recurrence: r_k and bit positions
    r, S = cx.stage("r", np.uint64), cx.stage("S", np.uint32)
    R, shifts = 1 << 63, 0
    for k in range(ns):
        l, h, t = int(tr["l"][k]), int(tr["h"][k]), int(tr["t"][k])
        assert S[k] == shifts, k
        if l == 0 and h == t:
            continue
        rr = R // t
        assert int(r[k]) == rr, k
        R = rr * (h - l) if h < t else R - rr * l
        while R <= (1 << 62):
            R <<= 1
            shifts += 1
    assert got == res.data
    return got


def test_stages_triangles_lossless(cx):
    check_stages(cx, mg.torus(18, 20))


def test_stages_mixed_polygons_components_nonmanifold(cx):
    m = mg.with_nonmanifold(mg.concat([mg.torus(14, 16, polys="mixed"), mg.torus(8, 10, polys="mixed", center=(4, 0, 0))]), 5, 3)
    check_stages(cx, m)


def test_stages_quantised_with_normals_colors_faceprops(cx):
    m = mg.with_face_props(mg.with_colors(mg.torus(12, 14, normals=True)))
    check_stages(cx, m, quant=[(1, 0, 14), (1, 1, 14), (1, 2, 14), (1, 3, 10), (1, 4, 10), (1, 5, 10), (0, -1, 7)])


def test_stages_open_quads(cx):
    check_stages(cx, mg.grid(15, 11, quads=True))


# ---------------------------------------------------------------- range coder back end on explicit triples
def test_range_coder_backend_known_answers(cx):
    with open(os.path.join(GOLD, "kat.json")) as f:
        kat = json.load(f)
    for triples, code in kat["range_lht"]:
        lht = np.array(triples, dtype=np.uint64).reshape(-1, 3)
        if lht[:, 2].max() >= 2 ** 32:
            continue
        assert cx.range_encode_lht(lht) == bytes.fromhex(code)
    rng = np.random.default_rng(5)
    for n in (0, 1, 63, 64, 65, 1000, 20000):
        t = rng.integers(2, 1 << 20, n).astype(np.uint64)
        l = (rng.random(n) * t).astype(np.uint64) % t
        h = np.minimum(l + 1 + (rng.random(n) * (t - l)).astype(np.uint64), t)
        h[::7] = t[::7]
        lht = np.stack([l, h, t], 1)
        assert cx.range_encode_lht(lht) == op.range_encode_lht(lht)


def test_range_coder_long_carry_chains(cx):
    """Symbols with l + count == t near the top of the interval force long carry propagation through the stream."""
    n = 5000
    t = np.full(n, 1 << 16, np.uint64)
    l = t - 1
    h = t.copy()
    lht = np.stack([l, h, t], 1)
    lht[::97] = [0, 1, 1 << 16]
    assert cx.range_encode_lht(lht) == op.range_encode_lht(lht)


@pytest.mark.parametrize("period", [97, 4099, 70001, 0])
def test_range_coder_carries_across_rows_wavefronts_and_blocks(cx, period):
    """The same over 160 000 output words: the carry kernels take a row of 64 words per ballot pair, sixteen rows per wavefront,
    4 096 words per block and the blocks' pairs through a scan of their own (kernels.hip: k_carry_*) -- runs of all-ones words of
    every length up to the whole stream, ended by a word that generates."""
    n = 320000
    t = np.full(n, 1 << 16, np.uint64)
    lht = np.stack([t - 1, t.copy(), t], 1)
    if period:
        lht[::period] = [0, 1, 1 << 16]
    lht[-1] = [(1 << 16) - 1, 1 << 16, 1 << 16]
    assert cx.range_encode_lht(lht) == op.range_encode_lht(lht)
    rng = np.random.default_rng(period + 1)
    k = rng.integers(0, n, 200)
    lht[k] = np.stack([rng.integers(0, 1 << 15, 200), rng.integers(1 << 15, 1 << 16, 200), np.full(200, 1 << 16)], 1).astype(np.uint64)
    assert cx.range_encode_lht(lht) == op.range_encode_lht(lht)


# ---------------------------------------------------------------- larger seeded meshes vs the oracle
@pytest.mark.parametrize("case", ["torus150_q14", "ico5", "multi40", "nm_big", "grid_quads"])
def test_encode_compat_matches_oracle_on_larger_meshes(cx, case):
    mesh, quant = {
        "torus150_q14": (lambda: mg.torus(150, 150, seed=2), [(1, -1, 14)]),
        "ico5": (lambda: mg.icosphere(5), []),
        "multi40": (lambda: mg.multi_component(40, 20, 22), []),
        "nm_big": (lambda: mg.with_nonmanifold(mg.torus(60, 64, polys="mixed"), 30, 12), []),
        "grid_quads": (lambda: mg.grid(120, 90, quads=True), [(1, -1, 12)]),
    }[case]
    ply = mesh().to_ply()
    a = hc.Mesh.from_ply(ply)
    o = op.Mesh.from_ply(ply)
    if quant:
        cx.requant(a, quant)
        o.requant(quant)
    assert cx.write_hry(a) == o.encode().data


def test_empty_face_list_and_tiny_meshes(cx):
    for m in (mg.grid(2), mg.grid(2, quads=True), mg.grid(3)):
        ply = m.to_ply()
        a, o = hc.Mesh.from_ply(ply), op.Mesh.from_ply(ply)
        assert cx.write_hry(a) == o.encode().data


def test_recurrence_on_host_core_or_on_one_wavefront_gives_identical_bytes(cx):
    """The serial range-register recurrence runs on a host core by default (streamed behind the device kernels) and on a single
    wavefront (k_rchain) with HRY_FLAG_DEVICE_RECURRENCE; bytes are those of the reference either way."""
    for mesh, quant in ((mg.torus(60, 64, polys="mixed", normals=True), []), (mg.torus(90, 90), [(1, -1, 14)])):
        ply = mesh.to_ply()
        a, b, o = hc.Mesh.from_ply(ply), hc.Mesh.from_ply(ply), op.Mesh.from_ply(ply)
        if quant:
            cx.requant(a, quant); cx.requant(b, quant); o.requant(quant)
        want = o.encode().data
        assert cx.write_hry(a) == want
        assert cx.write_hry(b, flags=hc.FLAG_DEVICE_RECURRENCE) == want


# ---------------------------------------------------------------- reading files written by the reference binary
@pytest.mark.parametrize("name,tag,v", SMALL, ids=[f"{n}.{t}" for n, t, _ in SMALL])
def test_decode_reference_file_identical_to_reference_decode(cx, name, tag, v):
    """.hry written by the reference -> arrays identical to the PLY the reference itself decoded from it."""
    hry = open(os.path.join(GOLD, f"{name}.{tag}.hry"), "rb").read()
    dec = open(os.path.join(GOLD, f"{name}.{tag}.dec.ply"), "rb").read()
    m = cx.read_hry(hry)
    vrec, degs, idx, frec = util.parse_ref_decoded_ply(dec, m.list_stride(1), m.list_stride(0))
    assert m.nv == len(vrec) and m.nf == len(degs)
    assert np.array_equal(np.diff(m.face_offsets()).astype(np.uint8), degs)
    assert np.array_equal(m.org(), idx)
    assert np.array_equal(m.list_data(1), vrec)
    assert np.array_equal(m.list_data(0), frec)


@pytest.mark.parametrize("case", ["torus150_q14", "multi40", "nm_big", "normals_colors"])
def test_decode_compat_stream_matches_oracle_on_larger_meshes(cx, case):
    mesh, quant = {
        "torus150_q14": (lambda: mg.torus(150, 150, seed=2), [(1, -1, 14)]),
        "multi40": (lambda: mg.multi_component(40, 20, 22), []),
        "nm_big": (lambda: mg.with_nonmanifold(mg.torus(60, 64, polys="mixed"), 30, 12), []),
        "normals_colors": (lambda: mg.with_face_props(mg.with_colors(mg.torus(40, 44, normals=True))), [(1, 0, 12), (1, 1, 12), (1, 2, 12), (1, 3, 9), (1, 4, 9), (1, 5, 9)]),
    }[case]
    o = op.Mesh.from_ply(mesh().to_ply())
    if quant:
        o.requant(quant)
    data = o.encode().data
    want = op.Mesh.from_hry(data)
    got = cx.read_hry(data)
    assert np.array_equal(got.face_offsets(), want.face_offsets())
    assert np.array_equal(got.org(), want.org()) and np.array_equal(got.twin(), want.twin())
    for l in range(2):
        assert got.list_fmt(l) == want.list_fmt(l)
        assert np.array_equal(got.list_data(l), want.list_data(l))
    # and the product's own compat encoder output decodes the same way
    a = hc.Mesh.from_ply(mesh().to_ply())
    if quant:
        cx.requant(a, quant)
    again = cx.read_hry(cx.write_hry(a))
    for l in range(2):
        assert np.array_equal(again.list_data(l), want.list_data(l))


def _fans(n_fans, spokes, seed=3):
    """n_fans cones: a hub joined to a closed ring of `spokes` vertices (hub valence = spokes); every third fan shares its ring
    with a second hub on the other side (two hubs over the same ring edges)"""
    rng = np.random.default_rng(seed)
    pts, tris = [], []
    for k in range(n_fans):
        base = len(pts)
        pts.append((3.0 * k, 0.0, 1.0))
        for i in range(spokes):
            a = 2 * np.pi * i / spokes
            pts.append((3.0 * k + np.cos(a), np.sin(a), 0.01 * rng.random()))
        for i in range(spokes):
            tris.append((base, base + 1 + i, base + 1 + (i + 1) % spokes))
        if k % 3 == 0:
            hub2 = len(pts)
            pts.append((3.0 * k, 0.0, -1.0))
            for i in range(spokes):
                tris.append((hub2, base + 1 + (i + 1) % spokes, base + 1 + i))
    v = np.zeros(len(pts), dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4")])
    p = np.asarray(pts, np.float32)
    v["x"], v["y"], v["z"] = p[:, 0], p[:, 1], p[:, 2]
    t = np.asarray(tris, np.uint32)
    return mg.Mesh(v, np.full(len(t), 3, np.uint8), t.reshape(-1))


@pytest.mark.parametrize("n_fans,spokes", [(3, 300), (40, 60), (4300, 50)])
def test_device_twin_matching_with_hubs(cx, n_fans, spokes):
    """twins.hip leaves vertices with more than a few dozen half-edges to the host (a few hubs: patched; thousands: the host's
    matcher does everything); either way the twins are the sequential matcher's (structs/conn.h:201-214)"""
    mesh = _fans(n_fans, spokes)
    ply = mesh.to_ply()
    a = hc.Mesh.from_ply(ply)
    cx.upload(a)                                   # twin matching on the device + hubs on the host
    assert np.array_equal(a.twin(), op.Mesh.from_ply(ply).twin())
    b = hc.Mesh.from_ply(ply)
    assert np.array_equal(b.twin(), a.twin())      # host-only matcher (no context)
    assert cx.write_hry(a) == op.Mesh.from_ply(ply).encode().data
