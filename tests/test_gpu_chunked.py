"""GPU parity tests of the chunked profile (.hry v0.2) through the C ABI.

Parity definition: the chunked container produced by the HIP path is byte-identical to the oracle's
CPU restatement of the same container; it decodes -- on the GPU and on the CPU oracle -- to exactly the mesh the
REFERENCE decodes from its own v0.1 stream of the same input (committed tests/golden/*.dec.ply)."""
import json
import os

import numpy as np
import pytest

from harry_amd import codec as hc
from harry_amd import meshgen as mg
from oracle import oracle_py as op
from tests import util

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


def same_mesh(a, b):
    assert (a.nv, a.nf, a.ne) == (b.nv, b.nf, b.ne)
    assert np.array_equal(a.face_offsets(), b.face_offsets())
    assert np.array_equal(a.org(), b.org())
    for l in range(2):
        assert a.list_fmt(l) == b.list_fmt(l)
        assert np.array_equal(a.list_data(l), b.list_data(l)), f"list {l} differs"


@pytest.mark.parametrize("name,tag,v", SMALL, ids=[f"{n}.{t}" for n, t, _ in SMALL])
def test_chunked_roundtrip_equals_reference_decode(cx, name, tag, v):
    ply = open(os.path.join(GOLD, name + ".ply"), "rb").read()
    dec_ref = open(os.path.join(GOLD, f"{name}.{tag}.dec.ply"), "rb").read()
    m = hc.Mesh.from_ply(ply)
    o = op.Mesh.from_ply(ply)
    quant, clear = util.flags_to_quant(v["flags"])
    if quant or clear:
        cx.requant(m, quant, clear)
        o.requant(quant, clear)
    for chunk in (0, 257):
        got = cx.write_hry(m.clone(), profile=hc.PROFILE_CHUNKED, chunk_syms=chunk)
        want = o.clone().encode_chunked(chunk).data
        assert got == want, "chunked container differs from the oracle's"
        dec = cx.read_hry(got)
        vrec, degs, idx, frec = util.parse_ref_decoded_ply(dec_ref, dec.list_stride(1), dec.list_stride(0))
        assert np.array_equal(np.diff(dec.face_offsets()).astype(np.uint8), degs)
        assert np.array_equal(dec.org(), idx)
        assert np.array_equal(dec.list_data(1), vrec)
        assert np.array_equal(dec.list_data(0), frec)


@pytest.mark.parametrize("case", ["torus150_q14", "ico5", "multi40", "nm_big", "grid_quads", "colors_normals"])
def test_chunked_larger_meshes(cx, case):
    mesh, quant = {
        "torus150_q14": (lambda: mg.torus(150, 150, seed=2), [(1, -1, 14)]),
        "ico5": (lambda: mg.icosphere(5), []),
        "multi40": (lambda: mg.multi_component(40, 20, 22), []),
        "nm_big": (lambda: mg.with_nonmanifold(mg.torus(60, 64, polys="mixed"), 30, 12), []),
        "grid_quads": (lambda: mg.grid(120, 90, quads=True), [(1, -1, 12)]),
        "colors_normals": (lambda: mg.with_face_props(mg.with_colors(mg.torus(90, 80, normals=True))), [(1, 0, 14), (1, 1, 14), (1, 2, 14), (1, 3, 10), (1, 4, 10), (1, 5, 10)]),
    }[case]
    ply = mesh().to_ply()
    a, o = hc.Mesh.from_ply(ply), op.Mesh.from_ply(ply)
    if quant:
        cx.requant(a, quant)
        o.requant(quant)
    compat = o.clone().encode().data
    ref_dec = op.Mesh.from_hry(compat)
    for chunk in (0, 4096):
        got = cx.write_hry(a.clone(), profile=hc.PROFILE_CHUNKED, chunk_syms=chunk)
        assert got == o.clone().encode_chunked(chunk).data
        same_mesh(cx.read_hry(got), ref_dec)                  # GPU decode == decode of the reference-format stream
        same_mesh(op.Mesh.from_hry_chunked(got), ref_dec)     # independent CPU decode of the GPU's container


@pytest.mark.parametrize("case", ["multi_tri", "shared_vertices"])
def test_chunked_decode_from_restart_points(cx, case, monkeypatch):
    """Encode with the threaded walk, decode with the replay cut at the directory's restart points on several host threads:
    same container as the oracle's, same mesh as the reference-format decode."""
    monkeypatch.setenv("HRY_HOST_THREADS", "4")
    monkeypatch.setenv("HRY_PARALLEL_MIN_FACES", "0")
    mesh = {"multi_tri": lambda: mg.multi_component(60, 30, 32, polys="tri"),
            "shared_vertices": lambda: mg.with_nonmanifold(mg.concat([mg.torus(40, 41, center=(3.0 * i, 0, 0), seed=i) for i in range(20)]), 30, 200)}[case]()
    ply = mesh.to_ply()
    a, o = hc.Mesh.from_ply(ply), op.Mesh.from_ply(ply)
    ref_dec = op.Mesh.from_hry(o.clone().encode().data)
    got = cx.write_hry(a.clone(), profile=hc.PROFILE_CHUNKED)
    assert got == o.clone().encode_chunked(0).data
    same_mesh(cx.read_hry(got), ref_dec)


@pytest.mark.parametrize("case", ["torus150_q14", "grid_quads_q12", "ico5_q10", "colors_normals", "open_grid_q8", "face_props"])
@pytest.mark.parametrize("faces,slice_", [(64, 64), (1000, 4096)])
def test_chunked_pipelined_decode(cx, case, faces, slice_, monkeypatch):
    """The pipelined decode (replay publishes its progress; uploads, candidates and the reconstruction chain of every
    finished slice of vertices run behind it) gives exactly the mesh of the sequential pipeline and of the reference-format
    decode.  Forced onto small meshes here with tiny publication intervals / slices so that many slices, patches of late twin
    links and ring reloads are exercised."""
    mesh, quant = {
        "torus150_q14": (lambda: mg.torus(150, 150, seed=2), [(1, -1, 14)]),
        "grid_quads_q12": (lambda: mg.grid(120, 90, quads=True), [(1, -1, 12)]),
        "ico5_q10": (lambda: mg.icosphere(5), [(1, -1, 10)]),
        "colors_normals": (lambda: mg.with_colors(mg.torus(90, 80, normals=True)), [(1, 0, 14), (1, 1, 14), (1, 2, 14), (1, 3, 10), (1, 4, 10), (1, 5, 10)]),
        "open_grid_q8": (lambda: mg.grid(200, 150), [(1, -1, 8)]),
        "face_props": (lambda: mg.with_face_props(mg.torus(90, 80, seed=4)), [(1, -1, 12)]),   # lossless face attributes next to the quantised vertices
    }[case]
    ply = mesh().to_ply()
    a, o = hc.Mesh.from_ply(ply), op.Mesh.from_ply(ply)
    cx.requant(a, quant)
    o.requant(quant)
    ref_dec = op.Mesh.from_hry(o.clone().encode().data)
    got = cx.write_hry(a.clone(), profile=hc.PROFILE_CHUNKED)
    monkeypatch.setenv("HRY_NO_PIPELINE", "1")
    plain = cx.read_hry(got)
    monkeypatch.delenv("HRY_NO_PIPELINE")
    monkeypatch.setenv("HRY_PIPELINE_MIN_VERTICES", "0")
    monkeypatch.setenv("HRY_PIPELINE_FACES", str(faces))
    monkeypatch.setenv("HRY_PIPELINE_SLICE", str(slice_))
    piped = cx.read_hry(got)
    same_mesh(piped, plain)
    same_mesh(piped, ref_dec)
    assert np.array_equal(piped.twin(), plain.twin())


@pytest.mark.parametrize("case", ["torus150_q14", "colors_normals", "torus_lossless", "quads_q12", "mixed_nm_multi", "open_grid_q8"])
@pytest.mark.parametrize("spacing", [97, 1500])
def test_border_snapshots_container_and_decodes(cx, case, spacing, monkeypatch):
    """Round 6, restart points INSIDE a component: the walk notes the cut-border every `spacing` faces of a component in the
    container's directory (parts, vertices, triangle counts), a decoder starts a span of the replay at every snapshot on a host
    thread of its own and joins the spans.  The container equals the oracle's restatement byte for byte (whose sequential
    decode checks every snapshot against its own replay); the decode -- one sequence ignoring the snapshots, spans beside each
    other, spans beside the publishing first stretch of the pipelined decode -- equals the reference-format decode."""
    mesh, quant = {
        "torus150_q14": (lambda: mg.torus(150, 150, seed=2), [(1, -1, 14)]),
        "colors_normals": (lambda: mg.with_colors(mg.torus(90, 80, normals=True)), [(1, 0, 14), (1, 1, 14), (1, 2, 14), (1, 3, 10), (1, 4, 10), (1, 5, 10)]),
        "torus_lossless": (lambda: mg.torus(100, 90, seed=5), []),
        "quads_q12": (lambda: mg.torus(70, 64, polys="quad"), [(1, -1, 12)]),
        "mixed_nm_multi": (lambda: mg.with_nonmanifold(mg.multi_component(5, 30, 34, polys="mixed", seed=3), 40, 25), []),
        "open_grid_q8": (lambda: mg.grid(200, 150), [(1, -1, 8)]),
    }[case]
    monkeypatch.setenv("HRY_SNAPSHOT_FACES", str(spacing))
    monkeypatch.setenv("HRY_PARALLEL_MIN_FACES", "0")
    monkeypatch.setenv("HRY_HOST_THREADS", "5")
    ply = mesh().to_ply()
    a, o = hc.Mesh.from_ply(ply), op.Mesh.from_ply(ply)
    if quant:
        cx.requant(a, quant)
        o.requant(quant)
    ref_dec = op.Mesh.from_hry(o.clone().encode().data)
    got = cx.write_hry(a.clone(), profile=hc.PROFILE_CHUNKED, chunk_syms=1000)
    want = o.clone().encode_chunked(1000, spacing).data
    assert got == want
    assert len(got) > len(o.clone().encode_chunked(1000, 0).data)          # (the section is there)
    same_mesh(op.Mesh.from_hry_chunked(got), ref_dec)                      # the oracle's own decode, which checks the snapshots
    monkeypatch.setenv("HRY_NO_PIPELINE", "1")
    monkeypatch.setenv("HRY_NO_SNAPSHOT_REPLAY", "1")
    plain = cx.read_hry(got)                                               # one sequence: the snapshots are skipped
    same_mesh(plain, ref_dec)
    monkeypatch.delenv("HRY_NO_SNAPSHOT_REPLAY")
    spans = cx.read_hry(got)                                               # spans beside each other (cut_border_replay)
    same_mesh(spans, ref_dec)
    assert np.array_equal(spans.twin(), plain.twin())
    monkeypatch.delenv("HRY_NO_PIPELINE")
    monkeypatch.setenv("HRY_PIPELINE_MIN_VERTICES", "0")
    for faces, slice_, piece in ((64, 64, 128), (1000, 4096, 1024)):
        monkeypatch.setenv("HRY_PIPELINE_FACES", str(faces))
        monkeypatch.setenv("HRY_PIPELINE_SLICE", str(slice_))
        monkeypatch.setenv("HRY_PIPELINE_LAST_PIECE", str(piece))          # (what is left behind the replay goes in pieces)
        piped = cx.read_hry(got)                                           # (where the pipelined decode applies: its first stretch publishes, the others run beside it)
        same_mesh(piped, ref_dec)
        assert np.array_equal(piped.twin(), plain.twin())
    # a container without snapshots is what it was before this round
    monkeypatch.setenv("HRY_NO_SNAPSHOTS", "1")
    assert cx.write_hry(a.clone(), profile=hc.PROFILE_CHUNKED, chunk_syms=1000) == o.clone().encode_chunked(1000, 0).data


@pytest.mark.parametrize("how", ["host-threads", "device-analysis"])
def test_repaired_twins_that_split_a_component_take_one_thread(cx, how, monkeypatch):
    """Triangle soups: edges shared by any number of faces.  The walk re-pairs such half-edges as it meets the faces
    (cbm/encoder.h:150,193-198), and a repair can cut a component -- as the analysis in front of a walk on several threads saw it --
    in two.  The encode notices (the component consumes fewer faces than promised), drops the attempt, matches the twins afresh
    and walks on one thread: the reference's stream byte for byte, the oracle's container, the reference-format decode."""
    monkeypatch.setenv("HRY_PARALLEL_MIN_FACES", "1")
    monkeypatch.setenv("HRY_HOST_THREADS", "6")
    if how == "device-analysis":
        monkeypatch.setenv("HRY_DEVICE_ANALYSIS_MIN_FACES", "1")
    gen = mg.concat([mg.torus(12, 13, seed=1)] + [mg.soup(seed=s) for s in (5, 6, 13, 14, 16, 22)] + [mg.torus(10, 11, seed=2)])
    ply = gen.to_ply()
    o = op.Mesh.from_ply(ply)
    compat = o.clone().encode().data
    ref_dec = op.Mesh.from_hry(compat)
    for _ in range(2):   # (twice: what the first encode leaves on the device must not matter)
        a = hc.Mesh.from_ply(ply)
        assert cx.write_hry(a.clone(), profile=hc.PROFILE_COMPAT) == compat
        got = cx.write_hry(a.clone(), profile=hc.PROFILE_CHUNKED, chunk_syms=1000)
        assert got == o.clone().encode_chunked(1000).data
        same_mesh(cx.read_hry(got), ref_dec)
        # a resident mesh: its device copy of the twins is the dropped attempt's until the whole array has gone up again
        cx.upload(a)
        assert cx.write_hry(a, profile=hc.PROFILE_CHUNKED, chunk_syms=1000) == got


def test_damaged_border_snapshots_are_refused(cx, monkeypatch):
    """every word and byte of the snapshots' section: a changed one either decodes to the same mesh (padding) or is refused"""
    monkeypatch.setenv("HRY_SNAPSHOT_FACES", "400")
    monkeypatch.setenv("HRY_PARALLEL_MIN_FACES", "0")
    monkeypatch.setenv("HRY_HOST_THREADS", "4")
    monkeypatch.setenv("HRY_NO_PIPELINE", "1")
    m = mg.torus(40, 36, seed=3)
    a = hc.Mesh.from_ply(m.to_ply())
    got = cx.write_hry(a.clone(), profile=hc.PROFILE_CHUNKED, chunk_syms=1000)
    ref = cx.read_hry(got)
    monkeypatch.setenv("HRY_NO_SNAPSHOTS", "1")
    without = cx.write_hry(a.clone(), profile=hc.PROFILE_CHUNKED, chunk_syms=1000)
    monkeypatch.delenv("HRY_NO_SNAPSHOTS")
    assert len(got) > len(without)
    # the section: where the two containers part company (behind the restart points' count), as long as their difference
    first = next(i for i in range(len(without)) if got[i] != without[i])
    sec0, sec1 = first + 1, first + 1 + (len(got) - len(without))   # (the count's top byte differs; the section follows the counters: none here)
    rng = np.random.default_rng(1)
    refused = same = 0
    for at in list(range(sec0 - 4, sec0 + 80)) + list(rng.integers(sec0, sec1, 120)):
        bad = bytearray(got)
        bad[at] ^= 1 << int(rng.integers(0, 8))
        try:
            dec = cx.read_hry(bytes(bad))
        except hc.HryError:
            refused += 1
            continue
        same += 1
        assert np.array_equal(dec.org(), ref.org()) and np.array_equal(dec.twin(), ref.twin()) and np.array_equal(dec.list_data(1), ref.list_data(1))
    assert refused > 100


def test_headline_workload_full_size(cx):
    """BASELINE configs[1] at its full size (closed torus 708 x 708 = 1 002 528 triangles, -l1 -q14), every default of the
    product path: threaded walk, pipelined decode, wavefront team on the reconstruction chain.
    * compat profile: the .hry bytes equal the CPU oracle's (= the reference's, byte-pinned on the committed fixtures);
    * chunked profile: the container equals the oracle's restatement of it, and the GPU decode equals the oracle's decode of
      the reference-format stream array for array;
    * size-independent property: the multiset of decoded vertex records equals the multiset of quantised input records."""
    mesh = mg.torus(708, 708, seed=2, sigma=1e-4)
    ply = mesh.to_ply()
    a, o = hc.Mesh.from_ply(ply), op.Mesh.from_ply(ply)
    quant = [(1, -1, 14)]
    cx.requant(a, quant)
    o.requant(quant)
    q_in = a.list_data(1).reshape(a.nv, -1).view(np.uint16)[:, ::2].astype(np.uint64)   # quantised values, low half of each slot
    compat_ref = o.clone().encode().data
    assert cx.write_hry(a.clone(), profile=hc.PROFILE_COMPAT) == compat_ref
    ref_dec = op.Mesh.from_hry(compat_ref)
    got = cx.write_hry(a.clone(), profile=hc.PROFILE_CHUNKED)
    assert got == o.clone().encode_chunked(0).data
    dec = cx.read_hry(got)
    same_mesh(dec, ref_dec)
    same_mesh(cx.read_hry(compat_ref), ref_dec)     # the reference-format stream through the device reconstruction
    q_out = dec.list_data(1).reshape(dec.nv, -1).view(np.uint16)[:, ::2].astype(np.uint64)
    key = lambda q: np.sort((q[:, 0] << 28) | (q[:, 1] << 14) | q[:, 2])
    assert np.array_equal(key(q_in), key(q_out))


def test_chunked_entropy_decode_planes(cx):
    """k_chunk_decode inverts k_chunk_encode symbol for symbol (checked before any mesh logic)."""
    m = mg.torus(64, 60, polys="mixed", normals=True)
    a = hc.Mesh.from_ply(m.to_ply())
    got = cx.write_hry(a, profile=hc.PROFILE_CHUNKED, chunk_syms=1000, keep_stages=True)
    vplanes = cx.stage("vplanes")
    fplanes = cx.stage("fplanes")
    cx.read_hry(got, keep_stages=True)
    syms = cx.stage("dec_syms")
    nsym = cx.stage("dec_nsym", np.uint32)
    tail = syms[int(nsym[:21].sum()):]
    assert np.array_equal(tail[:len(vplanes)], vplanes)
    assert np.array_equal(tail[len(vplanes):len(vplanes) + len(fplanes)], fplanes)


def test_high_valence_vertex_uses_fan_walk(cx):
    """A cone apex with 40 incident triangles exceeds the candidate table of the reconstruction kernel."""
    n = 40
    ang = np.linspace(0, 2 * np.pi, n, endpoint=False)
    ring1 = np.stack([np.cos(ang), np.sin(ang), np.zeros(n)], 1)
    ring2 = np.stack([2 * np.cos(ang + 0.05), 2 * np.sin(ang + 0.05), -np.ones(n)], 1)
    pts = np.concatenate([[[0, 0, 1]], ring1, ring2]).astype(np.float32)
    tris = []
    for i in range(n):
        j = (i + 1) % n
        tris += [[0, 1 + i, 1 + j], [1 + i, 1 + n + i, 1 + n + j], [1 + i, 1 + n + j, 1 + j]]
    verts = np.empty(len(pts), dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4")])
    verts["x"], verts["y"], verts["z"] = pts[:, 0], pts[:, 1], pts[:, 2]
    mesh = mg.Mesh(verts, np.full(len(tris), 3, np.uint8), np.array(tris, np.uint32).reshape(-1))
    # start the traversal away from the apex so that the apex is coded late, with many coded neighbours
    mesh = mg.Mesh(verts, mesh.degrees, np.roll(np.array(tris, np.uint32), -n, axis=0).reshape(-1))
    ply = mesh.to_ply()
    a, o = hc.Mesh.from_ply(ply), op.Mesh.from_ply(ply)
    ref_dec = op.Mesh.from_hry(o.clone().encode().data)
    got = cx.write_hry(a, profile=hc.PROFILE_CHUNKED)
    same_mesh(cx.read_hry(got), ref_dec)


def test_decode_rejects_corrupt_input(cx):
    a = hc.Mesh.from_ply(mg.torus(12, 12).to_ply())
    good = cx.write_hry(a, profile=hc.PROFILE_CHUNKED)
    with pytest.raises(hc.HryError):
        cx.read_hry(b"\x00" * 40)
    with pytest.raises(hc.HryError):
        cx.read_hry(good[:len(good) // 2])
    bad = bytearray(good)
    bad[5] = 7   # unknown minor version
    with pytest.raises(hc.HryError):
        cx.read_hry(bytes(bad))


@pytest.mark.parametrize("pipelined", [False, True])
def test_decode_survives_damaged_payload(cx, pipelined, monkeypatch):
    """Flipped bytes anywhere behind the header: the decoder either reports an error or returns some mesh -- it must neither
    crash, nor hang (the wavefront hand-overs of the chain are bounded), nor leave the context unusable.  Both pipelines."""
    mesh = mg.torus(70, 64, seed=5)
    a = hc.Mesh.from_ply(mesh.to_ply())
    cx.requant(a, [(1, -1, 12)])
    good = cx.write_hry(a.clone(), profile=hc.PROFILE_CHUNKED, chunk_syms=2048)
    if pipelined:
        monkeypatch.setenv("HRY_PIPELINE_MIN_VERTICES", "0")
        monkeypatch.setenv("HRY_PIPELINE_FACES", "256")
        monkeypatch.setenv("HRY_PIPELINE_SLICE", "512")
    else:
        monkeypatch.setenv("HRY_NO_PIPELINE", "1")
    ref = cx.read_hry(good)
    rng = np.random.default_rng(11)
    hdr = 200   # well inside the directory for this mesh; the header itself is covered by test_decode_rejects_corrupt_input
    outcomes = {"error": 0, "mesh": 0}
    for trial in range(24):
        bad = bytearray(good)
        for _ in range(1 + trial % 4):
            k = int(rng.integers(hdr, len(bad)))
            bad[k] ^= int(rng.integers(1, 256))
        try:
            cx.read_hry(bytes(bad))
            outcomes["mesh"] += 1
        except hc.HryError:
            outcomes["error"] += 1
    assert outcomes["error"] + outcomes["mesh"] == 24
    same_mesh(cx.read_hry(good), ref)   # the context still decodes the intact stream


def with_integer_props(m: mg.Mesh, seed=7) -> mg.Mesh:
    """Smooth integer-valued vertex properties of every PLY integer type next to the float coordinates."""
    rng = np.random.default_rng(seed)
    names = list(m.verts.dtype.names)
    dt = [(n, m.verts.dtype[n]) for n in names] + [("pi32", "<i4"), ("pu32", "<u4"), ("pi16", "<i2"), ("pu16", "<u2"), ("pi8", "i1"), ("pu8", "u1")]
    v = np.zeros(m.nv, dtype=dt)
    for n in names:
        v[n] = m.verts[n]
    base = (m.verts["x"].astype(np.float64) * 900 + m.verts["y"].astype(np.float64) * 300)
    v["pi32"] = (base * 1000).astype(np.int64).astype(np.int32) + rng.integers(-3, 4, m.nv, dtype=np.int32)
    v["pu32"] = (base * 1000 + 3_000_000_000).astype(np.int64).astype(np.uint32)
    v["pi16"] = np.clip(base * 10, -32000, 32000).astype(np.int16) + rng.integers(-2, 3, m.nv).astype(np.int16)
    v["pu16"] = (np.clip(base * 10, -32000, 32000) + 32768).astype(np.uint16)
    v["pi8"] = np.clip(base / 12, -120, 120).astype(np.int8)
    v["pu8"] = (np.clip(base / 12, -120, 120) + 128).astype(np.uint8)
    return mg.Mesh(v, m.degrees, m.indices, m.face_props)


@pytest.mark.parametrize("case", ["int_props_tri", "int_props_mixed", "mixed_normals_lossless", "quant_mixed_widths"])
def test_chunked_component_types_and_chain_variants(cx, case):
    """Every component type the reference stores (float, (u)int32, (u)int16, (u)int8) through the reconstruction chain,
    on meshes whose traversal mixes long runs with frequent batch cuts; decode must equal the reference-format decode."""
    mesh, quant = {
        "int_props_tri": (lambda: with_integer_props(mg.torus(70, 75, seed=5)), []),
        "int_props_mixed": (lambda: with_integer_props(mg.with_nonmanifold(mg.torus(50, 56, polys="mixed", seed=6), 20, 8)), []),
        "mixed_normals_lossless": (lambda: mg.torus(120, 130, polys="mixed", normals=True, seed=8), []),
        "quant_mixed_widths": (lambda: mg.torus(110, 90, normals=True, seed=9), [(1, 0, 20), (1, 1, 7), (1, 2, 16), (1, 3, 9), (1, 4, 24), (1, 5, 3)]),
    }[case]
    ply = mesh().to_ply()
    a, o = hc.Mesh.from_ply(ply), op.Mesh.from_ply(ply)
    if quant:
        cx.requant(a, quant)
        o.requant(quant)
    ref_dec = op.Mesh.from_hry(o.clone().encode().data)
    got = cx.write_hry(a.clone(), profile=hc.PROFILE_CHUNKED)
    assert got == o.clone().encode_chunked(0).data
    same_mesh(cx.read_hry(got), ref_dec)
    # the reference-format stream of the same mesh through the host reader + the same device reconstruction
    same_mesh(cx.read_hry(cx.write_hry(a.clone())), ref_dec)


def test_pipelined_decode_after_another_mesh(cx, monkeypatch):
    """The slices of the pipelined decode run on connectivity that is uploaded behind the replay.  A twin link made after the
    publication a slice rests on can point into the part of the device arrays that is not uploaded yet -- which holds whatever
    the previous decode left there.  Found by tests/tools/chain_stress.py: the first decode after a different mesh saw extra
    candidates (a fan walk that continued through the previous mesh's faces); the kernels now stop at the uploaded half-edges."""
    other = mg.torus(138, 156, polys="quad", seed=25, sigma=3.5e-2).to_ply()
    a0 = hc.Mesh.from_ply(other)
    cx.requant(a0, [(1, -1, 4)])
    first = cx.write_hry(a0, profile=hc.PROFILE_CHUNKED)
    ply = mg.icosphere(6, seed=52, sigma=1.5e-2).to_ply()
    a, o = hc.Mesh.from_ply(ply), op.Mesh.from_ply(ply)
    cx.requant(a, [(1, -1, 7)])
    o.requant([(1, -1, 7)])
    ref_dec = op.Mesh.from_hry(o.clone().encode().data)
    got = cx.write_hry(a.clone(), profile=hc.PROFILE_CHUNKED)
    monkeypatch.setenv("HRY_PIPELINE_MIN_VERTICES", "0")
    for faces, slice_ in ((1529, 10752), (700, 4096), (3000, 8192)):
        monkeypatch.setenv("HRY_NO_PIPELINE", "1")
        cx.read_hry(first)                       # fills the device's connectivity arrays with another mesh
        monkeypatch.delenv("HRY_NO_PIPELINE")
        monkeypatch.setenv("HRY_PIPELINE_FACES", str(faces))
        monkeypatch.setenv("HRY_PIPELINE_SLICE", str(slice_))
        same_mesh(cx.read_hry(got), ref_dec)


# ---- the component analysis on the device (analysis.cpp) against the host's (cbm_walk.cpp: analyse_components) ----------------------
@pytest.mark.gpu
@pytest.mark.parametrize("case", ["multi", "multi_tri", "slivers", "slivers_quad", "shuffled", "one", "hub", "big", "big_shuffled", "big_one"])
def test_device_component_analysis_equals_the_hosts(cx, case):
    """connected components (one-sided twins and non-manifold slivers included), coding order by the reference's start-face sequence,
    faces / half-edges / new vertices per component, face and vertex intervals, groups of components that share a vertex"""
    rng = np.random.default_rng(11)
    if case == "multi":
        g = mg.multi_component(40, 9, 11, seed=3, polys="mixed")
    elif case == "multi_tri":
        g = mg.multi_component(25, 12, 7, seed=4)
    elif case == "slivers":
        g = mg.with_nonmanifold(mg.multi_component(30, 10, 12, seed=5, polys="mixed"), 120, 60, seed=4)
    elif case == "slivers_quad":
        g = mg.with_nonmanifold(mg.multi_component(12, 14, 9, seed=6, polys="quad"), 50, 25, seed=2)
    elif case == "shuffled":   # faces in random order: components are not contiguous in memory
        b = mg.with_nonmanifold(mg.multi_component(20, 8, 9, seed=7, polys="mixed"), 40, 20, seed=3)
        deg = b.degrees.astype(np.int64); offs = np.concatenate(([0], np.cumsum(deg)))
        perm = rng.permutation(b.nf)
        idx = np.concatenate([b.indices[offs[f]:offs[f + 1]] for f in perm])
        g = mg.Mesh(b.verts, b.degrees[perm], idx.astype(np.uint32), None)
    elif case in ("big", "big_shuffled"):   # more faces than a workgroup labels in LDS (16 384): edges that leave a workgroup's faces go to the union-find in HBM
        b = mg.with_nonmanifold(mg.multi_component(28, 44, 41, seed=9, polys="mixed"), 200, 90, seed=5)
        if case == "big":
            g = b
        else:               # ... nearly all of them, with the faces in random order
            deg = b.degrees.astype(np.int64); offs = np.concatenate(([0], np.cumsum(deg)))
            perm = rng.permutation(b.nf)
            idx = np.concatenate([b.indices[offs[f]:offs[f + 1]] for f in perm])
            g = mg.Mesh(b.verts, b.degrees[perm], idx.astype(np.uint32), None)
    elif case == "big_one":
        g = mg.torus(210, 190, polys="mixed")
    elif case == "one":
        g = mg.torus(30, 40, polys="mixed")
    else:   # many components that all touch ONE vertex (a hub): one group
        parts = [mg.grid(4, 5, seed=s) for s in range(12)]
        b = mg.concat(parts)
        idx = b.indices.copy()
        nvp = parts[0].nv
        for k in range(1, 12):
            idx[idx == k * nvp] = 0        # the first vertex of every part becomes vertex 0
        g = mg.Mesh(b.verts, b.degrees, idx, None)
    m = hc.Mesh.from_arrays(g.verts, g.degrees, g.indices)
    cx.analysis_check(m)


@pytest.mark.gpu
def test_encode_with_the_device_analysis_equals_the_oracle(cx, monkeypatch):
    """the walk from the device's tables (every component in place on the host threads) gives the container the oracle writes"""
    monkeypatch.setenv("HRY_DEVICE_ANALYSIS_MIN_FACES", "1")
    monkeypatch.setenv("HRY_PARALLEL_MIN_FACES", "1")
    for seed, polys in ((3, "mixed"), (4, "tri"), (5, "quad")):
        g = mg.with_nonmanifold(mg.multi_component(20, 10, 12, seed=seed, polys=polys), 60, 30, seed=seed)
        ply = g.to_ply()
        a, o = hc.Mesh.from_ply(ply), op.Mesh.from_ply(ply)
        got = cx.write_hry(a, profile=hc.PROFILE_CHUNKED)
        assert got == o.encode_chunked(hc.container_info(got)["chunk_syms"]).data


@pytest.mark.gpu
@pytest.mark.parametrize("batch", ["1", "300", "small-slots", "direct-1", "direct-0", "direct-8-small-slots", None, "off"])
def test_encode_beside_the_walk_equals_the_oracle(cx, monkeypatch, batch):
    """EncodePipeline (round 5): the finished groups' runs of the coding order go to the planes while the other groups are still
    walked -- in batches of one group, of a few, with slots so small that groups are sent in pieces, in the production size (here: one
    batch behind the walk), and not at all; vertex
    planes of quantised positions + normals, face planes of face properties, the polygons' triangle counts, repaired twins"""
    monkeypatch.setenv("HRY_DEVICE_ANALYSIS_MIN_FACES", "1")
    monkeypatch.setenv("HRY_PARALLEL_MIN_FACES", "1")
    monkeypatch.setenv("HRY_HOST_THREADS", "6")
    if batch == "off":
        monkeypatch.setenv("HRY_NO_ENCODE_PIPELINE", "1")
    elif batch == "small-slots":   # slots of 64 entries: most groups outgrow them and go through the sending thread in pieces
        monkeypatch.setenv("HRY_ENCODE_PIPELINE_SLOT", "64")
        monkeypatch.setenv("HRY_ENCODE_PIPELINE_BATCH", "40")
    elif batch and batch.startswith("direct-"):   # runs of at least so many entries are copied from the walk's (registered) arrays; 0: none
        monkeypatch.setenv("HRY_ENCODE_PIPELINE_DIRECT", batch.split("-")[1])
        monkeypatch.setenv("HRY_ENCODE_PIPELINE_BATCH", "200")
        if batch.endswith("small-slots"):
            monkeypatch.setenv("HRY_ENCODE_PIPELINE_SLOT", "64")
    elif batch:
        monkeypatch.setenv("HRY_ENCODE_PIPELINE_BATCH", batch)
    cases = [(mg.with_nonmanifold(mg.multi_component(24, 10, 12, seed=7, polys="mixed"), 80, 40, seed=7), []),
             (mg.with_nonmanifold(mg.multi_component(17, 9, 11, seed=8, polys="tri"), 30, 30, seed=8), [(1, -1, 12)]),
             (mg.with_face_props(mg.multi_component(12, 9, 11, seed=9, polys="quad")), []),
             (mg.concat([mg.torus(14 + 2 * i, 12 + i, seed=20 + i, normals=True, center=(3.0 * i, 0, 0), polys="mixed") for i in range(9)]),
              [(1, 0, 14), (1, 1, 14), (1, 2, 14), (1, 3, 10), (1, 4, 10), (1, 5, 10)])]
    for g, quant in cases:
        ply = g.to_ply()
        a, o = hc.Mesh.from_ply(ply), op.Mesh.from_ply(ply)
        if quant:
            cx.requant(a, quant)
            o.requant(quant)
        got = cx.write_hry(a, profile=hc.PROFILE_CHUNKED)
        assert got == o.encode_chunked(hc.container_info(got)["chunk_syms"]).data
        # ... and a second encode on the same context (recycled device arrays, a rank table of another mesh)
        b = hc.Mesh.from_ply(ply)
        if quant:
            cx.requant(b, quant)
        assert cx.write_hry(b, profile=hc.PROFILE_CHUNKED) == got


@pytest.mark.gpu
@pytest.mark.parametrize("min_streams", ["1", "0"])
def test_encoder_in_two_kernels_writes_the_same_streams(cx, monkeypatch, min_streams):
    """k_chunk_model + k_chunk_ranges (round 5: the model a wavefront per stream, the range registers a lane per stream; what a
    container of thousands of streams takes) against the oracle's container, like the one kernel: streams of every length (the
    growing chunk schedule, plane ends, 512-symbol connectivity chunks), lossless floats and quantised values, planes that keep the
    reference's initial counts, more streams than a wavefront has lanes and fewer"""
    monkeypatch.setenv("HRY_ENCODE_SPLIT_MIN_STREAMS", min_streams)
    cases = [(mg.torus(60, 50, seed=5), [], 1024), (mg.torus(33, 21, seed=6, polys="mixed", normals=True), [(1, -1, 11)], 512),
             (mg.with_nonmanifold(mg.multi_component(9, 13, 15, seed=4, polys="mixed"), 9, 5, seed=3), [], 0), (mg.grid(3), [], 0),
             (mg.with_face_props(mg.multi_component(6, 9, 11, seed=9, polys="tri")), [(1, -1, 14)], 2048)]
    for g, quant, chunk in cases:
        ply = g.to_ply()
        a, o = hc.Mesh.from_ply(ply), op.Mesh.from_ply(ply)
        if quant:
            cx.requant(a, quant)
            o.requant(quant)
        got = cx.write_hry(a, profile=hc.PROFILE_CHUNKED, chunk_syms=chunk)
        assert got == o.encode_chunked(hc.container_info(got)["chunk_syms"]).data
        dec = cx.read_hry(got)
        o2 = op.Mesh.from_ply(ply)
        if quant:
            o2.requant(quant)
        ref = op.Mesh.from_hry(o2.encode().data)
        assert np.array_equal(dec.org(), ref.org()) and np.array_equal(dec.list_data(1), ref.list_data(1))
