"""Pin the CPU oracle (oracle/hry_oracle.cc) against outputs of the UNMODIFIED reference.

Fixtures in tests/golden/ were produced by tests/golden/make_golden.py + `make -C oracle kat` in the build
container (reference binary / reference headers).  The reference ships no tests of its own,
so these reference-generated vectors are the pin.  CPU only.
"""
import hashlib
import json
import os
import subprocess

import numpy as np
import pytest

from oracle import oracle_py as op
from tests import util
from harry_amd import meshgen as mg


def _small_variants(manifest):
    for name, e in sorted(manifest["small"].items()):
        for tag, v in sorted(e["variants"].items()):
            yield name, tag, v


def _load_manifest():
    with open(os.path.join(util.ROOT, "tests", "golden", "manifest.json")) as f:
        return json.load(f)


SMALL = list(_small_variants(_load_manifest()))


@pytest.mark.parametrize("name,tag,v", SMALL, ids=[f"{n}.{t}" for n, t, _ in SMALL])
def test_encode_byte_identical(golden_dir, name, tag, v):
    ply = open(os.path.join(golden_dir, name + ".ply"), "rb").read()
    ref = open(os.path.join(golden_dir, f"{name}.{tag}.hry"), "rb").read()
    quant, clear = util.flags_to_quant(v["flags"])
    _, res = op.encode_ply(ply, quant, clear)
    assert hashlib.sha256(ref).hexdigest() == v["hry_sha256"]
    assert res.data == ref


@pytest.mark.parametrize("name,tag,v", SMALL, ids=[f"{n}.{t}" for n, t, _ in SMALL])
def test_decode_array_identical(golden_dir, name, tag, v):
    hry = open(os.path.join(golden_dir, f"{name}.{tag}.hry"), "rb").read()
    dec = open(os.path.join(golden_dir, f"{name}.{tag}.dec.ply"), "rb").read()
    m = op.Mesh.from_hry(hry)
    vrec, degs, idx, frec = util.parse_ref_decoded_ply(dec, m.list_stride(1), m.list_stride(0))
    assert m.nv == len(vrec) and m.nf == len(degs)
    fo = m.face_offsets()
    assert np.array_equal(np.diff(fo).astype(np.uint8), degs)
    assert np.array_equal(m.org(), idx)
    assert np.array_equal(m.list_data(1), vrec)
    assert np.array_equal(m.list_data(0), frec)


@pytest.mark.parametrize("name,tag,v", [s for s in SMALL if s[1] == "ll"], ids=[n for n, t, _ in SMALL if t == "ll"])
def test_lossless_roundtrip_up_to_permutation(golden_dir, name, tag, v):
    """decode(encode(x)) == x as a multiset of faces over vertex records (SURVEY finding 0-4)."""
    ply = open(os.path.join(golden_dir, name + ".ply"), "rb").read()
    src = op.Mesh.from_ply(ply)
    src_faces = util.canonical_faces(src.list_data(1), np.diff(src.face_offsets()), src.org())
    res = src.encode()
    dec = op.Mesh.from_hry(res.data)
    dec_faces = util.canonical_faces(dec.list_data(1), np.diff(dec.face_offsets()), dec.org())
    assert src_faces == dec_faces


def test_requant_of_quantised_hry(golden_dir, manifest):
    for name, e in manifest["requant_of_hry"].items():
        src = open(os.path.join(golden_dir, e["src"]), "rb").read()
        ref = open(os.path.join(golden_dir, name + ".hry"), "rb").read()
        m = op.Mesh.from_hry(src)
        quant, clear = util.flags_to_quant(e["flags"])
        m.requant(quant, clear)
        assert m.encode().data == ref


BIG = {"torus150": lambda: mg.torus(150, 150, seed=2), "multi40": lambda: mg.multi_component(40, 20, 22),
       "ico5": lambda: mg.icosphere(5), "nm_big": lambda: mg.with_nonmanifold(mg.torus(60, 64, polys="mixed"), 30, 12)}


@pytest.mark.parametrize("name", sorted(BIG))
def test_big_cases_hash(manifest, name):
    """Regenerated inputs (not committed): the oracle must reproduce the reference's output hash."""
    e = manifest["big"][name]
    ply = BIG[name]().to_ply()
    if hashlib.sha256(ply).hexdigest() != e["ply_sha256"]:
        pytest.skip("synthetic generator produced different bytes on this platform (libm/numpy); fixture not comparable")
    for tag, v in e["variants"].items():
        quant, clear = util.flags_to_quant(v["flags"])
        _, res = op.encode_ply(ply, quant, clear)
        assert len(res.data) == v["hry_bytes"]
        assert hashlib.sha256(res.data).hexdigest() == v["hry_sha256"]


# ---------------------------------------------------------------- function-level known answers
@pytest.fixture(scope="module")
def kat(golden_dir):
    with open(os.path.join(golden_dir, "kat.json")) as f:
        return json.load(f)


def test_kat_survey_values():
    """Values captured from the reference headers during the survey."""
    L = op.lib()
    f = lambda x: int(np.float32(x).view(np.uint32))
    assert L.ho_kat_encode_delta_f32(f(1.25), f(1.0)) == 0x00400000
    assert L.ho_kat_encode_delta_f32(f(-0.5), f(0.25)) == 0xc0ffffff
    assert L.ho_kat_encode_delta_u(1000, 1003, 2, 14) == 5
    assert L.ho_kat_predict_u(10, 16380, 3, 2, 14) == 16383
    assert L.ho_kat_predict_u(5, 3, 100, 2, 14) == 0
    assert op.range_encode_bytes(b"Hello, my name is Max!").hex() == "243303c575148c9f1471a5adf322f6df45c6fdcee782dac33acb3e1a00"


def test_kat_delta_f32(kat):
    L = op.lib()
    for raw, pred, enc, dec in kat["delta_f32"]:
        assert L.ho_kat_encode_delta_f32(raw, pred) == enc
        assert L.ho_kat_decode_delta_f32(enc, pred) == dec


def test_kat_delta_u(kat):
    L = op.lib()
    for b, q, raw, pred, enc, dec in kat["delta_u"]:
        assert L.ho_kat_encode_delta_u(raw, pred, b, q) == enc, (b, q, raw, pred)
        assert L.ho_kat_decode_delta_u(enc, pred, b, q) == dec, (b, q, raw, pred)


def test_kat_predict(kat):
    L = op.lib()
    for b, q, v0, v1, v2, r in kat["predict_u"]:
        assert L.ho_kat_predict_u(v0, v1, v2, b, q) == r
    for v0, v1, v2, r in kat["predict_f32"]:
        assert L.ho_kat_predict_f32(v0, v1, v2) == r


def test_kat_requant(kat):
    L = op.lib()
    for v, mn, sc, q, r in kat["requant_f32"]:
        assert L.ho_kat_requant_f32(v, mn, sc, q) == r


def test_kat_range_coder(kat):
    for src, code in kat["range_bytes"]:
        s, c = bytes.fromhex(src), bytes.fromhex(code)
        assert op.range_encode_bytes(s) == c
        assert op.range_decode_bytes(c, len(s)) == s
    for triples, code in kat["range_lht"]:
        assert op.range_encode_lht(np.array(triples, dtype=np.uint64)) == bytes.fromhex(code)


def test_kat_range_coder_32bit_registers(kat):
    """arith::Encoder<uint32_t> / Decoder<uint32_t> of the reference (the instantiation every stream of the chunked
    container uses): outputs produced by the reference's own headers (oracle/ref_kat.cc)."""
    for src, code in kat["range_bytes32"]:
        s, c = bytes.fromhex(src), bytes.fromhex(code)
        assert op.range_encode_bytes(s, bits=32) == c
        assert op.range_decode_bytes(c, len(s), bits=32) == s
    for triples, code in kat["range_lht32"]:
        assert op.range_encode_lht(np.array(triples, dtype=np.uint64), bits=32) == bytes.fromhex(code)


# ---------------------------------------------------------------- live reference (where its binary exists)
@pytest.mark.skipif(not os.path.exists(util.REF_BIN), reason="reference binary not built (oracle/_ref)")
@pytest.mark.parametrize("seed", range(4))
def test_live_reference_random_meshes(tmp_path, seed):
    rng = np.random.default_rng(100 + seed)
    nu, nv = int(rng.integers(6, 40)), int(rng.integers(3, 20)) * 2
    polys = ["tri", "quad", "mixed"][seed % 3]
    m = mg.torus(nu, nv, seed=seed, polys=polys, normals=bool(seed & 1))
    if seed == 3:
        m = mg.with_nonmanifold(mg.concat([m, mg.torus(7, 8, seed=9, polys=polys, normals=bool(seed & 1), center=(5, 0, 0))]), 4, 3)
    ply = m.to_ply()
    p = tmp_path / "in.ply"
    p.write_bytes(ply)
    for flags in ([], ["-l1", f"-q{int(rng.integers(2, 17))}"]):
        out = tmp_path / "out.hry"
        subprocess.run([util.REF_BIN, str(p), str(out)] + flags, check=True, capture_output=True)
        quant, clear = util.flags_to_quant(flags)
        _, res = op.encode_ply(ply, quant, clear)
        assert res.data == out.read_bytes(), (seed, flags)
