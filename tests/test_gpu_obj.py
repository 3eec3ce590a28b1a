"""SURVEY.md section 8 row f3 on the GPU: OBJ input, regions, records shared between elements (global / per-vertex history), corner
attributes.  The pin is the UNMODIFIED reference binary through tests/golden/obj/ (tests/golden/make_golden_obj.py): the encoder
must write the reference's bytes, the decoder must give the reference's decoded OBJ / PLY text; beyond the fixtures the CPU oracle
(pinned to the same fixtures, tests/test_oracle_obj.py) checks larger scenes.
Reference: formats/obj/reader.rl:108-299, writer.cc:20-132, formats/hry/attrcode.h:23-80,135-154,321-393,443-531."""
import hashlib
import json
import os

import numpy as np
import pytest

from harry_amd import codec as hc
from harry_amd import meshgen as mg
from harry_amd import objgen as og
from oracle import oracle_py as op   # checker only
from tests import util

pytestmark = pytest.mark.gpu

OBJ = os.path.join(util.ROOT, "tests", "golden", "obj")
GOLD = os.path.join(util.ROOT, "tests", "golden")
with open(os.path.join(OBJ, "manifest.json")) as _f:
    MAN = json.load(_f)
SMALL = [(n, t) for n, e in sorted(MAN["small"].items()) for t in sorted(e["variants"])]


@pytest.fixture(scope="module")
def cx():
    c = hc.Codec(0)
    yield c
    c.close()


def _read(name, root=OBJ):
    with open(os.path.join(root, name), "rb") as f:
        return f.read()


def same_decoded(a, o):
    """a: harry_amd decode, o: oracle decode of the same bytes"""
    assert a.general == o.general and (a.nv, a.nf, a.ne, a.nlists) == (o.nv, o.nf, o.ne, o.nlists)
    assert np.array_equal(a.face_offsets(), o.face_offsets()) and np.array_equal(a.org(), o.org()) and np.array_equal(a.twin(), o.twin())
    for which in (0, 1):
        assert np.array_equal(a.regions_of(which), o.regions_of(which))
    for kind in (0, 1, 2):
        assert np.array_equal(a.bindings(kind), o.bindings(kind)), kind
    for l in range(a.nlists):
        assert a.list_target(l) == o.list_target(l)
        if a.list_target(l) == 3:
            continue
        assert a.list_fmt(l) == o.list_fmt(l) and np.array_equal(a.list_data(l), o.list_data(l)), f"list {l} differs"


@pytest.mark.parametrize("name,tag", SMALL, ids=[f"{n}.{t}" for n, t in SMALL])
def test_obj_to_hry_is_byte_identical_to_the_reference(cx, name, tag):
    m = hc.Mesh.from_obj(_read(name + ".obj"), OBJ)
    quant, clear = util.flags_to_quant(MAN["small"][name]["variants"][tag]["flags"])
    if quant or clear:
        cx.requant(m, quant, clear)
    assert cx.write_hry(m, profile=hc.PROFILE_COMPAT) == _read(f"{name}.{tag}.hry")


@pytest.mark.parametrize("name,tag", SMALL, ids=[f"{n}.{t}" for n, t in SMALL])
def test_reference_hry_decodes_to_the_reference_obj_and_ply(cx, name, tag):
    ref = _read(f"{name}.{tag}.hry")
    d = cx.read_hry(ref)
    same_decoded(d, op.Mesh.from_hry(ref))
    assert d.to_obj() == _read(f"{name}.{tag}.dec.obj")
    assert d.to_ply(ascii=True) == _read(f"{name}.{tag}.dec.ply")
    assert cx.write_hry(d, profile=hc.PROFILE_COMPAT) == op.Mesh.from_hry(ref).encode().data   # and codes again like the oracle


@pytest.mark.parametrize("name,tag", SMALL, ids=[f"{n}.{t}" for n, t in SMALL])
def test_generic_vertex_chain_gives_the_same_records(cx, name, tag, monkeypatch):
    """vertex lists normally take the reconstruction chains of the PLY layout when every vertex owns a record of one list; the
    relaxation chain of the general path (several vertex regions, shared vertex records) must give the same records"""
    ref = _read(f"{name}.{tag}.hry")
    monkeypatch.setenv("HRY_GENERIC_VERTEX", "1")
    same_decoded(cx.read_hry(ref), op.Mesh.from_hry(ref))


@pytest.mark.parametrize("name", sorted(MAN["requant_of_hry"]))
def test_requant_of_an_obj_hry_matches_reference_golden(cx, name):
    e = MAN["requant_of_hry"][name]
    m = cx.read_hry(_read(e["src"]))
    quant, clear = util.flags_to_quant(e["flags"])
    cx.requant(m, quant, clear)
    out = cx.write_hry(m, profile=hc.PROFILE_COMPAT)
    assert out == _read(name + ".hry")
    assert cx.read_hry(out).to_obj() == _read(name + ".dec.obj")


@pytest.mark.parametrize("src", sorted(MAN["ply_to_obj"]))
def test_ply_layout_writes_the_reference_obj(cx, src):
    assert cx.read_hry(_read(src, GOLD)).to_obj() == _read(src[:-4] + ".dec.obj")


@pytest.mark.parametrize("name", sorted(MAN["big"]))
def test_big_scene_matches_reference_hash_and_oracle_decode(cx, name):
    e = MAN["big"][name]
    sc = {"torus150": lambda: og.scene(mg.torus(150, 150, seed=2), normals="smooth", tex="atlas", charts=7),
          "flat_ico5": lambda: og.scene(mg.icosphere(5), normals="flat", tex="corner")}[name]()
    for tag, v in e["variants"].items():
        m = hc.Mesh.from_obj(sc.obj, "")
        quant, clear = util.flags_to_quant(v["flags"])
        if quant or clear:
            cx.requant(m, quant, clear)
        got = cx.write_hry(m, profile=hc.PROFILE_COMPAT)
        assert len(got) == v["hry_bytes"] and hashlib.sha256(got).hexdigest() == v["hry_sha256"]
        d = cx.read_hry(got)
        same_decoded(d, op.Mesh.from_hry(got))
        assert hashlib.sha256(d.to_obj()).hexdigest() == v["dec_obj_sha256"]


@pytest.mark.parametrize("name,tag", SMALL, ids=[f"{n}.{t}" for n, t in SMALL])
@pytest.mark.parametrize("chunk", [0, 256])
def test_chunked_container_of_general_bindings(cx, name, tag, chunk, monkeypatch):
    """the parallel container (.hry v0.2) with general bindings: byte-identical to the oracle's restatement, and its GPU decode
    equals the reference-format decode of the same mesh (bindings and records), through both vertex paths"""
    m = hc.Mesh.from_obj(_read(name + ".obj"), OBJ)
    o = op.Mesh.from_obj(_read(name + ".obj"), OBJ)
    quant, clear = util.flags_to_quant(MAN["small"][name]["variants"][tag]["flags"])
    if quant or clear:
        cx.requant(m, quant, clear)
        o.requant(quant, clear)
    got = cx.write_hry(m, profile=hc.PROFILE_CHUNKED, chunk_syms=chunk)
    info = hc.container_info(got)
    assert info["minor"] == 2
    assert got == o.clone().encode_chunked(info["chunk_syms"]).data
    ref = op.Mesh.from_hry(_read(f"{name}.{tag}.hry"))
    same_decoded(cx.read_hry(got), ref)
    monkeypatch.setenv("HRY_GENERIC_VERTEX", "1")
    same_decoded(cx.read_hry(got), ref)


def test_chunked_container_larger_scenes(cx):
    for sc, quant in ((og.scene(mg.torus(60, 64, polys="mixed"), normals="smooth", tex="atlas", charts=6, colors="some", materials=3), []),
                      (og.scene(mg.icosphere(5), normals="flat", tex="atlas", charts=9), [(0, -1, 14), (1, -1, 11), (2, -1, 10)])):
        o = op.Mesh.from_obj(sc.obj, "")
        m = hc.Mesh.from_obj(sc.obj, "")
        if quant:
            cx.requant(m, quant)
            o.requant(quant)
        got = cx.write_hry(m, profile=hc.PROFILE_CHUNKED)
        assert got == o.clone().encode_chunked(hc.container_info(got)["chunk_syms"]).data
        same_decoded(cx.read_hry(got), op.Mesh.from_hry(o.clone().encode().data))


def test_a_one_component_scene_shards_into_one_segment(cx):
    """general bindings shard like the PLY layout (below: test_obj_scene_as_virtual_ranks_...); a scene of one component is one group"""
    m = hc.Mesh.from_obj(_read("smooth.obj"), OBJ)
    plan = hc.ShardPlan(m, 2)
    assert plan.ngroups == 1
    shards = [plan.extract(m, s) for s in range(2)]
    assert sorted(sh.nf for sh in shards) == [0, m.nf]


def _disk_scene(n=28, colors_every=0):
    """a disk: centre, inner ring, outer ring; the outer band comes first in the file so that the centre is coded late, with
    more coded parallelograms (n) and more coded faces around it (n) than the decoder's source table holds per record"""
    import math
    rng = np.random.default_rng(5)
    pts = [(0.0, 0.0, 0.3)]
    for ring, rad in ((1, 1.0), (2, 2.0)):
        for i in range(n):
            a = 2 * math.pi * i / n
            pts.append((rad * math.cos(a) + 0.01 * rng.random(), rad * math.sin(a) + 0.01 * rng.random(), 0.05 * rng.random()))
    inner = lambda i: 1 + i % n
    outer = lambda i: 1 + n + i % n
    faces = []
    for i in range(n):
        faces.append((inner(i), outer(i), outer(i + 1)))
        faces.append((inner(i), outer(i + 1), inner(i + 1)))
    for i in range(n):
        faces.append((0, inner(i), inner(i + 1)))
    out = []
    for k, p in enumerate(pts):
        vals = [f"{x:.6f}" for x in p]
        if colors_every and k % colors_every == 0:
            vals += [f"{x:.6f}" for x in rng.random(3)]
        out.append("v " + " ".join(vals))
    for p in pts:
        out.append(f"vt {p[0] * 0.2 + 0.5:.6f} {p[1] * 0.2 + 0.5:.6f}")
    for k in range(len(faces)):
        out.append(f"vn {rng.random():.6f} {rng.random():.6f} {rng.random():.6f}")      # one normal per face: every face adds a record at the centre
    for k, f in enumerate(faces):
        out.append("f " + " ".join(f"{v + 1}/{v + 1}/{k + 1}" for v in f))
    return ("\n".join(out) + "\n").encode()


@pytest.mark.parametrize("colors_every", [0, 3])
@pytest.mark.parametrize("quantised", [False, True])
def test_fans_larger_than_the_source_table(cx, monkeypatch, colors_every, quantised):
    """high-valence vertex: more candidates than k_gen_sources keeps per record (the chain walks those fans itself), in both
    vertex paths; colors_every=3 gives two vertex regions (x y z and x y z r g b), which only the general chain handles"""
    data = _disk_scene(28, colors_every)
    o = op.Mesh.from_obj(data, "")
    m = hc.Mesh.from_obj(data, "")
    if quantised:
        q = [(l, -1, 9 + l) for l in range(m.nlists)]
        cx.requant(m, q)
        o.requant(q)
    want = o.clone().encode().data
    got = cx.write_hry(m, profile=hc.PROFILE_COMPAT)
    assert got == want
    ref = op.Mesh.from_hry(want)
    same_decoded(cx.read_hry(want), ref)
    monkeypatch.setenv("HRY_GENERIC_VERTEX", "1")
    same_decoded(cx.read_hry(want), ref)


def test_larger_random_scenes_against_the_oracle(cx):
    for sc, quant in ((og.scene(mg.torus(40, 44, polys="mixed"), normals="smooth", tex="atlas", charts=6, colors="some", materials=3), []),
                      (og.scene(mg.with_nonmanifold(mg.multi_component(6, 14, 15, polys="mixed"), 6, 4), normals="flat", tex="corner"), [(0, -1, 14), (1, -1, 11)]),
                      (og.scene(mg.icosphere(4), normals="flat", tex="atlas", charts=9, tex3=True), [(2, -1, 10)])):
        o = op.Mesh.from_obj(sc.obj, "")
        m = hc.Mesh.from_obj(sc.obj, "")
        if quant:
            cx.requant(m, quant)
            o.requant(quant)
        want = o.clone().encode().data
        assert cx.write_hry(m, profile=hc.PROFILE_COMPAT) == want
        same_decoded(cx.read_hry(want), op.Mesh.from_hry(want))


def test_decode_survives_damaged_payload(cx):
    """flipped bytes behind the header of an OBJ-made stream: an error or some mesh, never a crash, a hang or an unusable context"""
    good = _read("mixedfmt.ll.hry")
    ref = op.Mesh.from_hry(good)
    hdr = hc.container_info(good)["header_bytes"]
    rng = np.random.default_rng(17)
    outcomes = {"error": 0, "mesh": 0}
    for trial in range(40):
        bad = bytearray(good)
        for _ in range(1 + trial % 3):
            k = int(rng.integers(hdr, len(bad)))
            bad[k] ^= int(rng.integers(1, 256))
        try:
            cx.read_hry(bytes(bad)).to_obj()
            outcomes["mesh"] += 1
        except hc.HryError:
            outcomes["error"] += 1
    assert outcomes["error"] + outcomes["mesh"] == 40 and outcomes["error"] > 0
    try:
        cx.read_hry(good[:hdr + 10])     # a cut stream reads ones past its end (arith/bitstream.h:27): usually a format error
    except hc.HryError:
        pass
    same_decoded(cx.read_hry(good), ref)


# ---- one OBJ scene over N shards (general bindings in the sharded container, .hry v0.3) -------------------------------------------
def _sharded_general(cx, obj, n_shards, quant, chunk=1024):
    from harry_amd import sharding
    whole = hc.Mesh.from_obj(obj, "")
    plan = hc.ShardPlan(whole, n_shards)
    shards = [plan.extract(whole, s) for s in range(n_shards)]
    tabs = [sharding.shard_bounds(cx, sh) for sh in shards]
    parts = []
    for sh in shards:
        sharding.combine_bounds(tabs, sh)
        if quant:
            cx.requant(sh, quant)
        parts.append(cx.write_hry(sh, profile=hc.PROFILE_CHUNKED, chunk_syms=chunk))
    return plan, shards, parts


_SHARED_NORMAL_SCENE = b"""# three parts; the first two use the same vn line (they must stay in one group), the third has its own
v 0 0 0
v 1 0 0
v 1 1 0
v 0 1 0
v 3 0 0
v 4 0 0
v 4 1 0
v 3 1 0.5
v 6 0 0
v 7 0 1
v 7 1 0
vn 0 0 1
vn 0.6 0 0.8
vt 0 0
vt 1 0
vt 1 1
vt 0 1
f 1/1/1 2/2/1 3/3/1 4/4/1
f 5/1/1 6/2/1 7/3/1
f 5/1/1 7/3/1 8/4/1
f 9/1/2 10/2/2 11/3/2
"""


@pytest.mark.parametrize("n_shards", [2, 8])
@pytest.mark.parametrize("kw,quant", [(dict(normals="smooth", tex="atlas", charts=3), []),
                                      (dict(normals="flat", tex="corner", colors="some", materials=2, mtl_name="none.mtl"), [(0, -1, 12)]),
                                      (dict(normals="smooth"), [(l, -1, 11) for l in range(2)])])
def test_obj_scene_as_virtual_ranks_decodes_like_the_reference_format(cx, n_shards, kw, quant):
    """plan (components tied by shared vertices AND shared records), extract (lists, regions, bindings of the shard), bounds of every
    list combined from the shards', one segment per shard, merge; the merged container decodes -- on the GPU whole and on several
    contexts, and by the oracle's independent CPU decoder -- to the reference-format decode of the whole scene: same connectivity,
    same regions, same bindings, the records of every list in the same creation order (attrcode.h:443-531)"""
    base = mg.with_nonmanifold(mg.multi_component(7, 8, 9, seed=5, polys="mixed"), 3, 2, seed=4)
    sc = og.scene(base, **kw)
    plan, shards, parts = _sharded_general(cx, sc.obj, n_shards, quant)
    assert plan.ngroups >= 2
    merged = hc.merge(parts)
    assert hc.container_info(merged)["minor"] == 3 and hc.container_check(merged)
    o = op.Mesh.from_obj(sc.obj, "")
    if quant:
        o.requant(quant)
    ref = op.Mesh.from_hry(o.encode().data)                 # the reference stream of the whole scene, decoded
    same_decoded(cx.read_hry(merged), ref)
    same_decoded(cx.read_hry(merged), op.Mesh.from_hry_chunked(merged))
    mc = hc.MultiCodec([0] * min(n_shards, 4))
    try:
        assert mc.write_hry(hc.Mesh.from_obj(sc.obj, ""), quant, n_shards=n_shards, chunk_syms=1024) == merged
        same_decoded(mc.read_hry(merged), ref)
    finally:
        mc.close()
    # a share of the segments: partial, its runs hold the reference's values
    part = cx.read_hry(merged, shard=(0, 2))
    assert part.partial and len(part.runs())
    with pytest.raises(hc.HryError):
        part.to_obj()
    # text round trip of the merged container: what `harry merged.hry out.obj` writes is what the reference-format decode writes
    assert cx.read_hry(merged).to_obj() == cx.read_hry(o.encode().data).to_obj()


def test_components_that_share_a_record_stay_in_one_group(cx):
    whole = hc.Mesh.from_obj(_SHARED_NORMAL_SCENE, "")
    plan = hc.ShardPlan(whole, 3)
    assert plan.ncomponents == 3 and plan.ngroups == 1          # the shared "vt" lines tie all three; ...
    scene2 = _SHARED_NORMAL_SCENE.replace(b"f 9/1/2 10/2/2 11/3/2", b"vt 0.5 0.5\nvt 0.25 0.5\nvt 0.5 0.25\nf 9/5/2 10/6/2 11/7/2")
    whole = hc.Mesh.from_obj(scene2, "")
    plan = hc.ShardPlan(whole, 2)
    assert plan.ncomponents == 3 and plan.ngroups == 2          # ... with its own texture coordinates the third part is free
    _, shards, parts = _sharded_general(cx, scene2, 2, [])
    merged = hc.merge(parts)
    o = op.Mesh.from_obj(scene2, "")
    same_decoded(cx.read_hry(merged), op.Mesh.from_hry(o.encode().data))
    same_decoded(cx.read_hry(merged), op.Mesh.from_hry_chunked(merged))


def _with_normals(obj: bytes, values) -> bytes:
    """the scene with the components of its `vn` lines replaced, in order, by values(k) -> three numbers as text"""
    out, k = [], 0
    for line in obj.split(b"\n"):
        if line.startswith(b"vn "):
            out.append(b"vn " + " ".join(values(k)).encode())
            k += 1
        else:
            out.append(line)
    return b"\n".join(out)


_SPECIAL_NORMALS = {
    # axis-aligned: zeros of both signs and +-1 (predictions of exactly 0, equal distances everywhere)
    "axis": lambda rng: (lambda k: [["0", "-0", "1", "-1"][int(x)] for x in rng.integers(0, 4, 3)]),
    # every record changes sign against the one before it (far residual codes in every step)
    "signs": lambda rng: (lambda k: [("-" if (k + c) & 1 else "") + f"{1e-3 * (1 + (k % 7)):.6g}" for c in range(3)]),
    # denormals and values whose float sum overflows while the reference's double does not (the exact form takes the run)
    "range": lambda rng: (lambda k: [["1e-40", "-1e-40", "3e+38", "-3e+38", "0.5", "2.5e+38"][int(x)] for x in rng.integers(0, 6, 3)]),
    # one value everywhere (every code 0, every distance equal)
    "constant": lambda rng: (lambda k: ["0.25", "0.25", "0.25"]),
    # two values only: the mean of two sources lies exactly between them (ties decided by the order of the sources)
    "two": lambda rng: (lambda k: [["0.125", "0.375"][int(x)] for x in rng.integers(0, 2, 3)]),
}


@pytest.mark.parametrize("pattern", sorted(_SPECIAL_NORMALS))
@pytest.mark.parametrize("shape", ["ico", "mixed"])
def test_one_normal_per_face_with_special_values(cx, pattern, shape):
    """the short form of the corner-record step (general.hip) against the oracle where its shortcuts do not hold: zeros of both
    signs, a change of sign in every record, equal distances, float sums that overflow; runs of 2 to 6 sources"""
    base = mg.icosphere(4) if shape == "ico" else mg.with_nonmanifold(mg.multi_component(3, 20, 22, polys="mixed"), 4, 3)
    sc = og.scene(base, normals="flat", tex="corner" if shape == "ico" else None)
    text = _with_normals(sc.obj, _SPECIAL_NORMALS[pattern](np.random.default_rng(11)))
    m = hc.Mesh.from_obj(text, "")
    o = op.Mesh.from_obj(text, "")
    ref_bytes = o.clone().encode().data
    got = cx.write_hry(m.clone(), profile=hc.PROFILE_COMPAT)
    assert got == ref_bytes
    ref = op.Mesh.from_hry(ref_bytes)
    same_decoded(cx.read_hry(got), ref)
    chunked = cx.write_hry(m.clone(), profile=hc.PROFILE_CHUNKED, chunk_syms=512)
    assert chunked == o.clone().encode_chunked(512).data
    same_decoded(cx.read_hry(chunked), ref)
    # the same records as integers: 7 bits in a byte, 16 in a short, 20 and 27 in a word (sums below 2^31: the multiplication
    # form of the mean), 30 (past it: the exact form)
    for bits in (7, 16, 20, 27, 30):
        quant = [(l, -1, bits) for l in range(m.nlists) if m.list_target(l) != 3]
        mq, oq = m.clone(), o.clone()
        cx.requant(mq, quant, False)
        oq.requant(quant, False)
        ref_bytes = oq.clone().encode().data
        assert cx.write_hry(mq.clone(), profile=hc.PROFILE_COMPAT) == ref_bytes
        ref = op.Mesh.from_hry(ref_bytes)
        same_decoded(cx.read_hry(ref_bytes), ref)
        same_decoded(cx.read_hry(cx.write_hry(mq.clone(), profile=hc.PROFILE_CHUNKED, chunk_syms=512)), ref)


def test_a_resident_scene_writes_the_bytes_of_an_uploaded_one(cx):
    """hry_mesh_upload keeps a mesh with general bindings resident like one in the PLY layout (connectivity, every list, the region
    and record tables: general.cpp upload_general): encodes from the resident copy, repeated, in both profiles, after another mesh has
    taken the context, and after a decode has, all write the bytes an unprepared clone writes (which equal the oracle's)."""
    sc = og.scene(mg.with_nonmanifold(mg.multi_component(5, 14, 15, seed=21, polys="mixed"), 6, 3, seed=3), normals="flat", tex="corner", colors="some")
    other = og.scene(mg.torus(17, 13, seed=4), normals="smooth", tex="atlas", charts=3)
    m = hc.Mesh.from_obj(sc.obj, "")
    m2 = hc.Mesh.from_obj(other.obj, "")
    want_compat = cx.write_hry(m.clone(), profile=hc.PROFILE_COMPAT)
    want_chunked = cx.write_hry(m.clone(), profile=hc.PROFILE_CHUNKED, chunk_syms=512)
    assert want_compat == op.Mesh.from_obj(sc.obj, "").encode().data
    a = m.clone()
    cx.upload(a)
    assert cx.write_hry(a, profile=hc.PROFILE_CHUNKED, chunk_syms=512) == want_chunked
    assert cx.write_hry(a, profile=hc.PROFILE_COMPAT) == want_compat
    assert cx.write_hry(a, profile=hc.PROFILE_CHUNKED, chunk_syms=512) == want_chunked
    other_bytes = cx.write_hry(m2.clone(), profile=hc.PROFILE_CHUNKED)      # another mesh takes the context's arrays
    assert cx.write_hry(a, profile=hc.PROFILE_CHUNKED, chunk_syms=512) == want_chunked
    b = m2.clone()
    cx.upload(b)
    assert cx.write_hry(b, profile=hc.PROFILE_CHUNKED) == other_bytes
    d = cx.read_hry(want_chunked)                                            # a decode takes them
    assert cx.write_hry(b, profile=hc.PROFILE_CHUNKED) == other_bytes
    again = cx.write_hry(d.clone(), profile=hc.PROFILE_COMPAT)              # (the decoded scene, in its own numbering)
    assert cx.write_hry(d, profile=hc.PROFILE_COMPAT) == again              # straight from the decode's arrays
    cx.upload(d)
    assert cx.write_hry(d, profile=hc.PROFILE_COMPAT) == again
    assert cx.write_hry(a, profile=hc.PROFILE_COMPAT) == want_compat


def _cone_scene(n):
    import math
    lines = ["v 0 0 1"]
    for i in range(n):
        a = 2 * math.pi * i / n
        lines.append(f"v {math.cos(a):.6f} {math.sin(a):.6f} 0")
    for i in range(n):
        a = 2 * math.pi * (i + 0.5) / n
        lines.append(f"vn {math.cos(a) * 0.7:.6f} {math.sin(a) * 0.7:.6f} 0.7")
    for i in range(n):
        lines.append(f"f 1//{i + 1} {2 + i}//{i + 1} {2 + (i + 1) % n}//{i + 1}")
    return ("\n".join(lines) + "\n").encode()


def test_a_hub_with_more_names_than_a_device_thread_walks(cx):
    """a cone: n faces round one apex, a normal of its own for every face -- the apex gets n different records at the normals'
    corner slot.  On the device every reference walks the vertex' list of names (events.hip: k_ev_names), and a thread that passes
    16 384 of them gives the mesh to the host's loop: the container equals the oracle's either way (600 faces: the device's lists;
    20 000: the host's).  The round trip is checked on the small cone only: the decoder's chain walks such a fan record by record."""
    for n in (600, 20000):
        data = _cone_scene(n)
        m = hc.Mesh.from_obj(data, "")
        o = op.Mesh.from_obj(data, "")
        got = cx.write_hry(m, profile=hc.PROFILE_CHUNKED)
        assert got == o.clone().encode_chunked(hc.container_info(got)["chunk_syms"]).data
        if n == 600:
            same_decoded(cx.read_hry(got), op.Mesh.from_hry(o.clone().encode().data))
