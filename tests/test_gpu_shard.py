"""One mesh -> N shards -> one container (row e of SURVEY.md section 8), run as N VIRTUAL ranks one after the other on the
one GPU of the test box: plan, extract, per-shard bounds + combination, per-shard encode (one segment each), merge, decode
(whole, and segment by segment).  The pin is the reference-format decode of the WHOLE mesh by the oracle: same arrays, same
numbering (cbm/decoder.h:48,75,145,162)."""
import numpy as np
import pytest

from harry_amd import codec as hc
from harry_amd import meshgen as mg
from harry_amd import sharding
from oracle import oracle_py as op   # checker only
from tests import util

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cx():
    c = hc.Codec(0)
    yield c
    c.close()


def _mesh(kind):
    if kind == "mixed_nm":
        return mg.with_nonmanifold(mg.multi_component(9, 13, 15, seed=4, polys="mixed"), 9, 5, seed=3)
    if kind == "tori_normals":
        return mg.concat([mg.torus(14 + 2 * i, 12 + i, seed=20 + i, normals=True, center=(3.0 * i, 0, 0)) for i in range(7)])
    if kind == "faceprops":
        return mg.with_face_props(mg.multi_component(6, 9, 11, seed=9, polys="tri"))
    raise ValueError(kind)


def plan_extract_again(gen, n_shards, s):
    """a fresh, unquantised copy of shard s (the encode mutated the one that was coded)"""
    whole = hc.Mesh.from_arrays(gen.verts, gen.degrees, gen.indices, gen.face_props)
    return hc.ShardPlan(whole, n_shards).extract(whole, s)


def _encode_sharded(cx, gen, n_shards, quant, chunk_syms=0):
    whole = hc.Mesh.from_arrays(gen.verts, gen.degrees, gen.indices, gen.face_props)
    plan = hc.ShardPlan(whole, n_shards)
    shards = [plan.extract(whole, s) for s in range(n_shards)]
    for sh in shards:
        cx.upload(sh)
    tabs = [sharding.shard_bounds(cx, sh) for sh in shards]
    parts = []
    for sh in shards:
        sharding.combine_bounds(tabs, sh)
        if quant:
            cx.requant(sh, quant)
        parts.append(cx.write_hry(sh, profile=hc.PROFILE_CHUNKED, chunk_syms=chunk_syms))
    return whole, shards, parts


@pytest.mark.parametrize("kind,quant", [("mixed_nm", []), ("mixed_nm", [(1, -1, 12)]), ("tori_normals", [(1, 0, 14), (1, 1, 14), (1, 2, 14), (1, 3, 10), (1, 4, 10), (1, 5, 10)]),
                                        ("faceprops", [])])
@pytest.mark.parametrize("n_shards", [2, 8])
def test_virtual_ranks_merge_decodes_like_the_reference(cx, kind, quant, n_shards):
    gen = _mesh(kind)
    whole, shards, parts = _encode_sharded(cx, gen, n_shards, quant, chunk_syms=1024)
    merged = hc.merge(parts)
    # the pin: the oracle's reference-format encode + decode of the whole mesh
    o = op.Mesh.from_ply(gen.to_ply())
    o0 = o.clone()                                           # unquantised: bounds and formats of the whole mesh
    if quant:
        o.requant(quant)
    ref = op.Mesh.from_hry(o.encode().data)
    # header of the merged container = header of the whole mesh (bounds combined from the shards' bounds)
    one = hc.Mesh.from_arrays(gen.verts, gen.degrees, gen.indices, gen.face_props)
    if quant:
        cx.requant(one, quant)
    single = cx.write_hry(one, profile=hc.PROFILE_CHUNKED, chunk_syms=1024)
    # every rank's segment is byte-identical to the oracle's restatement of the shard container, and the oracle decodes the
    # merged container (independent CPU decoder) to the reference's arrays
    for sh, part in zip(shards, parts):
        if sh.nf == 0:
            continue
        os_ = util.oracle_shard(plan_extract_again(gen, n_shards, shards.index(sh)), o0)
        if quant:
            os_.requant(quant)
        assert part == os_.encode_chunked(1024).data
    odec = op.Mesh.from_hry_chunked(merged)
    assert np.array_equal(odec.org(), ref.org()) and np.array_equal(odec.list_data(1), ref.list_data(1)) and np.array_equal(odec.list_data(0), ref.list_data(0))
    dec = cx.read_hry(merged)
    assert dec.nv == ref.nv and dec.nf == ref.nf
    assert np.array_equal(dec.face_offsets(), ref.face_offsets())
    assert np.array_equal(dec.org(), ref.org())
    assert np.array_equal(dec.twin(), ref.twin())
    for l in (0, 1):
        assert np.array_equal(dec.list_data(l), ref.list_data(l)), f"list {l}"
    # one GPU or N: the same mesh as the single-GPU container of the whole mesh
    dec1 = cx.read_hry(single)
    assert np.array_equal(dec1.org(), dec.org()) and np.array_equal(dec1.list_data(1), dec.list_data(1))
    # segment by segment, as N processes would: every share fills exactly its runs
    seen_f = np.zeros(dec.nf, bool)
    for r in range(n_shards):
        part = cx.read_hry(merged, shard=(r, n_shards))
        runs = part.runs()
        want = shards[r].runs()
        assert sorted(map(tuple, runs)) == sorted(map(tuple, want))
        fo, fo_ref = part.face_offsets(), ref.face_offsets()
        pv, porg, ptw, rv, rorg, rtw = part.list_data(1), part.org(), part.twin(), ref.list_data(1), ref.org(), ref.twin()
        for fv, ff, fh, nv, nf, nh in runs:
            assert np.array_equal(pv[fv:fv + nv], rv[fv:fv + nv])
            assert np.array_equal(porg[fh:fh + nh], rorg[fh:fh + nh])
            assert np.array_equal(ptw[fh:fh + nh], rtw[fh:fh + nh])
            assert np.array_equal(fo[ff:ff + nf + 1], fo_ref[ff:ff + nf + 1])
            assert not seen_f[ff:ff + nf].any()
            seen_f[ff:ff + nf] = True
        # a rank's own one-segment container decodes the same way
        if len(want):
            own = cx.read_hry(parts[r]).list_data(1)
            for fv, ff, fh, nv, nf, nh in want:
                assert np.array_equal(own[fv:fv + nv], rv[fv:fv + nv])
    assert seen_f.all()


def test_shard_bounds_combine_to_the_whole_meshs_bounds(cx):
    # all-negative coordinates (max stays FLT_MIN, quant.h:33) and signed zeros across shards
    gen = mg.negated(mg.multi_component(5, 8, 9, seed=12, polys="tri"))
    v = gen.verts.copy()
    v["z"][:] = np.float32(-0.0)
    v["z"][gen.nv // 2:] = np.float32(0.0)
    gen = mg.Mesh(v, gen.degrees, gen.indices, None)
    whole = hc.Mesh.from_arrays(gen.verts, gen.degrees, gen.indices)
    cx.bounds(whole)
    plan = hc.ShardPlan(whole, 3)
    shards = [plan.extract(whole, s) for s in range(3)]
    tabs = [sharding.shard_bounds(cx, sh) for sh in shards]
    for sh in shards:
        sharding.combine_bounds(tabs, sh)
        assert bytes(sh.list_min(1)) == bytes(whole.list_min(1))
        assert bytes(sh.list_max(1)) == bytes(whole.list_max(1))


def test_a_shard_does_not_code_into_the_reference_stream(cx):
    gen = _mesh("mixed_nm")
    whole = hc.Mesh.from_arrays(gen.verts, gen.degrees, gen.indices)
    sh = hc.ShardPlan(whole, 2).extract(whole, 0)
    with pytest.raises(hc.HryError) as e:
        cx.write_hry(sh, profile=hc.PROFILE_COMPAT)
    assert e.value.code == -3


def test_merge_checks_headers(cx):
    a = _encode_sharded(cx, _mesh("mixed_nm"), 2, [])[2]
    b = _encode_sharded(cx, _mesh("faceprops"), 2, [])[2]
    with pytest.raises(hc.HryError):
        hc.merge([a[0], b[1]])
    with pytest.raises(hc.HryError):
        cx.read_hry(hc.merge(a)[:-7])
