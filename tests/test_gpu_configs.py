"""Every configuration BASELINE.json names, at its full size, against the oracle (SURVEY.md section 8d for the synthetic
stand-ins: no mesh files exist offline).  configs[1] lives in test_gpu_chunked.py::test_headline_workload_full_size.

  configs[0]  bunny-class ~80 k triangles, lossless round trip (the reference's own CPU-runnable case)
  configs[2]  Lucy-class 28 M triangles, positions 14 bits + normals 10 bits, one GPU
  configs[3]  non-manifold polygon mesh, lossless, sharded by connected component: one GPU's share of the 100 M-triangle
              mesh (128 components, 12.6 M triangles) on one GPU, and the same mesh through the real sharded path as 8
              virtual ranks (one after the other on this GPU) -> one merged container
  configs[4]  decode-only of the merged container: whole, and segment by segment as 8 processes would
The pin is always the oracle's REFERENCE-FORMAT encode + decode of the same input (the oracle is byte-pinned to the
reference binary on the committed fixtures); the chunked containers are additionally compared byte for byte with the
oracle's restatement of the container."""
import os
import time

import numpy as np
import pytest

from harry_amd import codec as hc
from harry_amd import meshgen as mg
from harry_amd import sharding
from oracle import oracle_py as op   # checker only
from tests import util

pytestmark = pytest.mark.gpu

GOLD = os.path.join(util.ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def cx():
    c = hc.Codec(0)
    yield c
    c.close()


def same_mesh(a, b, twins=True):
    assert (a.nv, a.nf, a.ne) == (b.nv, b.nf, b.ne)
    assert np.array_equal(a.face_offsets(), b.face_offsets())
    assert np.array_equal(a.org(), b.org())
    if twins:
        assert np.array_equal(a.twin(), b.twin())
    for l in range(2):
        assert a.list_fmt(l) == b.list_fmt(l)
        assert np.array_equal(a.list_data(l), b.list_data(l)), f"list {l} differs"


def _need_memory(gib):
    try:
        import psutil
        free = psutil.virtual_memory().available / 2**30
    except Exception:
        return
    if free < gib:
        pytest.skip(f"needs about {gib} GiB of host memory for the CPU oracle at this size, {free:.0f} GiB available")


def test_cfg0_bunny_class_lossless_roundtrip(cx):
    """configs[0]: icosphere level 6 with radial noise + confidence / intensity (81 920 triangles), lossless.
    compat bytes == oracle (== reference); decode == the oracle's decode; chunked container == the oracle's."""
    mesh = mg.cfg1_bunny_class()
    ply = mesh.to_ply()
    a, o = hc.Mesh.from_ply(ply), op.Mesh.from_ply(ply)
    ref_bytes = o.clone().encode().data
    assert cx.write_hry(a.clone(), profile=hc.PROFILE_COMPAT) == ref_bytes
    ref_dec = op.Mesh.from_hry(ref_bytes)
    same_mesh(cx.read_hry(ref_bytes), ref_dec, twins=False)
    got = cx.write_hry(a.clone(), profile=hc.PROFILE_CHUNKED)
    assert got == o.clone().encode_chunked(0).data
    same_mesh(cx.read_hry(got), ref_dec, twins=False)
    # lossless: the decoded records are the input records (as a multiset: the codec renumbers vertices)
    key = lambda rec: np.sort(np.ascontiguousarray(rec).view([("", rec.dtype)] * rec.shape[1]).reshape(-1), axis=0)
    assert np.array_equal(key(a.list_data(1)), key(ref_dec.list_data(1)))


with open(os.path.join(GOLD, "manifest.json")) as _f:
    import json as _json
    REQUANT = sorted(_json.load(_f)["requant_of_hry"].items())


@pytest.mark.parametrize("name,e", REQUANT, ids=[n for n, _ in REQUANT])
def test_requant_of_a_quantised_file_matches_reference_golden(cx, name, e):
    """`harry q.hry out.hry [-c] [-l L -a A -q Q]` on a file the reference quantised: q -> q' (structs/quant.h:121-129,169-171),
    dequantisation into the original type for float / double / integer components (:180-212, `-c`: main.cc:44,108) and
    mixtures of both.  decode -> hry_requant on the device -> compat encode must give the bytes the reference wrote."""
    src = open(os.path.join(GOLD, e["src"]), "rb").read()
    want = open(os.path.join(GOLD, name + ".hry"), "rb").read()
    quant, clear = util.flags_to_quant(e["flags"])
    m = cx.read_hry(src)
    cx.requant(m, quant, clear)
    assert cx.write_hry(m, profile=hc.PROFILE_COMPAT) == want
    # and the oracle's view of the same records
    o = op.Mesh.from_hry(src)
    o.requant(quant, clear)
    assert np.array_equal(m.list_data(1), o.list_data(1)) and np.array_equal(m.list_data(0), o.list_data(0))


def test_ply_of_a_quantised_mesh_packed_and_dequantised(cx):
    """row f4: the reference's binary PLY of a quantised mesh announces the storage types but dumps original-width records
    (formats/ply/writer.cc:72-75,168).  HRY_PLY_PACKED writes what the header says; `-c` first gives the original types back."""
    src = open(os.path.join(GOLD, "colors_normals.posnrm.hry"), "rb").read()
    m = cx.read_hry(src)
    packed = hc.Mesh.from_ply(m.to_ply(packed=True))        # a well-formed PLY: our own reader takes it by its header
    assert [t for t, _q, _o in packed.list_fmt(1)][:6] == [6, 6, 6, 6, 6, 6]       # ushort positions and normals
    for c in range(len(m.list_fmt(1))):
        assert np.array_equal(packed.component(1, c), m.component(1, c))
    assert np.array_equal(packed.org(), m.org())
    ascii_ = hc.Mesh.from_ply(m.to_ply(ascii=True))
    for c in range(len(m.list_fmt(1))):
        assert np.array_equal(ascii_.component(1, c), m.component(1, c))
    cx.requant(m, [], clear=True)                            # -c
    o = op.Mesh.from_hry(src)
    o.requant([], True)
    back = hc.Mesh.from_ply(m.to_ply())
    assert [t for t, _q, _o in back.list_fmt(1)][:6] == [0] * 6 and np.array_equal(back.list_data(1), o.list_data(1))


def test_cfg2_lucy_class_28m_full_size(cx):
    """configs[2] at full size: torus 3742 x 3742 = 28 005 128 triangles with analytic normals; positions 14 bits, normals 10
    (`-l1 -a0 -q14 -a1 -q14 -a2 -q14 -a3 -q10 -a4 -q10 -a5 -q10`: SURVEY 8d flag caveat).  The chunked container equals
    the oracle's, and its GPU decode equals the oracle's reference-format decode array for array."""
    _need_memory(40)
    t0 = time.time()
    mesh = mg.cfg3_lucy_class()
    quant = [(1, 0, 14), (1, 1, 14), (1, 2, 14), (1, 3, 10), (1, 4, 10), (1, 5, 10)]
    a = hc.Mesh.from_arrays(mesh.verts, mesh.degrees, mesh.indices)
    cx.requant(a, quant)
    got = cx.write_hry(a.clone(), profile=hc.PROFILE_CHUNKED)
    dec = cx.read_hry(got)
    print(f"[cfg2] GPU side done {time.time() - t0:.0f} s", flush=True)
    o = op.Mesh.from_ply(mesh.to_ply())
    del mesh
    o.requant(quant)
    assert np.array_equal(o.list_data(1), a.list_data(1)), "quantised integers differ from the CPU reference path"
    print(f"[cfg2] oracle mesh + quantisation {time.time() - t0:.0f} s", flush=True)
    info = hc.container_info(got)       # larger meshes get larger default chunks: tell the oracle which
    assert info["minor"] == 2 and info["nf"] == a.nf
    want = o.clone().encode_chunked(info["chunk_syms"]).data
    print(f"[cfg2] oracle chunked container ({info['chunk_syms']} symbols per chunk) {time.time() - t0:.0f} s", flush=True)
    assert len(got) == len(want) and got == want
    del want
    ref_dec = op.Mesh.from_hry(o.encode().data)
    print(f"[cfg2] oracle reference-format encode + decode {time.time() - t0:.0f} s", flush=True)
    same_mesh(dec, ref_dec)


@pytest.fixture(scope="module")
def cfg4_share():
    """one GPU's share of configs[3]: 128 components of 221 x 222 mixed polygons, 0.1 % extra faces on existing edges, 0.05 %
    cones on existing vertices (both become tiny components tied to a large one through shared vertices)"""
    mesh = mg.multi_component(128, 221, 222, seed=4, polys="mixed")
    mesh = mg.with_nonmanifold(mesh, n_edges=max(1, mesh.ntri // 1000), n_vtx=max(1, mesh.ntri // 2000))
    o = op.Mesh.from_ply(mesh.to_ply())
    ref_dec = op.Mesh.from_hry(o.clone().encode().data)
    return mesh, o, ref_dec


def test_cfg3_share_one_gpu_lossless(cx, cfg4_share):
    """12.6 M triangles, lossless float: container == the oracle's, decode == the oracle's reference-format decode"""
    _need_memory(24)
    mesh, o, ref_dec = cfg4_share
    a = hc.Mesh.from_arrays(mesh.verts, mesh.degrees, mesh.indices)
    got = cx.write_hry(a, profile=hc.PROFILE_CHUNKED)
    assert got == o.clone().encode_chunked(hc.container_info(got)["chunk_syms"]).data
    same_mesh(cx.read_hry(got), ref_dec)


def test_cfg3_share_chains_in_batches_beside_the_replay(cx, cfg4_share, monkeypatch):
    """The float chains of a mesh of many components start beside the replay, batch by batch, once the mesh is large enough for
    that to pay (unchunk.cpp: ChainBatches; 20 M vertices by default).  Forced onto the 12.6 M-triangle share: every batch's
    components wait for owners in earlier batches through the same progress words, the last batch follows the replay, the
    first batches' records come down early -- the decoded mesh is the reference-format decode's, twice in a row on one context."""
    _need_memory(24)
    mesh, o, ref_dec = cfg4_share
    a = hc.Mesh.from_arrays(mesh.verts, mesh.degrees, mesh.indices)
    got = cx.write_hry(a, profile=hc.PROFILE_CHUNKED)
    monkeypatch.setenv("HRY_CHAIN_BATCH_MIN_VERTICES", "1")
    for _ in range(2):
        same_mesh(cx.read_hry(got), ref_dec)


def test_cfg3_cfg4_sharded_as_8_virtual_ranks(cx, cfg4_share):
    """The same mesh through the sharded path: plan -> 8 shards -> per-shard bounds + combination -> 8 segments -> ONE
    container; decode-only (configs[4]) of that container whole and segment by segment == reference-format decode."""
    _need_memory(24)
    mesh, o, ref_dec = cfg4_share
    whole = hc.Mesh.from_arrays(mesh.verts, mesh.degrees, mesh.indices)
    n = 8
    plan = hc.ShardPlan(whole, n)
    loads = [plan.triangles(r) for r in range(n)]
    assert sum(loads) == whole.ntri and max(loads) <= 1.05 * min(loads), loads
    shards = [plan.extract(whole, r) for r in range(n)]
    tabs = [sharding.shard_bounds(cx, sh) for sh in shards]
    parts = []
    for sh in shards:
        sharding.combine_bounds(tabs, sh)
        assert bytes(sh.list_min(1)) == bytes(o.list_min(1)) and bytes(sh.list_max(1)) == bytes(o.list_max(1))
        parts.append(cx.write_hry(sh, profile=hc.PROFILE_CHUNKED))
    merged = hc.merge(parts)
    same_mesh(cx.read_hry(merged), ref_dec)
    covered = 0
    ref_v, ref_org = ref_dec.list_data(1), ref_dec.org()
    for r in range(n):
        part = cx.read_hry(merged, shard=(r, n))
        pv, porg = part.list_data(1), part.org()
        mask_v, mask_h = np.zeros(len(pv), bool), np.zeros(len(porg), bool)
        for fv, ff, fh, nv, nf, nh in part.runs():
            mask_v[fv:fv + nv] = True
            mask_h[fh:fh + nh] = True
            covered += int(nf)
        assert np.array_equal(pv[mask_v], ref_v[mask_v]) and np.array_equal(porg[mask_h], ref_org[mask_h])
    assert covered == ref_dec.nf


def test_damaged_large_container_while_spans_upload(cx, cfg4_share):
    """the share's container with bytes flipped behind its directory: the parallel replay (spans on host threads, finished spans
    going to the device beside it) ends in an error or some mesh -- never a crash or a hang -- and the context decodes the intact
    container afterwards; the environment switch for the comparison gives the same mesh"""
    _need_memory(24)
    mesh, o, ref_dec = cfg4_share
    a = hc.Mesh.from_arrays(mesh.verts, mesh.degrees, mesh.indices)
    good = cx.write_hry(a, profile=hc.PROFILE_CHUNKED)
    info = hc.container_info(good)
    rng = np.random.default_rng(23)
    outcomes = {"error": 0, "mesh": 0}
    for trial in range(6):
        bad = bytearray(good)
        # the connectivity streams come first in the body: damage lands in the replay's input
        for _ in range(1 + trial % 3):
            k = int(rng.integers(info["header_bytes"] + 4096, info["header_bytes"] + (len(bad) - info["header_bytes"]) // 8))
            bad[k] ^= int(rng.integers(1, 256))
        try:
            cx.read_hry(bytes(bad))
            outcomes["mesh"] += 1
        except hc.HryError:
            outcomes["error"] += 1
    assert outcomes["error"] + outcomes["mesh"] == 6 and outcomes["error"] > 0
    same_mesh(cx.read_hry(good), ref_dec)
    os.environ["HRY_NO_SPAN_UPLOAD"] = "1"
    try:
        same_mesh(cx.read_hry(good), ref_dec)
    finally:
        del os.environ["HRY_NO_SPAN_UPLOAD"]


@pytest.mark.timeout(900)
def test_cfg3_cfg4_named_size_100m_one_context_and_eight(cx):
    """configs[3] / [4] at the size BASELINE names: 1 024 mixed-polygon components + 150 000 non-manifold slivers, 100.6 M triangles,
    float32 xyz, lossless.  One context: the chunked container decodes to what the oracle decodes from ITS reference-format stream
    of the same mesh (the oracle is byte-pinned to the reference binary on the committed fixtures).  Eight contexts on this one
    device through the in-process executor (shards coded where they lie in the whole mesh): the merged container equals the eight
    virtual ranks' (extracted sub-meshes, one after the other on one context) byte for byte and decodes -- whole, on eight
    contexts -- to the same mesh.  About three minutes, two of them the oracle on one core."""
    _need_memory(96)
    gen = mg.multi_component(1024, 221, 222, seed=4, polys="mixed")
    gen = mg.with_nonmanifold(gen, n_edges=max(1, gen.ntri // 1000), n_vtx=max(1, gen.ntri // 2000))
    assert gen.ntri > 100_000_000
    m0 = hc.Mesh.from_arrays(gen.verts, gen.degrees, gen.indices)
    cx.upload(m0)                                                    # twin matching on the device; every clone has its twins
    one = cx.write_hry(m0.clone(), profile=hc.PROFILE_CHUNKED, as_buffer=True)
    dec = cx.read_hry(one)
    mc = hc.MultiCodec([0] * 8)
    try:
        merged = mc.write_hry(m0.clone(), as_buffer=True)
        assert mc.last["n_segments"] == 8 and mc.last["n_components"] > 150_000
        mdec = mc.read_hry(merged)
    finally:
        mc.close()
    same_mesh(mdec, dec)
    del mdec
    # the same shards as sub-meshes of their own, one after the other on ONE context (what eight ranks would each do)
    whole = m0.clone()
    plan = hc.ShardPlan(whole, 8)
    parts = []
    tabs = []
    shards = [plan.extract(whole, s) for s in range(8)]
    tabs = [sharding.shard_bounds(cx, sh) for sh in shards]
    for sh in shards:
        sharding.combine_bounds(tabs, sh)
        parts.append(cx.write_hry(sh, profile=hc.PROFILE_CHUNKED))
    assert hc.merge(parts) == merged
    del shards, parts, whole, plan, merged
    # the oracle: reference-format stream of the same PLY, decoded by the oracle
    o = op.Mesh.from_ply(gen.to_ply())
    ref = op.Mesh.from_hry(o.encode().data)
    assert np.array_equal(dec.face_offsets(), ref.face_offsets())
    assert np.array_equal(dec.org(), ref.org())
    assert np.array_equal(dec.list_data(1), ref.list_data(1))
