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
    if kind == "soup_between_tori":   # edges shared by several faces: the walk REPAIRS twins (cbm/encoder.h:150,193-198) -- in the second of three shards
        return mg.concat([mg.torus(12, 12, seed=2), mg.soup(20, 200, 28), mg.torus(9, 11, seed=7), mg.torus(6, 9, seed=1)])
    if kind == "soups":   # ... and repairs that cut components in two: such a mesh ends up on ONE walking thread (host.hpp WalkMismatch)
        return mg.concat([mg.torus(12, 13, seed=1)] + [mg.soup(seed=s) for s in (5, 6, 13, 14, 16, 22)] + [mg.torus(10, 11, seed=2)])
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
        assert part.partial
        with pytest.raises(hc.HryError):   # only its runs are real: no consumer takes it
            part.to_ply()
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
            own = cx.read_hry(parts[r], partial=n_shards > 1 and sum(1 for q in shards if q.nf) > 1)
            assert own.partial == (sum(1 for q in shards if q.nf) > 1)
            own = own.list_data(1)
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


# ---- one process, N contexts (include/harry_amd.h: hry_encode_sharded / hry_decode_sharded) ----------------------------------
@pytest.mark.parametrize("kind,quant", [("mixed_nm", []), ("mixed_nm", [(1, -1, 12)]), ("tori_normals", [(1, 0, 14), (1, 1, 14), (1, 2, 14), (1, 3, 10), (1, 4, 10), (1, 5, 10)])])
@pytest.mark.parametrize("n_ctx", [2, 8])
@pytest.mark.parametrize("how", ["host-plan", "host-plan-turns", "device-plan", "device-plan-turns", "device-plan-small-batches"])
def test_in_process_contexts_run_concurrently_and_match_the_virtual_ranks(cx, kind, quant, n_ctx, how, monkeypatch):
    """N contexts on device 0, one worker thread each, all at once: the merged container is byte for byte what the shards give
    when they are coded one after the other on one context, and decodes (on the N contexts at once, and on one) to the
    reference-format decode of the whole mesh.  device-plan (round 5; what a large mesh takes): the whole mesh resident on the
    first context, its components analysed there, the other contexts of that device reading the first one's arrays; the walks of
    the workers all at once (or in turn), the encode's device side beside them."""
    if how.startswith("device-plan"):
        monkeypatch.setenv("HRY_DEVICE_ANALYSIS_MIN_FACES", "1")
        monkeypatch.setenv("HRY_PARALLEL_MIN_FACES", "1")
        monkeypatch.setenv("HRY_HOST_THREADS", "6")
    if how.endswith("-turns"):
        monkeypatch.setenv("HRY_SHARD_TURNS", "1")
    if how == "device-plan-small-batches":
        monkeypatch.setenv("HRY_ENCODE_PIPELINE_BATCH", "1")
    gen = _mesh(kind)
    _, _, parts = _encode_sharded(cx, gen, n_ctx, quant, chunk_syms=1024)
    want = hc.merge(parts)
    mc = hc.MultiCodec([0] * n_ctx)
    try:
        whole = hc.Mesh.from_arrays(gen.verts, gen.degrees, gen.indices, gen.face_props)
        got = mc.write_hry(whole, quant, chunk_syms=1024)
        assert got == want
        assert mc.last["n_contexts"] == n_ctx and mc.last["n_shards"] == n_ctx and mc.last["plan_ms"] > 0
        assert whole.list_min(1) is not None          # the whole mesh received its bounds (ply/reader.cc:428)
        # more shards than contexts: every worker codes several, same bytes as that many virtual ranks
        _, _, parts3 = _encode_sharded(cx, gen, n_ctx + 3, quant, chunk_syms=1024)
        assert mc.write_hry(hc.Mesh.from_arrays(gen.verts, gen.degrees, gen.indices, gen.face_props), quant, n_shards=n_ctx + 3, chunk_syms=1024) == hc.merge(parts3)
        o = op.Mesh.from_ply(gen.to_ply())
        if quant:
            o.requant(quant)
        ref = op.Mesh.from_hry(o.encode().data)
        for dec in (mc.read_hry(got), cx.read_hry(got)):
            assert not dec.partial
            assert np.array_equal(dec.face_offsets(), ref.face_offsets()) and np.array_equal(dec.org(), ref.org()) and np.array_equal(dec.twin(), ref.twin())
            for l in (0, 1):
                assert np.array_equal(dec.list_data(l), ref.list_data(l))
        assert mc.last["n_segments"] >= 1
        # an unsharded container through the same entry
        one = cx.write_hry(hc.Mesh.from_arrays(gen.verts, gen.degrees, gen.indices, gen.face_props), profile=hc.PROFILE_CHUNKED)
        assert np.array_equal(mc.read_hry(one).org(), cx.read_hry(one).org())
    finally:
        mc.close()


def test_resident_mesh_keeps_the_twins_that_contexts_of_other_devices_repaired(monkeypatch):
    """Round 6: with the plan on the first context's device the whole mesh stays resident there.  A worker on ANOTHER
    device repairs non-manifold twins in the host array and in its own copy only; the first context receives those (half-edge,
    twin) pairs after the workers, so that a later hry_encode of the resident mesh -- whose walk finds nothing left to repair --
    predicts from the twins the decoder will have.  HRY_SHARD_FOREIGN_CONTEXTS makes the contexts of this one-GPU box behave as
    contexts of other devices (own interval copies, own twins)."""
    monkeypatch.setenv("HRY_DEVICE_ANALYSIS_MIN_FACES", "1")
    monkeypatch.setenv("HRY_PARALLEL_MIN_FACES", "1")
    monkeypatch.setenv("HRY_HOST_THREADS", "6")
    monkeypatch.setenv("HRY_SHARD_FOREIGN_CONTEXTS", "1")
    gen = _mesh("soup_between_tori")
    one = hc.Codec(0)
    try:
        want_one = one.write_hry(hc.Mesh.from_arrays(gen.verts, gen.degrees, gen.indices, gen.face_props), profile=hc.PROFILE_CHUNKED, chunk_syms=1024)
        _, _, parts = _encode_sharded(one, gen, 3, [], chunk_syms=1024)
    finally:
        one.close()
    mc = hc.MultiCodec([0, 0, 0])
    try:
        whole = hc.Mesh.from_arrays(gen.verts, gen.degrees, gen.indices, gen.face_props)
        assert mc.write_hry(whole, [], chunk_syms=1024) == hc.merge(parts)
        # the same host mesh again through the first context alone: resident there, its twins repaired by the other contexts' walks
        again = mc.ctx[0].write_hry(whole, profile=hc.PROFILE_CHUNKED, chunk_syms=1024, keep_stages=True)
        assert again == want_one
        repaired = whole.twin()
        assert not np.array_equal(repaired, hc.Mesh.from_arrays(gen.verts, gen.degrees, gen.indices, gen.face_props).twin())   # (the walks did repair some)
        assert np.array_equal(mc.ctx[0].stage("d_twin", np.uint32), repaired)   # the fan walks followed the repaired twins, every one of them
        ref = op.Mesh.from_hry(op.Mesh.from_ply(gen.to_ply()).encode().data)
        dec = mc.ctx[0].read_hry(again)
        assert np.array_equal(dec.twin(), ref.twin()) and np.array_equal(dec.list_data(1), ref.list_data(1))
    finally:
        mc.close()


def test_two_threads_two_contexts_independent_meshes():
    """section 8b's threading contract: distinct contexts are independent -- two threads code and decode different meshes at once"""
    import threading
    gens = [_mesh("mixed_nm"), _mesh("tori_normals")]
    want = []
    c0 = hc.Codec(0)
    for g in gens:
        want.append(c0.write_hry(hc.Mesh.from_arrays(g.verts, g.degrees, g.indices), profile=hc.PROFILE_CHUNKED, chunk_syms=1024))
    c0.close()
    errs = []

    def work(i):
        try:
            c = hc.Codec(0)
            for _ in range(6):
                out = c.write_hry(hc.Mesh.from_arrays(gens[i].verts, gens[i].degrees, gens[i].indices), profile=hc.PROFILE_CHUNKED, chunk_syms=1024)
                assert out == want[i]
                assert c.read_hry(out).nf == gens[i].nf
                comp = c.write_hry(hc.Mesh.from_arrays(gens[i].verts, gens[i].degrees, gens[i].indices))
                assert c.read_hry(comp).nf == gens[i].nf
            c.close()
        except Exception as e:   # noqa: BLE001
            errs.append((i, repr(e)))

    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs


def test_unreferenced_vertices_enter_the_bounds_of_the_whole_mesh(cx):
    """vertices no face references are never coded, but the reference's bounds scan reads every record (structs/quant.h:30-44):
    shard 0 carries them, so the combined bounds -- and with them the header and every quantised value -- are the whole mesh's"""
    gen = mg.multi_component(5, 8, 9, seed=12, polys="tri")
    v = np.zeros(gen.nv + 3, dtype=gen.verts.dtype)
    v[:gen.nv] = gen.verts
    v["x"][gen.nv:] = [1e3, -1e3, 0.0]
    v["y"][gen.nv:] = [5.0, 5.0, -77.0]
    gen = mg.Mesh(v, gen.degrees, gen.indices, None)
    quant = [(1, -1, 12)]
    one = hc.Mesh.from_arrays(gen.verts, gen.degrees, gen.indices)
    cx.requant(one, quant)
    ref = cx.read_hry(cx.write_hry(one, profile=hc.PROFILE_CHUNKED))
    whole, shards, parts = _encode_sharded(cx, gen, 3, quant)
    assert bytes(shards[1].list_min(1)) == bytes(one.list_min(1)) and bytes(shards[1].list_max(1)) == bytes(one.list_max(1))
    dec = cx.read_hry(hc.merge(parts))
    assert np.array_equal(dec.list_data(1), ref.list_data(1)) and np.array_equal(dec.org(), ref.org())
    mc = hc.MultiCodec([0, 0, 0])
    try:
        assert mc.write_hry(hc.Mesh.from_arrays(gen.verts, gen.degrees, gen.indices), quant) == hc.merge(parts)
    finally:
        mc.close()


def _damage_cases(merged):
    info = hc.container_info(merged)
    hdr, nseg = info["header_bytes"], info["segments"]
    lens = np.frombuffer(merged, "<u8", nseg, hdr + 4)
    seg0 = hdr + 4 + 8 * nseg
    b = bytearray(merged)
    cases = {}
    cases["truncated segment table"] = bytes(b[:hdr + 4 + 5])
    cases["truncated body"] = bytes(b[:-9])
    z = bytearray(b); z[hdr:hdr + 4] = (0).to_bytes(4, "little"); cases["no segments, trailing bytes"] = bytes(z)
    cases["PARTIAL no segments"] = bytes(b[:hdr]) + (0).to_bytes(4, "little")
    z = bytearray(b); z[hdr + 4:hdr + 12] = int(lens[0] + 1).to_bytes(8, "little"); cases["segment length off by one"] = bytes(z)
    # the first run of segment 0 moved / grown
    nr0 = int.from_bytes(b[seg0:seg0 + 4], "little")
    assert nr0 >= 1
    run0 = seg0 + 4
    for name, word, val in (("run beyond the vertices", 0, info["nv"]), ("run beyond the faces", 1, info["nf"]), ("run beyond the half-edges", 2, info["ne"]),
                            ("run larger than the mesh", 4, info["nf"] + 1)):
        z = bytearray(b); z[run0 + 4 * word:run0 + 4 * word + 4] = int(val).to_bytes(4, "little"); cases[name] = bytes(z)
    if nseg >= 2:
        # two segments swapped with their lengths: still a valid container (segments are independent) -- must decode the same
        s0, s1 = bytes(b[seg0:seg0 + int(lens[0])]), bytes(b[seg0 + int(lens[0]):seg0 + int(lens[0]) + int(lens[1])])
        z = bytearray(b)
        z[hdr + 4:hdr + 12] = int(lens[1]).to_bytes(8, "little"); z[hdr + 12:hdr + 20] = int(lens[0]).to_bytes(8, "little")
        z[seg0:seg0 + len(s0) + len(s1)] = s1 + s0
        cases["OK swapped segments"] = bytes(z)
        # segment 1's first run claims segment 0's place
        seg1 = seg0 + int(lens[0])
        z = bytearray(b); z[seg1 + 4:seg1 + 4 + 12] = b[run0:run0 + 12]; cases["overlapping runs"] = bytes(z)
        # a segment dropped: what is left does not cover the mesh
        z = bytearray(b[:hdr]) + (nseg - 1).to_bytes(4, "little") + bytes(b[hdr + 12:seg0]) + bytes(b[seg0 + int(lens[0]):])
        cases["PARTIAL missing segment"] = bytes(z)
    return cases


def test_damaged_sharded_containers_are_refused(cx):
    gen = _mesh("mixed_nm")
    _, _, parts = _encode_sharded(cx, gen, 3, [], chunk_syms=1024)
    merged = hc.merge(parts)
    good = cx.read_hry(merged)
    assert hc.container_check(merged)
    for name, data in _damage_cases(merged).items():
        if name.startswith("OK"):
            assert hc.container_check(data)
            d = cx.read_hry(data)
            assert np.array_equal(d.org(), good.org()) and np.array_equal(d.list_data(1), good.list_data(1)), name
            continue
        if name.startswith("PARTIAL"):
            assert not hc.container_check(data)
            with pytest.raises(hc.HryError):
                cx.read_hry(data)
            d = cx.read_hry(data, partial=True)
            assert d.partial
            fo = d.face_offsets()
            assert (np.diff(fo.astype(np.int64)) >= 0).all() and fo[-1] == d.ne and (d.org() < d.nv).all() and (d.twin() < d.ne).all()
            with pytest.raises(hc.HryError):
                cx.write_hry(d, profile=hc.PROFILE_CHUNKED)
            with pytest.raises(hc.HryError):
                hc.ShardPlan(d, 2)
            continue
        with pytest.raises(hc.HryError):
            hc.container_check(data)
        for kw in ({}, {"partial": True}, {"shard": (0, 2)}):
            with pytest.raises(hc.HryError):
                cx.read_hry(data, **kw)
        with pytest.raises(hc.HryError):
            hc.merge([data, merged])
    # bit flips inside a segment body must end in an error or a mesh, never in a crash (bounded number of cases)
    info = hc.container_info(merged)
    rng = np.random.default_rng(5)
    body0 = info["header_bytes"] + 4 + 8 * info["segments"]
    for _ in range(24):
        z = bytearray(merged)
        at = int(rng.integers(body0, len(z)))
        z[at] ^= 1 << int(rng.integers(0, 8))
        try:
            d = cx.read_hry(bytes(z))
            assert d.nf == good.nf
        except hc.HryError:
            pass


@pytest.mark.gpu
def test_the_librarys_buffers_pass_through_the_binding(cx):
    """as_buffer=True: the binding returns what hry_encode / hry_merge allocated (no copy into bytes); such a buffer has the bytes'
    length and content, decodes, merges (next to bytes and numpy arrays) and is freed with the object"""
    from harry_amd import _native as nat
    gen = _mesh("mixed_nm")
    _, _, parts = _encode_sharded(cx, gen, 3, [])
    merged = hc.merge(parts)
    buf = hc.merge([parts[0], np.frombuffer(parts[1], dtype=np.uint8), parts[2]], as_buffer=True)
    assert isinstance(buf, nat.NativeBuffer) and len(buf) == len(merged) and buf == merged and bytes(buf.view()) == merged
    assert hc.container_info(buf)["minor"] == 3 and hc.container_check(buf)
    a, b = cx.read_hry(merged), cx.read_hry(buf)
    assert np.array_equal(a.list_data(1), b.list_data(1)) and np.array_equal(a.org(), b.org())
    assert hc.merge([buf]) == merged                         # a NativeBuffer as a part
    whole = hc.Mesh.from_arrays(gen.verts, gen.degrees, gen.indices, gen.face_props)
    one = cx.write_hry(whole.clone(), profile=hc.PROFILE_CHUNKED)
    one_buf = cx.write_hry(whole.clone(), profile=hc.PROFILE_CHUNKED, as_buffer=True)
    assert one_buf == one and len(one_buf) == len(one)
    mc = hc.MultiCodec([0, 0])
    try:
        m_bytes = mc.write_hry(whole.clone(), (), n_shards=3)
        m_buf = mc.write_hry(whole.clone(), (), n_shards=3, as_buffer=True)
        assert m_buf == m_bytes
        assert np.array_equal(mc.read_hry(m_buf).list_data(1), a.list_data(1))
    finally:
        mc.close()
    del buf, one_buf, m_buf                                  # (freed here: hry_free)
