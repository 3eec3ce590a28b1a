"""Sharding of one mesh by groups of connected components (host side, no GPU): the plan, the extraction of a shard and the
seeded walk of a shard must reproduce exactly the part of the whole mesh's walk that belongs to the shard -- the same
vertices and faces in the same order at the positions the plan's exclusive scans give them
(cbm/encoder.h:61-68,79-113,187,215; formats/hry/writer.cc:28-46)."""
import numpy as np
import pytest

from harry_amd import codec as hc
from harry_amd import meshgen as mg


def _mesh(kind):
    if kind == "tori":
        return mg.concat([mg.torus(14 + 2 * i, 12 + i, seed=20 + i, center=(3.0 * i, 0, 0)) for i in range(7)])
    if kind == "mixed_nm":
        base = mg.multi_component(9, 13, 15, seed=4, polys="mixed")
        return mg.with_nonmanifold(base, 9, 5, seed=3)
    if kind == "quads_nm":
        return mg.with_nonmanifold(mg.multi_component(5, 10, 12, seed=5, polys="quad"), 4, 3, seed=8)
    raise ValueError(kind)


def _face_of_halfedge(foff, e):
    return np.searchsorted(foff, e, side="right") - 1


@pytest.mark.parametrize("kind", ["tori", "mixed_nm", "quads_nm"])
@pytest.mark.parametrize("n_shards", [1, 2, 3, 8])
def test_shard_walk_is_the_whole_walk_restricted(kind, n_shards):
    gen = _mesh(kind)
    whole = hc.Mesh.from_arrays(gen.verts, gen.degrees, gen.indices)
    plan = hc.ShardPlan(whole, n_shards)
    assert plan.ncomponents >= plan.ngroups >= 1
    assert sum(plan.triangles(s) for s in range(n_shards)) == whole.ntri
    org_w, foff_w = whole.org(), whole.face_offsets()
    shards = [plan.extract(whole, s) for s in range(n_shards)]      # before the walk mutates the whole mesh's twins
    ww = whole.clone().host_walk(plain=True)
    vseq_w = org_w[ww["order_v"]]                                    # input vertex at every position of the decoded numbering
    fseq_w = _face_of_halfedge(foff_w, ww["order_f"])
    nv_cov = nf_cov = 0
    covered_v = np.zeros(len(vseq_w), bool)
    for s, sh in enumerate(shards):
        runs = sh.runs()
        if sh.nf == 0:
            assert len(runs) == 0
            continue
        assert int(runs[:, 3].sum()) == sh.nv and int(runs[:, 4].sum()) == sh.nf and int(runs[:, 5].sum()) == sh.ne
        vof, fof = sh.shard_elements(1), sh.shard_elements(0)
        # records travel with their elements
        assert np.array_equal(sh.list_data(1), whole.list_data(1)[vof])
        org_s, foff_s = sh.org(), sh.face_offsets()
        sw = sh.host_walk(plain=True)
        vseq_s = vof[org_s[sw["order_v"]]]
        fseq_s = fof[_face_of_halfedge(foff_s, sw["order_f"])]
        want_v = np.concatenate([vseq_w[r[0]:r[0] + r[3]] for r in runs])
        want_f = np.concatenate([fseq_w[r[1]:r[1] + r[4]] for r in runs])
        assert np.array_equal(vseq_s, want_v)
        assert np.array_equal(fseq_s, want_f)
        # the start half-edge of every face is the same corner of the same face
        he_w = np.concatenate([ww["order_f"][r[1]:r[1] + r[4]] for r in runs]) - foff_w[want_f]
        he_s = sw["order_f"] - foff_s[_face_of_halfedge(foff_s, sw["order_f"])]
        assert np.array_equal(he_w, he_s)
        for r in runs:
            assert not covered_v[r[0]:r[0] + r[3]].any()
            covered_v[r[0]:r[0] + r[3]] = True
        nv_cov += sh.nv
        nf_cov += sh.nf
    assert nf_cov == whole.nf and nv_cov == len(vseq_w) and covered_v.all()


def test_plan_balances_equal_components():
    gen = mg.multi_component(16, 9, 10, seed=4, polys="tri")
    whole = hc.Mesh.from_arrays(gen.verts, gen.degrees, gen.indices)
    plan = hc.ShardPlan(whole, 4)
    assert plan.ncomponents == 16 and plan.ngroups == 16
    loads = [plan.triangles(s) for s in range(4)]
    assert max(loads) == min(loads) == whole.ntri // 4
    # deterministic
    plan2 = hc.ShardPlan(whole, 4)
    assert [plan2.triangles(s) for s in range(4)] == loads


def test_single_group_cannot_be_split():
    gen = mg.torus(12, 14, seed=2)
    whole = hc.Mesh.from_arrays(gen.verts, gen.degrees, gen.indices)
    plan = hc.ShardPlan(whole, 4)
    loads = sorted(plan.triangles(s) for s in range(4))
    assert loads == [0, 0, 0, whole.ntri]
    empty = [plan.extract(whole, s) for s in range(4) if plan.triangles(s) == 0][0]
    assert empty.nf == 0 and empty.nv == 0


def test_merge_rejects_foreign_input():
    with pytest.raises(hc.HryError):
        hc.merge([b"not a container"])


# ---- general bindings (OBJ scenes): components tied by shared records, record ranges of the runs --------------------------------
_TWO_PARTS_SHARING_A_NORMAL = b"""v 0 0 0
v 1 0 0
v 1 1 0
v 0 1 0
v 3 0 0
v 4 0 0
v 4 1 0
v 6 0 0
v 7 0 1
v 7 1 0
vn 0 0 1
vn 0.6 0 0.8
f 1//1 2//1 3//1 4//1
f 5//1 6//1 7//1
f 8//2 9//2 10//2
"""


def test_general_plan_ties_components_that_share_a_record():
    whole = hc.Mesh.from_obj(_TWO_PARTS_SHARING_A_NORMAL, "")
    assert whole.general
    plan = hc.ShardPlan(whole, 2)
    assert plan.ncomponents == 3 and plan.ngroups == 2          # the two parts with "vn 1" are one group, the third is free
    shards = [plan.extract(whole, s) for s in range(2)]
    assert sorted(sh.nf for sh in shards) == [1, 2]
    nl = whole.nlists
    for sh in shards:
        assert sh.general and sh.nlists == nl
        # every element of a shard names records of the shard
        for l in range(nl):
            if sh.list_target(l) == 3:
                continue
            rec_of = sh.shard_elements(16 + l)
            assert len(rec_of) == sh.list_count(l)
            assert np.array_equal(sh.list_data(l), whole.list_data(l)[rec_of])
        b = sh.bindings(2)
        for l in range(nl):
            if sh.list_target(l) == 2 and sh.list_count(l):
                assert b.max() < max(sh.list_count(x) for x in range(nl))
    # the two shards' records of every list partition the whole list
    for l in range(nl):
        if whole.list_target(l) == 3:
            continue
        both = np.concatenate([sh.shard_elements(16 + l) for sh in shards])
        assert sorted(both.tolist()) == list(range(whole.list_count(l)))


def test_container_check_is_host_only_and_rejects_garbage():
    with pytest.raises(hc.HryError):
        hc.container_check(b"\xfa\xff\xaf\xaf\x00\x03" + b"\x00" * 8)
    with pytest.raises(hc.HryError):
        hc.container_check(b"not a container")


@pytest.mark.parametrize("kind", ["tori", "mixed_nm", "quads_nm"])
@pytest.mark.parametrize("n_shards", [1, 3, 8])
@pytest.mark.parametrize("threads", [1, 3])
def test_shard_walked_in_place_equals_the_walk_of_its_extracted_mesh(kind, n_shards, threads, monkeypatch):
    """hry_walk_run_shard: the components of a shard walked where they lie in the whole mesh (what a worker of the in-process
    executor does since round 4) -- every symbol, position, count and mark equal to the walk of the extracted sub-mesh, the
    coded vertices and faces the same corners of the same elements, the repaired twins the same edges."""
    monkeypatch.setenv("HRY_HOST_THREADS", str(threads))
    monkeypatch.setenv("HRY_PARALLEL_MIN_FACES", "1")
    gen = _mesh(kind)
    whole = hc.Mesh.from_arrays(gen.verts, gen.degrees, gen.indices)
    plan = hc.ShardPlan(whole, n_shards)
    foff_w = whole.face_offsets()
    shards = [plan.extract(whole, s) for s in range(n_shards)]      # before any walk mutates the whole mesh's twins
    place = whole.clone()
    for s, sh in enumerate(shards):
        if sh.nf == 0:
            continue
        fof = sh.shard_elements(0)
        foff_s = sh.face_offsets()
        sw = sh.host_walk(plain=True)
        pw = plan.walk_in_place(place, s)
        for k in sw:
            if k in ("order_v", "order_f"):
                f_loc = _face_of_halfedge(foff_s, sw[k])
                want = foff_w[fof[f_loc]] + (sw[k] - foff_s[f_loc])     # the same half-edge in the whole mesh's numbering
                assert np.array_equal(pw[k], want), (k, s)
            elif k == "marks":
                assert np.array_equal(pw[k], sw[k]), (k, s)
            else:
                assert np.array_equal(pw[k], sw[k]), (k, s)
        # repaired twins: the shard's half-edges of the whole mesh now point where the extracted mesh's do
        tw_s, tw_p = sh.twin(), place.twin()
        he_w = np.concatenate([np.arange(foff_w[f], foff_w[f + 1]) for f in fof]) if len(fof) else np.zeros(0, np.int64)
        f_loc = _face_of_halfedge(foff_s, tw_s)
        want = foff_w[fof[f_loc]] + (tw_s - foff_s[f_loc])
        assert np.array_equal(tw_p[he_w], want), s
