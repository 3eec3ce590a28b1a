"""OBJ reader of the product (host side, no GPU): the mesh it builds -- lists, regions, element -> record tables, connectivity,
bounds formats -- equals the CPU oracle's, which is byte-pinned to the reference binary (tests/test_oracle_obj.py).
Reference: formats/obj/reader.rl:27-299."""
import json
import os

import numpy as np
import pytest

from harry_amd import codec as hc
from harry_amd import meshgen as mg
from harry_amd import objgen as og
from oracle import oracle_py as op   # checker only
from tests import util

OBJ = os.path.join(util.ROOT, "tests", "golden", "obj")
with open(os.path.join(OBJ, "manifest.json")) as _f:
    MAN = json.load(_f)


def same_general_mesh(a, o):
    """a: harry_amd mesh, o: oracle mesh"""
    assert a.general and o.general
    assert (a.nv, a.nf, a.ne, a.nlists) == (o.nv, o.nf, o.ne, o.nlists)
    assert np.array_equal(a.face_offsets(), o.face_offsets()) and np.array_equal(a.org(), o.org()) and np.array_equal(a.twin(), o.twin())
    for l in range(a.nlists):
        assert a.list_target(l) == o.list_target(l)
        if a.list_target(l) == 3:
            continue
        assert a.list_fmt(l) == o.list_fmt(l) and a.list_count(l) == o.list_count(l)
        for c in range(len(a.list_fmt(l))):
            assert np.array_equal(a.component(l, c), o.component(l, c)), (l, c)
    for which in (0, 1):
        assert a.nregions(which) == o.nregions(which)
        assert np.array_equal(a.regions_of(which), o.regions_of(which))
    for r in range(a.nregions(0)):
        assert a.region_lists(0, r) == o.region_lists(0, r) and a.region_lists(2, r) == o.region_lists(2, r)
    for r in range(a.nregions(1)):
        assert a.region_lists(1, r) == o.region_lists(1, r)
    for kind in (1, 2):
        assert np.array_equal(a.bindings(kind), o.bindings(kind)), kind


@pytest.mark.parametrize("name", sorted(MAN["small"]))
def test_obj_reader_builds_the_oracles_mesh(name):
    data = open(os.path.join(OBJ, name + ".obj"), "rb").read()
    same_general_mesh(hc.Mesh.from_obj(data, OBJ), op.Mesh.from_obj(data, OBJ))


def test_obj_reader_bigger_scene():
    sc = og.scene(mg.torus(60, 64, polys="mixed"), normals="smooth", tex="atlas", charts=5, colors="some")
    same_general_mesh(hc.Mesh.from_obj(sc.obj, ""), op.Mesh.from_obj(sc.obj, ""))


def test_obj_grammar_errors_read_like_the_reference():
    tri = b"v 0 0 0\nv 1 0 0\nv 0 1 0\n"
    for bad in (b"v 1e2 0 0\n", b"g\n", b"v 1 2\n", b"v 1 2 3 4 5\n", b"vt 1\n", b"vn 1 2\n", b"x 1\n", b" v 1 2 3\n", b"v 1 2 3\rx\n"):
        with pytest.raises(hc.HryError, match="Unable to parse this OBJ file") as e:
            hc.Mesh.from_obj(tri + bad + b"f 1 2 3\n")
        assert e.value.code == -2
    with pytest.raises(hc.HryError, match="n too big"):
        hc.Mesh.from_obj(tri + b"vt 0 0\nf 1/1 2/1 3/1\n")
    with pytest.raises(hc.HryError, match="index cannot be 0"):
        hc.Mesh.from_obj(tri + b"f 0 1 2\n")
    with pytest.raises(hc.HryError, match="n too small"):
        hc.Mesh.from_obj(tri + b"f -4 1 2\n")
    m = hc.Mesh.from_obj(tri + b"f 1 2 3")      # last line without a line feed: dropped
    assert (m.nv, m.nf) == (3, 0)
    m = hc.Mesh.from_obj(b"v 1e-1 2e+1 -.5\nv 0 0 0\nv 1 1 1\nf 1 2 3\n")
    assert m.list_data(0).view("<f4").reshape(-1, 3)[0].tolist() == [10.0, 20.0, -0.5]


def test_obj_writer_roundtrips_what_the_reader_built():
    """to_obj of an OBJ-read mesh, read again, is the same mesh when normals and texture coordinates do not both occur (the writer
    numbers normals after ALL texture coordinates, formats/obj/writer.cc:68-93)"""
    sc = og.scene(mg.torus(9, 11), normals="smooth")
    a = hc.Mesh.from_obj(sc.obj, "")
    b = hc.Mesh.from_obj(a.to_obj(), "")
    assert np.array_equal(a.org(), b.org()) and np.array_equal(a.bindings(2), b.bindings(2))
    for l in range(a.nlists):
        x, y = a.list_data(l).view("<f4"), b.list_data(l).view("<f4")
        plain = np.abs(x) >= 1e-4     # "%g" prints smaller values as 8.2e-05, which the scanner reads as 8.2e+05 (reader.rl:36-37,44)
        assert np.allclose(x[plain], y[plain], rtol=1e-5, atol=1e-6) and plain.mean() > 0.9


@pytest.mark.timeout(600)
def test_fast_event_collection_equals_the_plain_one(tmp_path):
    """host/general_events.cpp (round 5: sized once, bare cursors, no character-typed stores, positions only on request) against a
    plain restatement of the same bookkeeping (tests/native/events_check.cpp): every array of every list equal, with and without
    positions, on the reference's OBJ fixtures and on generated scenes (shared records, one normal per face, per-corner texture
    coordinates, several materials, mixed polygons, non-manifold parts)."""
    import glob
    import shutil
    import subprocess
    from concurrent.futures import ThreadPoolExecutor
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    root = util.ROOT
    host = os.path.join(root, "harry_amd", "csrc", "host")
    units = [(os.path.join(host, s + ".cpp"), str(tmp_path / (s + ".o"))) for s in
             ("block_pool", "thread_pool", "ply_io", "obj_io", "header", "cbm_walk", "cbm_unwalk", "compat_read", "shard", "general_events")]
    units.append((os.path.join(root, "tests", "native", "events_check.cpp"), str(tmp_path / "driver.o")))
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 2)) as ex:
        for r in ex.map(lambda u: subprocess.run(["g++", "-O1", "-std=c++17", "-pthread", "-c", u[0], "-o", u[1]], capture_output=True, text=True), units):
            assert r.returncode == 0, r.stderr[-3000:]
    exe = str(tmp_path / "events_check")
    r = subprocess.run(["g++", "-pthread", *[u[1] for u in units], "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    files = [f for f in sorted(glob.glob(os.path.join(root, "tests", "golden", "obj", "*.obj"))) if ".dec." not in f and ".ll." not in f and ".q" not in os.path.basename(f)]
    scenes = [og.scene(mg.torus(24, 20, seed=3, polys="mixed"), normals="smooth", tex="atlas", charts=5, colors="some"),
              og.scene(mg.torus(22, 18, seed=4), normals="flat", tex="corner"),
              og.scene(mg.with_nonmanifold(mg.multi_component(6, 9, 10, seed=5, polys="mixed"), 7, 4, seed=2), normals="flat", tex="corner")]
    for i, sc in enumerate(scenes):
        files.append(str(tmp_path / f"scene{i}.obj"))
        with open(files[-1], "wb") as f:
            f.write(sc.obj)
    r = subprocess.run([exe, *files], capture_output=True, text=True, timeout=500)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok %d files" % len(files)), (r.stdout + r.stderr)[-3000:]


def test_faces_with_mixed_index_kinds_are_refused():
    """`f 1/1 2 3`: the reference reads past its index arrays there (formats/obj/reader.rl:212-277); refused with the reason"""
    head = b"v 0 0 0\nv 1 0 0\nv 0 1 0\nvt 0 0\nvt 1 0\nvn 0 0 1\nvn 0 1 0\n"
    for face in (b"f 1/1/1 2 3\n", b"f 1/1/1 2/2/2 3\n", b"f 1//1 2 3//2\n"):
        with pytest.raises(hc.HryError, match="same kinds of index"):
            hc.Mesh.from_obj(head + face, "")
    ok = hc.Mesh.from_obj(head + b"f 1/1/1 2/2/2 3/1/1\n", "")
    assert ok.nf == 1
