"""Random OBJ scenes through both profiles on the GPU against the CPU oracle (development aid; the committed tests hold the fixed cases).
    python tests/tools/obj_stress.py [n_scenes] [seed]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from harry_amd import codec as hc
from harry_amd import meshgen as mg
from harry_amd import objgen as og
from oracle import oracle_py as op   # checker only

n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
cx = hc.Codec(0)


def same(a, o):
    assert (a.nv, a.nf, a.ne, a.nlists) == (o.nv, o.nf, o.ne, o.nlists)
    assert np.array_equal(a.org(), o.org()) and np.array_equal(a.twin(), o.twin())
    for k in (0, 1, 2):
        assert np.array_equal(a.bindings(k), o.bindings(k)), k
    for w in (0, 1):
        assert np.array_equal(a.regions_of(w), o.regions_of(w))
    for l in range(a.nlists):
        if a.list_target(l) != 3:
            assert np.array_equal(a.list_data(l), o.list_data(l)), l


t0 = time.time()
for it in range(n_scenes):
    kind = int(rng.integers(0, 5))
    polys = ["tri", "quad", "mixed"][int(rng.integers(0, 3))]
    if kind == 0:
        base = mg.torus(int(rng.integers(5, 40)), int(rng.integers(5, 40)), polys=polys, seed=int(rng.integers(1, 99)))
    elif kind == 1:
        base = mg.grid(int(rng.integers(4, 40)), int(rng.integers(4, 40)), seed=int(rng.integers(1, 99)))
    elif kind == 2:
        base = mg.icosphere(int(rng.integers(1, 5)))
    elif kind == 3:
        base = mg.multi_component(int(rng.integers(2, 7)), int(rng.integers(5, 14)), int(rng.integers(5, 14)), polys=polys, seed=int(rng.integers(1, 99)))
    else:
        base = mg.with_nonmanifold(mg.multi_component(int(rng.integers(2, 5)), 9, 11, polys=polys), int(rng.integers(1, 6)), int(rng.integers(1, 4)), seed=int(rng.integers(1, 99)))
    normals = [None, "smooth", "flat"][int(rng.integers(0, 3))]
    tex = [None, "atlas", "corner"][int(rng.integers(0, 3))]
    kw = dict(normals=normals, tex=tex, charts=int(rng.integers(1, 9)), materials=int(rng.integers(0, 4)), colors=[None, "all", "some"][int(rng.integers(0, 3))],
              tex3=bool(rng.integers(0, 2)), interleave=bool(rng.integers(0, 2)), negative=bool(rng.integers(0, 2)), crlf=bool(rng.integers(0, 2)), seed=int(rng.integers(1, 99)))
    if kw["materials"]:
        kw["mtl_name"] = "none.mtl"     # not on disk: every "usemtl" falls back to material 0, like the reference without the file
    sc = og.scene(base, **kw)
    m, o = hc.Mesh.from_obj(sc.obj, ""), op.Mesh.from_obj(sc.obj, "")
    quant = []
    if rng.integers(0, 2):
        for l in range(m.nlists):
            if rng.integers(0, 3):
                quant.append((l, -1, int(rng.integers(4, 17))))
    if quant:
        cx.requant(m, quant)
        o.requant(quant)
    want = o.clone().encode().data
    got = cx.write_hry(m.clone(), profile=hc.PROFILE_COMPAT)
    assert got == want, (it, kw, quant, "compat bytes")
    ref = op.Mesh.from_hry(want)
    same(cx.read_hry(want), ref)
    chunk = int(rng.choice([0, 64, 300, 4096]))
    c = cx.write_hry(m.clone(), profile=hc.PROFILE_CHUNKED, chunk_syms=chunk)
    assert c == o.clone().encode_chunked(hc.container_info(c)["chunk_syms"]).data, (it, kw, quant, "chunked bytes")
    os.environ["HRY_GENERIC_VERTEX"] = "1" if rng.integers(0, 2) else ""
    if not os.environ["HRY_GENERIC_VERTEX"]:
        del os.environ["HRY_GENERIC_VERTEX"]
    same(cx.read_hry(c), ref)
    os.environ.pop("HRY_GENERIC_VERTEX", None)
    assert cx.read_hry(want).to_obj() is not None
    print(f"scene {it}: {m.nf} faces, {m.nlists} lists, regions {m.nregions(0)}/{m.nregions(1)}, {kw['normals']}/{kw['tex']}/{kw['colors']}, quant {len(quant)}: ok ({time.time() - t0:.0f} s)", flush=True)
print("all ok")
