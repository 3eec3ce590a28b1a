#!/usr/bin/env python3
"""Golden fixtures of the OBJ path, written by the UNMODIFIED reference binary into tests/golden/obj/.

Run in the build container only (needs oracle/_ref/harry_ref, `make -C oracle ref`).  What is committed is DATA: our own
synthetic OBJ scenes (harry_amd/objgen.py), the reference's .hry of each, and the reference's OBJ / PLY decodes of those.

    python tests/golden/make_golden_obj.py
"""
from __future__ import annotations

import hashlib
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(HERE, "obj")
sys.path.insert(0, ROOT)
from harry_amd import meshgen as mg  # noqa: E402
from harry_amd import objgen as og  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "harry_ref")

HAND = b"""# hand-written: the grammar's corners (formats/obj/reader.rl:27-80)
mtllib hand.mtl
o thing
v 0 0 0
v 1e-1 0 0
v 1. 1 0 1
v +.5 1 -0
v 0.5 0.5 1.25 1 0.5 0.25
v 2e+0 0 0.5 0.1 0.2 0.3 0.4
v 2 1 0.5 0.1 0.2 0.3 0.4 0.5
vt 0 0
vt 1 0 0.5
vt 1 1
vn 0 0 1
vn 0 1 0
g a b c
s off
usemtl  shiny
f 1/1/1 2/1/1 3/3/2 4/1/2
f 1//1 5//2 2//2\t
usemtl shiny
f -1 -2 -6
f 2/2 6/2 7/2
l 1 2
p 1

\t
f 3/1/1 2/3/2 -1/3/1
"""
HAND_MTL = b"newmtl shiny\nKd 1 1 1\nnewmtl dull extra tokens\nnewmtl shiny\n"


def small_cases():
    """name -> (scene, [(tag, flags)])"""
    return {
        "plain": (og.scene(mg.grid(12, 9, seed=2)), [("ll", []), ("q12", ["-l0", "-q12"])]),
        "smooth": (og.scene(mg.torus(10, 12), normals="smooth", tex="atlas", charts=3),
                   [("ll", []), ("q", ["-l0", "-q14", "-l1", "-q12", "-l2", "-q10"]), ("qn", ["-l2", "-a1", "-q9"])]),
        "flat": (og.scene(mg.icosphere(2), normals="flat"), [("ll", []), ("q8", ["-l1", "-q8"])]),
        "mtl": (og.scene(mg.torus(9, 10, polys="mixed"), tex="corner", materials=3, chatter=True, mtl_name="mtl.mtl"), [("ll", [])]),
        "mixedfmt": (og.scene(mg.grid(8, 7, seed=4), normals="smooth", tex="atlas", charts=4, colors="some", tex3=True,
                              interleave=True, negative=True, crlf=True, materials=2, mtl_name="mixedfmt.mtl"), [("ll", []), ("q10", ["-l0", "-q10"])]),
        "colors": (og.scene(mg.torus(8, 9, polys="quad"), colors="all", normals="flat", tex="atlas", charts=2), [("ll", [])]),
        "nm": (og.scene(mg.with_nonmanifold(mg.multi_component(4, 9, 10, polys="mixed"), 4, 3), normals="smooth", tex="atlas", charts=5),
               [("ll", []), ("q11", ["-l0", "-q11", "-l1", "-q11", "-l2", "-q11"])]),
    }


def requant_cases():
    return {
        "smooth.q_c": ("smooth.q.hry", ["-c"]),
        "smooth.q_to_q8": ("smooth.q.hry", ["-l0", "-q8", "-l2", "-q6"]),
    }


def from_ply_cases():
    """.hry files the reference wrote from PLY input, decoded to OBJ (formats/obj/writer.cc:20-132 on the PLY layout)"""
    return ["grid50.ll.hry", "colors_normals.ll.hry", "torus_mixed.q12.hry"]


def big_cases():
    return {
        "torus150": (lambda: og.scene(mg.torus(150, 150, seed=2), normals="smooth", tex="atlas", charts=7), [("ll", []), ("q", ["-l0", "-q14", "-l1", "-q12", "-l2", "-q10"])]),
        "flat_ico5": (lambda: og.scene(mg.icosphere(5), normals="flat", tex="corner"), [("ll", [])]),
    }


def run_ref(args, cwd=None):
    r = subprocess.run([REF] + args, capture_output=True, text=True, cwd=cwd)
    if r.returncode != 0:
        raise RuntimeError(f"reference failed: {args}: {r.stderr[-400:]}")


def sha(b: bytes) -> str:
    return hashlib.sha256(b).hexdigest()


def write_scene(dirname, name, sc):
    with open(os.path.join(dirname, name + ".obj"), "wb") as f:
        f.write(sc.obj)
    for fn, data in sc.files.items():
        with open(os.path.join(dirname, fn), "wb") as f:
            f.write(data)


def variants_of(dirname, name, variants):
    out = {}
    for tag, flags in variants:
        hry = os.path.join(dirname, f"{name}.{tag}.hry")
        run_ref([os.path.join(dirname, name + ".obj"), hry] + flags)   # a path with a directory: "mtllib" is looked up next to the file
        dec = os.path.join(dirname, f"{name}.{tag}.dec.obj")
        run_ref([hry, dec])
        decp = os.path.join(dirname, f"{name}.{tag}.dec.ply")
        run_ref([hry, decp, "--ply-ascii"])
        out[tag] = {"flags": flags, "hry_bytes": os.path.getsize(hry), "hry_sha256": sha(open(hry, "rb").read()),
                    "dec_obj_sha256": sha(open(dec, "rb").read()), "dec_ply_sha256": sha(open(decp, "rb").read())}
    return out


def main():
    if not os.path.exists(REF):
        sys.exit("build the reference first: make -C oracle ref")
    os.makedirs(OUT, exist_ok=True)
    manifest = {"small": {}, "big": {}, "requant_of_hry": {}, "ply_to_obj": {}}
    for name, (sc, variants) in small_cases().items():
        write_scene(OUT, name, sc)
        manifest["small"][name] = {"nv": sc.nv, "nf": sc.nf, "mtl": sorted(sc.files), "variants": variants_of(OUT, name, variants)}
    with open(os.path.join(OUT, "hand.obj"), "wb") as f:
        f.write(HAND)
    with open(os.path.join(OUT, "hand.mtl"), "wb") as f:
        f.write(HAND_MTL)
    manifest["small"]["hand"] = {"nv": 7, "nf": 5, "mtl": ["hand.mtl"], "variants": variants_of(OUT, "hand", [("ll", [])])}
    for name, (src_name, flags) in requant_cases().items():
        dst = os.path.join(OUT, name + ".hry")
        run_ref([os.path.join(OUT, src_name), dst] + flags)
        dec = os.path.join(OUT, name + ".dec.obj")
        run_ref([dst, dec])
        manifest["requant_of_hry"][name] = {"src": src_name, "flags": flags, "hry_sha256": sha(open(dst, "rb").read())}
    for src in from_ply_cases():
        dst = os.path.join(OUT, src[:-4] + ".dec.obj")
        run_ref([os.path.join(HERE, src), dst])
        manifest["ply_to_obj"][src] = {"dec_obj_sha256": sha(open(dst, "rb").read())}
    with tempfile.TemporaryDirectory() as tmp:
        for name, (make, variants) in big_cases().items():
            sc = make()
            write_scene(tmp, name, sc)
            manifest["big"][name] = {"nv": sc.nv, "nf": sc.nf, "obj_sha256": sha(sc.obj), "variants": variants_of(tmp, name, variants)}
    with open(os.path.join(OUT, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    total = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT))
    print("OBJ golden fixtures written:", len(os.listdir(OUT)), "files,", total, "bytes")


if __name__ == "__main__":
    main()
