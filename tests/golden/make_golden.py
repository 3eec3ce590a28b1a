#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ by running the UNMODIFIED reference binary.

Run in the build container only (needs oracle/_ref/harry_ref, built by `make -C oracle ref` from
/root/reference).  What is committed is DATA: our own synthetic PLY inputs, the reference's .hry outputs,
the reference's decoded PLY outputs, function-level known answers, and a manifest with sizes/hashes for
larger regenerated cases.  No reference source text is copied.

    python tests/golden/make_golden.py
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
sys.path.insert(0, ROOT)
from harry_amd import meshgen as mg  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "harry_ref")

POSNRM = ["-l1", "-a0", "-q14", "-a1", "-q14", "-a2", "-q14", "-a3", "-q10", "-a4", "-q10", "-a5", "-q10"]


def small_cases():
    """name -> (mesh, ply format, list of (tag, cli flags))"""
    return {
        "grid50": (mg.grid(50), "binary_little_endian", [("ll", []), ("q14", ["-l1", "-q14"]), ("q8", ["-l1", "-q8"]), ("c", ["-c"])]),
        "grid_ascii": (mg.grid(20, 13, seed=5), "ascii", [("ll", []), ("q11", ["-l1", "-q11"])]),
        "torus_mixed": (mg.torus(24, 30, polys="mixed"), "binary_big_endian", [("ll", []), ("q12", ["-l1", "-q12"])]),
        "torus_quad": (mg.torus(16, 12, polys="quad"), "binary_little_endian", [("ll", [])]),
        "ico3": (mg.icosphere(3), "binary_little_endian", [("ll", []), ("q16", ["-l1", "-q16"]), ("q20", ["-l1", "-q20"]),
                                                            ("a0q9", ["-l1", "-a0", "-q9"])]),
        "multi5": (mg.multi_component(5, 10, 12), "binary_little_endian", [("ll", []), ("q10", ["-l1", "-q10"])]),
        "nonmanifold": (mg.with_nonmanifold(mg.torus(12, 12), 3, 2), "binary_little_endian", [("ll", []), ("q14", ["-l1", "-q14"])]),
        "colors_normals": (mg.with_colors(mg.torus(10, 14, normals=True)), "binary_little_endian",
                           [("ll", []), ("posnrm", POSNRM), ("q6", ["-l1", "-q6"])]),
        "faceprops": (mg.with_face_props(mg.grid(9, 7)), "binary_little_endian", [("ll", []), ("q", ["-l0", "-q6", "-l1", "-q11"])]),
        "negative": (mg.negated(mg.grid(11)), "binary_little_endian", [("ll", []), ("q14", ["-l1", "-q14"])]),
        "tiny_tri": (mg.grid(2), "ascii", [("ll", []), ("q4", ["-l1", "-q4"])]),
        # 8-byte sources of the quantiser (quant.h:137-139); lossless doubles are outside the reference's own defined behaviour
        "grid_double": (mg.doubles(mg.grid(12, 9, seed=3)), "binary_little_endian", [("q14", ["-l1", "-q14"]), ("q30", ["-l1", "-q30"])]),
    }


def requant_cases():
    """re-quantisation / dequantisation of an already quantised .hry: name -> (source fixture, flags)  (quant.h:169-212, main.cc:44,108)"""
    return {
        "grid50.q14_to_q10": ("grid50.q14.hry", ["-l1", "-q10"]),
        "grid50.q14_c": ("grid50.q14.hry", ["-c"]),
        "grid50.q14_a1q0": ("grid50.q14.hry", ["-l1", "-a1", "-q0"]),
        "colors_normals.q6_c": ("colors_normals.q6.hry", ["-c"]),
        "colors_normals.posnrm_c_q8": ("colors_normals.posnrm.hry", ["-c", "-l1", "-a6", "-q5", "-a3", "-q8"]),
        "faceprops.q_c": ("faceprops.q.hry", ["-c"]),
        "grid_double.q14_to_q9": ("grid_double.q14.hry", ["-l1", "-q9"]),
    }


def big_cases():
    """Regenerated, not committed: only size + sha256 of the reference output go into the manifest."""
    return {
        "torus150": (lambda: mg.torus(150, 150, seed=2), [("ll", []), ("q14", ["-l1", "-q14"])]),
        "multi40": (lambda: mg.multi_component(40, 20, 22), [("ll", [])]),
        "ico5": (lambda: mg.icosphere(5), [("ll", []), ("q12", ["-l1", "-q12"])]),
        "nm_big": (lambda: mg.with_nonmanifold(mg.torus(60, 64, polys="mixed"), 30, 12), [("ll", [])]),
    }


def run_ref(args):
    r = subprocess.run([REF] + args, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"reference failed: {args}: {r.stderr[-400:]}")


def sha(b: bytes) -> str:
    return hashlib.sha256(b).hexdigest()


def main():
    if not os.path.exists(REF):
        sys.exit("build the reference first: make -C oracle ref")
    manifest = {"libstdcxx_note": "start-face order depends on std::unordered_set iteration order (SURVEY App. B-1); "
                                  "generated with g++ 11 / libstdc++ GLIBCXX_3.4.30", "small": {}, "big": {}, "requant_of_hry": {}}
    with tempfile.TemporaryDirectory() as tmp:
        for name, (mesh, fmt, variants) in small_cases().items():
            ply = mesh.to_ply(fmt)
            with open(os.path.join(HERE, name + ".ply"), "wb") as f:
                f.write(ply)
            entry = {"nv": mesh.nv, "nf": mesh.nf, "ntri": mesh.ntri, "ply_format": fmt, "variants": {}}
            for tag, flags in variants:
                hry = os.path.join(HERE, f"{name}.{tag}.hry")
                run_ref([os.path.join(HERE, name + ".ply"), hry] + flags)
                dec = os.path.join(HERE, f"{name}.{tag}.dec.ply")
                run_ref([hry, dec])
                entry["variants"][tag] = {"flags": flags, "hry_bytes": os.path.getsize(hry),
                                          "hry_sha256": sha(open(hry, "rb").read()),
                                          "dec_sha256": sha(open(dec, "rb").read())}
            manifest["small"][name] = entry
        # re-quantisation / dequantisation of an already quantised .hry (quant.h:169-212)
        for name, (src_name, flags) in requant_cases().items():
            src = os.path.join(HERE, src_name)
            dst = os.path.join(HERE, name + ".hry")
            run_ref([src, dst] + flags)
            manifest["requant_of_hry"][name] = {"src": src_name, "flags": flags, "hry_sha256": sha(open(dst, "rb").read())}
        for name, (make, variants) in big_cases().items():
            mesh = make()
            p = os.path.join(tmp, name + ".ply")
            with open(p, "wb") as f:
                f.write(mesh.to_ply())
            entry = {"nv": mesh.nv, "nf": mesh.nf, "ntri": mesh.ntri, "ply_sha256": sha(open(p, "rb").read()), "variants": {}}
            for tag, flags in variants:
                hry = os.path.join(tmp, f"{name}.{tag}.hry")
                run_ref([p, hry] + flags)
                dec = os.path.join(tmp, f"{name}.{tag}.dec.ply")
                run_ref([hry, dec])
                entry["variants"][tag] = {"flags": flags, "hry_bytes": os.path.getsize(hry),
                                          "hry_sha256": sha(open(hry, "rb").read()),
                                          "dec_sha256": sha(open(dec, "rb").read())}
            manifest["big"][name] = entry
    with open(os.path.join(HERE, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    total = sum(os.path.getsize(os.path.join(HERE, f)) for f in os.listdir(HERE))
    print("golden fixtures written:", len(os.listdir(HERE)), "files,", total, "bytes")


if __name__ == "__main__":
    main()
