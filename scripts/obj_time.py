"""Timing of the OBJ / general-bindings path on one GPU: parse, encode (reference stream), decode.
    python scripts/obj_time.py [n] [bits]  torus n x n (2 n^2 triangles), smooth normals + a 7-chart texture atlas, then flat normals;
                                           bits: quantise every list to that many bits first (integer records)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from harry_amd import codec as hc
from harry_amd import meshgen as mg
from harry_amd import objgen as og

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
bits = int(sys.argv[2]) if len(sys.argv) > 2 else 0
cx = hc.Codec(0)
for label, kw in (("smooth normals + atlas", dict(normals="smooth", tex="atlas", charts=7)), ("flat normals", dict(normals="flat"))):
    t = time.time()
    sc = og.scene(mg.torus(n, n, seed=2), **kw)
    print(f"[{label}] scene text {len(sc.obj) / 1e6:.1f} MB in {time.time() - t:.1f} s", flush=True)
    t = time.time()
    m = hc.Mesh.from_obj(sc.obj, "")
    t_parse = time.time() - t
    if bits:
        cx.requant(m, [(l, -1, bits) for l in range(m.nlists) if m.list_target(l) != 3], False)
    ntri = m.ntri
    for rep in range(4):
        prof = hc.PROFILE_COMPAT if rep < 2 else hc.PROFILE_CHUNKED
        a = m.clone(); cx.upload(a)   # (resident inputs, as bench.py measures)
        t = time.time()
        data = cx.write_hry(a, profile=prof)
        t_enc = time.time() - t
        te = cx.timing()
        t = time.time()
        d = cx.read_hry(data)
        t_dec = time.time() - t
        td = cx.timing()
        print(f"[{label}] {'compat ' if rep < 2 else 'chunked'} pass {rep % 2}: {ntri} triangles, parse {t_parse * 1e3:.0f} ms, encode {t_enc * 1e3:.0f} ms ({ntri / t_enc / 1e6:.2f} Mtri/s; host {te['host_walk_ms']:.0f} ms, "
              f"kernels {te['device_ms']:.0f} ms), decode {t_dec * 1e3:.0f} ms ({ntri / t_dec / 1e6:.2f} Mtri/s; host {td['host_walk_ms']:.0f} ms, kernels {td['device_ms']:.0f} ms), "
              f"{len(data)} bytes = {8 * len(data) / max(m.nv, 1):.1f} bits/vertex", flush=True)
    assert cx.write_hry(d, profile=hc.PROFILE_COMPAT) is not None
cx.close()
