import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from harry_amd import codec as hc, meshgen as mg
from oracle import oracle_py as op
"""Development: the host side of reading a reference stream (.hry v0.1: serial entropy decode + replay), quantised and lossless."""
mesh = mg.torus(400, 400, seed=2)
for name, q in (("q14", [(1, -1, 14)]), ("lossless float", [])):
    o = op.Mesh.from_ply(mesh.to_ply())
    if q:
        o.requant(q)
    data = o.encode().data
    nsym = mesh.ntri + mesh.nv * (1 + (6 if q else 12)) + mesh.nf   # operations + (type + residual bytes) per vertex + a type per face
    best = 1e9
    for i in range(5):
        t = time.perf_counter(); r = hc.read_stream_host(data); dt = time.perf_counter() - t; best = min(best, dt)
    print(os.environ.get("HRY_LIB", "library"), os.environ.get("HRY_SCALAR_DECODER", ""), name, f"{best*1e3:.1f} ms for {mesh.ntri} triangles = {best/nsym*1e9:.1f} ns per symbol, {len(data)} bytes")
