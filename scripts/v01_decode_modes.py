"""Development: the host side of reading a reference stream (.hry v0.1) -- serial entropy decode + replay -- on a mesh shaped like
configs[3] (mixed polygons, lossless floats) and on the headline torus, for the decoder variants HRY_V01_DECODER selects.
python scripts/v01_decode_modes.py [components]   (host only; HRY_PERF=1 HRY_TRACE=1 for counters and phases)"""
import sys, os, time, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from harry_amd import codec as hc, meshgen as mg
from oracle import oracle_py as op

def streams(nc):
    mesh = mg.multi_component(nc, 221, 222, seed=4, polys="mixed")
    mesh = mg.with_nonmanifold(mesh, n_edges=max(1, mesh.ntri // 1000), n_vtx=max(1, mesh.ntri // 2000))
    o = op.Mesh.from_ply(mesh.to_ply())
    yield f"configs[3]-like, {nc} components, lossless", o.encode().data, mesh.ntri + mesh.nv * 13 + mesh.nf, mesh.ntri
    mesh = mg.torus(708, 708, seed=2)
    for name, q in (("torus q14", [(1, -1, 14)]), ("torus lossless", [])):
        o = op.Mesh.from_ply(mesh.to_ply())
        if q:
            o.requant(q)
        yield name, o.encode().data, mesh.ntri + mesh.nv * (1 + (6 if q else 12)) + mesh.nf, mesh.ntri

if __name__ == "__main__":
    nc = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    for name, data, nsym, ntri in streams(nc):
        ts = []
        for i in range(int(os.environ.get("REPS", "7"))):
            t = time.perf_counter(); r = hc.read_stream_host(data); ts.append(time.perf_counter() - t)
        ts.sort()
        print(f"decoder {os.environ.get('HRY_V01_DECODER', 'default')}: {name}: min {ts[0]*1e3:.1f} ms, median {ts[len(ts)//2]*1e3:.1f} ms for {ntri} triangles = {ts[0]/nsym*1e9:.2f} ns per symbol", flush=True)
