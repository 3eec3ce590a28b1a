"""Development: container size and entropy-kernel times of the chunked profile against the chunk size, on a mesh shaped like
configs[3] (python scripts/chunk_size_sweep.py COMPONENTS [sizes...]; 0 = the default policy)."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from harry_amd import codec as hc, meshgen as mg
nc = int(sys.argv[1]) if len(sys.argv) > 1 else 128
sizes = [int(a) for a in sys.argv[2:]] or [0, 131072, 32768, 16384, 8192, 4096, 2048]
mesh = mg.multi_component(nc, 221, 222, seed=4, polys="mixed")
mesh = mg.with_nonmanifold(mesh, n_edges=max(1, mesh.ntri // 1000), n_vtx=max(1, mesh.ntri // 2000))
m0 = hc.Mesh.from_arrays(mesh.verts, mesh.degrees, mesh.indices)
cx = hc.Codec(0)
cx.upload(m0)
ref = None
for ch in sizes:
    best = None
    for it in range(3):
        m = m0.clone(); cx.upload(m)
        t = time.time(); out = cx.write_hry(m, profile=hc.PROFILE_CHUNKED, chunk_syms=ch, as_buffer=True); te = time.time() - t
        tme = cx.timing()
        t = time.time(); dec = cx.read_hry(out); td = time.time() - t
        tmd = cx.timing()
        rec = (te + td, te, td, tme, tmd)
        if best is None or rec[0] < best[0]:
            best = rec
    if ref is None:
        ref = dec
    else:
        assert np.array_equal(dec.list_data(1), ref.list_data(1)) and np.array_equal(dec.org(), ref.org())
    _, te, td, tme, tmd = best
    print(f"chunk {ch:7d}: {len(out):11d} bytes ({8*len(out)/mesh.nv:.3f} bpv) encode {te*1e3:7.1f} ms (k_entropy {tme.get('k_entropy_ms', 0):6.2f}) "
          f"decode {td*1e3:7.1f} ms (k_entropy {tmd.get('k_entropy_ms', 0):6.2f}, k_chain {tmd.get('k_chain_ms', 0):6.2f}, replay {tmd.get('host_walk_ms', 0):6.1f})", flush=True)
