"""cfg3-like run (BASELINE configs[2] stand-in): closed torus 3742 x 3742 -> 28 005 128 triangles, float32 positions + analytic
normals, positions 14 bits / normals 10 bits (`-l1 -a0 -q14 -a1 -q14 -a2 -q14 -a3 -q10 -a4 -q10 -a5 -q10`, SURVEY 8d flag
caveat), chunked encode + decode on one GPU.  python tests/tools/cfg3_check.py [SIDE] [--oracle]
Checks: order-independent invariants of the decoded mesh against the quantised input (per-component sums and xor-folds,
face degree histogram); --oracle additionally decodes with the CPU oracle (slow at full size)."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from harry_amd import codec as hc, meshgen as mg

side = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 3742
t = time.time()
mesh = mg.torus(side, side, seed=3, sigma=1e-4, normals=True)
print(f"mesh: {mesh.ntri} tris, {mesh.nv} verts, built in {time.time()-t:.1f}s", flush=True)
t = time.time()
m0 = hc.Mesh.from_arrays(mesh.verts, mesh.degrees, mesh.indices)
print(f"half-edge build {time.time()-t:.1f}s", flush=True)
cx = hc.Codec(0)
quant = [(1, 0, 14), (1, 1, 14), (1, 2, 14), (1, 3, 10), (1, 4, 10), (1, 5, 10)]
r = lambda tm: json.dumps({k: round(v, 1) if isinstance(v, float) else v for k, v in tm.items() if v})
alg = None
for it in range(2):
    m = m0.clone(); cx.upload(m)
    t = time.time(); cx.requant(m, quant); tq = time.time() - t
    t = time.time(); out = cx.write_hry(m, profile=hc.PROFILE_CHUNKED); te = time.time() - t
    alg = m.nv * m.list_stride(1) + 4 * m.ne + len(out)
    print(f"encode {te*1e3:.0f} ms (+ quantisation {tq*1e3:.0f} ms) {mesh.ntri/(te+tq)/1e6:.1f} Mtri/s, {alg/(te+tq)/1e9:.2f} GB/s algorithmic; bytes {len(out)} "
          f"bits/vertex {8*len(out)/mesh.nv:.2f} " + r(cx.timing()), flush=True)
    qrec = m.list_data(1).copy()
    t = time.time(); dec = cx.read_hry(out); td = time.time() - t
    print(f"decode {td*1e3:.0f} ms {mesh.ntri/td/1e6:.1f} Mtri/s, {alg/td/1e9:.2f} GB/s algorithmic " + r(cx.timing()), flush=True)
# invariants (the decoder renumbers vertices in coding order)
# a quantised value occupies the low bytes of its 4-byte slot; the encoder-side in-place quantisation leaves the rest of the
# slot as it was (like the reference), the decoder zeroes it: compare the values
a = np.ascontiguousarray(qrec.reshape(mesh.nv, -1).view(np.uint16)[:, ::2]).astype(np.uint32)
b = np.ascontiguousarray(dec.list_data(1).reshape(mesh.nv, -1).view(np.uint16)[:, ::2]).astype(np.uint32)
ok = (dec.nv, dec.nf, dec.ne) == (m0.nv, m0.nf, m0.ne)
ok &= bool(np.array_equal(a.astype(np.uint64).sum(0), b.astype(np.uint64).sum(0)))
ok &= bool(np.array_equal(np.bitwise_xor.reduce(a, 0), np.bitwise_xor.reduce(b, 0)))
# the same multiset of records: sort whole records (packed into one integer per half)
ka = (a[:, 0].astype(np.uint64) << 28) | (a[:, 1].astype(np.uint64) << 14) | a[:, 2]
kb = (b[:, 0].astype(np.uint64) << 28) | (b[:, 1].astype(np.uint64) << 14) | b[:, 2]
na = (a[:, 3].astype(np.uint64) << 20) | (a[:, 4].astype(np.uint64) << 10) | a[:, 5]
nb = (b[:, 3].astype(np.uint64) << 20) | (b[:, 4].astype(np.uint64) << 10) | b[:, 5]
ok &= bool(np.array_equal(np.sort((ka << 30) ^ na), np.sort((kb << 30) ^ nb)))
print("round-trip invariants", "OK" if ok else "MISMATCH", flush=True)
assert ok
if "--oracle" in sys.argv:
    from oracle import oracle_py as op
    t = time.time()
    ref = op.Mesh.from_hry_chunked(out)
    same = np.array_equal(dec.org(), ref.org()) and np.array_equal(dec.list_data(1), ref.list_data(1)) and np.array_equal(dec.face_offsets(), ref.face_offsets())
    print(f"oracle decode of the GPU's container {'OK' if same else 'MISMATCH'} ({time.time()-t:.1f}s)")
    assert same
