"""cfg4-like run: many mixed-polygon components with non-manifold edges/vertices, lossless, chunked encode + decode,
verified against the CPU oracle (optional).  python tests/tools/cfg4_check.py NCOMP NU NV [--no-verify] [--contexts N] [--compat] [--profile-run]
--contexts N: additionally the in-process executor (hry_encode_sharded / hry_decode_sharded) with N contexts on device 0 -- the
merged container must equal the shard-by-shard (virtual rank) result and decode to the same mesh.
--compat: additionally the reference's single stream (.hry v0.1) of the whole mesh, compared with the oracle's bytes."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from harry_amd import codec as hc, meshgen as mg
nc, nu, nv = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
t = time.time()
mesh = mg.multi_component(nc, nu, nv, seed=4, polys="mixed")
mesh = mg.with_nonmanifold(mesh, n_edges=max(1, mesh.ntri // 1000), n_vtx=max(1, mesh.ntri // 2000))
print(f"mesh: {mesh.ntri} tris, {mesh.nf} faces, {mesh.nv} verts, built in {time.time()-t:.1f}s", flush=True)
t = time.time()
m0 = hc.Mesh.from_arrays(mesh.verts, mesh.degrees, mesh.indices)
print(f"half-edge build {time.time()-t:.1f}s", flush=True)
cx = hc.Codec(0)
t = time.time(); cx.upload(m0); print(f"twins matched on the device (first upload of a freshly built mesh) {time.time()-t:.2f}s", flush=True)   # every clone below has its twins
r = lambda tm: json.dumps({k: round(v, 1) if isinstance(v, float) else v for k, v in tm.items() if v})
for it in range(2):   # the first pass pays for module loading, stream creation and first-touch of the pools
    m = m0.clone(); cx.upload(m)
    t = time.time(); out = cx.write_hry(m, profile=hc.PROFILE_CHUNKED, as_buffer=True); te = time.time() - t
    print(f"pass {it}: encode {te*1e3:.0f} ms {mesh.ntri/te/1e6:.1f} Mtri/s bytes {len(out)} bpv {8*len(out)/mesh.nv:.2f} " + r(cx.timing()), flush=True)
    t = time.time(); dec = cx.read_hry(out); td = time.time() - t
    print(f"pass {it}: decode {td*1e3:.0f} ms {mesh.ntri/td/1e6:.1f} Mtri/s " + r(cx.timing()), flush=True)
# the same from a mesh in host memory (upload inside the timed call): what the in-process executor below is to be compared with
# (--profile-run: left out, so that a profile of this command holds exactly two encodes and two decodes)
for it in range(0 if "--profile-run" in sys.argv else 2):
    m = m0.clone()
    t = time.time(); out2 = cx.write_hry(m, profile=hc.PROFILE_CHUNKED, as_buffer=True); te = time.time() - t
    print(f"from the host mesh, pass {it}: encode {te*1e3:.0f} ms {mesh.ntri/te/1e6:.1f} Mtri/s " + r(cx.timing()), flush=True)
if "--profile-run" not in sys.argv:
    assert out2 == out
    del out2
def opt(name, default=0):
    return int(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else default
nctx = opt("--contexts")
if nctx:
    mc = hc.MultiCodec([0] * nctx)
    for it in range(2):
        m = m0.clone()
        t = time.time(); merged = mc.write_hry(m, as_buffer=True); te = time.time() - t
        print(f"in-process x{nctx} pass {it}: encode {te*1e3:.0f} ms {mesh.ntri/te/1e6:.1f} Mtri/s bytes {len(merged)} " + r(mc.last), flush=True)
        t = time.time(); mdec = mc.read_hry(merged); td = time.time() - t
        print(f"in-process x{nctx} pass {it}: decode {td*1e3:.0f} ms {mesh.ntri/td/1e6:.1f} Mtri/s " + r(mc.last), flush=True)
    same = np.array_equal(mdec.org(), dec.org()) and np.array_equal(mdec.list_data(1), dec.list_data(1)) and np.array_equal(mdec.face_offsets(), dec.face_offsets()) and np.array_equal(mdec.twin(), dec.twin())
    print(f"in-process x{nctx}: merged container decodes to the single-context mesh: {'OK' if same else 'MISMATCH'}", flush=True)
    assert same
    # the same shards one after the other on ONE context (what N ranks would each do), merged: byte for byte the same container
    from harry_amd import sharding
    t = time.time()
    whole = m0.clone()
    plan = hc.ShardPlan(whole, nctx)
    shards = [plan.extract(whole, s) for s in range(nctx)]
    tabs = [sharding.shard_bounds(cx, sh) for sh in shards]
    parts = []
    for sh in shards:
        sharding.combine_bounds(tabs, sh)
        parts.append(cx.write_hry(sh, profile=hc.PROFILE_CHUNKED))
    same = hc.merge(parts) == merged
    print(f"in-process x{nctx}: container equals the {nctx} virtual ranks' merged container: {'OK' if same else 'MISMATCH'} ({time.time()-t:.1f}s)", flush=True)
    assert same
    del shards, parts, whole, plan, mdec
    mc.close()
compat = None
if "--compat" in sys.argv:
    for it in range(2):
        m = m0.clone(); cx.upload(m)
        t = time.time(); compat = cx.write_hry(m, profile=hc.PROFILE_COMPAT, as_buffer=True); te = time.time() - t
        print(f"compat pass {it}: encode {te*1e3:.0f} ms {mesh.ntri/te/1e6:.1f} Mtri/s bytes {len(compat)} ({8*len(compat)/1e9:.2f} Gbit of the 4.29 Gbit the 32-bit positions hold) " + r(cx.timing()), flush=True)
    t = time.time(); cdec = cx.read_hry(compat); td = time.time() - t
    print(f"compat decode {td*1e3:.0f} ms {mesh.ntri/td/1e6:.1f} Mtri/s " + r(cx.timing()), flush=True)
    same = np.array_equal(cdec.org(), dec.org()) and np.array_equal(cdec.list_data(1), dec.list_data(1))
    print(f"compat decode equals the chunked decode: {'OK' if same else 'MISMATCH'}", flush=True)
    assert same
    del cdec
if "--no-verify" not in sys.argv:
    from oracle import oracle_py as op
    t = time.time()
    o = op.Mesh.from_ply(mesh.to_ply())
    ob = o.encode().data
    if compat is not None:
        print(f"compat stream byte-identical to the oracle's: {'OK' if ob == compat else 'MISMATCH'}", flush=True)
        assert ob == compat
    ref = op.Mesh.from_hry(ob)
    ok = np.array_equal(dec.org(), ref.org()) and np.array_equal(dec.list_data(1), ref.list_data(1)) and np.array_equal(dec.face_offsets(), ref.face_offsets())
    print(f"oracle check {'OK' if ok else 'MISMATCH'} ({time.time()-t:.1f}s)")
    if not ok:
        a, b = dec.list_data(1).view(np.uint32).reshape(-1, 3), ref.list_data(1).view(np.uint32).reshape(-1, 3)
        bad = np.flatnonzero((a != b).any(axis=1))
        print(f"  {len(bad)} vertex records differ; first {bad[:12].tolist()} last {bad[-4:].tolist()}; columns {np.flatnonzero((a != b).any(axis=0)).tolist()}")
        for v in bad[:6]:
            print(f"    v {v}: got {a[v].tolist()} want {b[v].tolist()}")
        d = np.diff(bad)
        print(f"  gaps between bad vertices: min {d.min() if len(d) else 0}, runs of consecutive: {int((d == 1).sum())}")
    assert ok
