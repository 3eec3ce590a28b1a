"""End to end at configs[3]'s size: a mixed-polygon PLY file -> `harry in.ply out.hry --gpus N --profile chunked` -> `harry out.hry back.ply`,
wall clock of each process (runtime start and code-object load included), phases as the command prints them.
    python tests/tools/cfg4_e2e.py NCOMP NU NV [--gpus N] [--dir /dev/shm] [--verify]
--verify: back.ply must hold the arrays the oracle decodes from its own stream of in.ply (slow: one core)."""
import os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from harry_amd import meshgen as mg

def opt(name, default):
    return type(default)(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else default
nc, nu, nv = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
gpus, d = opt("--gpus", 1), opt("--dir", "/dev/shm")
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
harry = os.path.join(root, "harry_amd", "bin", "harry")
t = time.time()
mesh = mg.multi_component(nc, nu, nv, seed=4, polys="mixed")
mesh = mg.with_nonmanifold(mesh, n_edges=max(1, mesh.ntri // 1000), n_vtx=max(1, mesh.ntri // 2000))
ply = mesh.to_ply()
src, hry, back = (os.path.join(d, f"cfg4_{os.getpid()}.{e}") for e in ("ply", "hry", "back.ply"))
with open(src, "wb") as f:
    f.write(ply)
print(f"mesh: {mesh.ntri} triangles, {mesh.nf} faces, {mesh.nv} vertices; {len(ply)} bytes of PLY written in {time.time()-t:.1f}s", flush=True)
try:
    for rep in range(2):   # the second pass finds the files in the page cache and the code objects in the runtime's cache
        for what, cmd in (("encode", [harry, src, hry, "--profile", "chunked"] + (["--gpus", str(gpus)] if gpus > 1 else [])),
                          ("decode", [harry, hry, back] + (["--gpus", str(gpus)] if gpus > 1 else []))):
            t = time.time()
            r = subprocess.run(cmd, capture_output=True, text=True)
            dt = time.time() - t
            if r.returncode != 0:
                print(r.stdout, r.stderr)
                raise SystemExit(f"{what} failed with status {r.returncode}")
            phases = "; ".join(l.strip() for l in r.stdout.splitlines() if "took" in l or "shard(s)" in l or "segment(s)" in l)
            if os.environ.get("HRY_TRACE"):
                print(r.stderr, flush=True)
            print(f"pass {rep}: {what} {dt*1e3:.0f} ms wall clock = {mesh.ntri/dt/1e6:.1f} Mtriangles/s end to end ({gpus} context(s)) | {phases}", flush=True)
        print(f"pass {rep}: {os.path.getsize(hry)} bytes of .hry, {os.path.getsize(back)} bytes of PLY back", flush=True)
    if "--verify" in sys.argv:
        from harry_amd import codec as hc
        from oracle import oracle_py as op
        t = time.time()
        got = hc.Mesh.from_ply(open(back, "rb").read())
        ref = op.Mesh.from_hry(op.Mesh.from_ply(ply).encode().data)
        ok = np.array_equal(got.org(), ref.org()) and np.array_equal(got.list_data(1), ref.list_data(1)) and np.array_equal(got.face_offsets(), ref.face_offsets())
        print(f"oracle check {'OK' if ok else 'MISMATCH'} ({time.time()-t:.1f}s)", flush=True)
        assert ok
finally:
    for f in (src, hry, back):
        if os.path.exists(f):
            os.remove(f)
