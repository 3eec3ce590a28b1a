import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from harry_amd import codec as hc, meshgen as mg
mesh = mg.torus(708, 708, seed=2, sigma=1e-4)
m0 = hc.Mesh.from_arrays(mesh.verts, mesh.degrees, mesh.indices)
cx = hc.Codec(0)
cx.requant(m0, [(1, -1, 14)])
for it in range(3):
    m = m0.clone(); cx.upload(m)
    out = cx.write_hry(m, profile=hc.PROFILE_CHUNKED)
    if it == 2: os.environ["HRY_TRACE"] = "1"
    t = time.time(); d = cx.read_hry(out); print("decode ms", (time.time() - t) * 1e3)
