"""Development: does the chain (three lone workgroups) run faster while the rest of the GPU is busy (clock / power state)?"""
import sys, time, json, threading
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from harry_amd import codec as hc, meshgen as mg

mesh = mg.torus(708, 708, seed=2, sigma=1e-4)
m0 = hc.Mesh.from_arrays(mesh.verts, mesh.degrees, mesh.indices)
cx = hc.Codec(0)
cx.requant(m0, [(1, -1, 14)])
m = m0.clone(); cx.upload(m)
out = cx.write_hry(m, profile=hc.PROFILE_CHUNKED)
def dec(tag):
    for it in range(4):
        t = time.time(); cx.read_hry(out); dt = time.time() - t
        tm = cx.timing()
        print(tag, f"dec {dt*1e3:.2f} ms chain {tm.get('k_chain_ms', 0):.3f}")
dec("idle ")
stop = False
def load():
    a = torch.randn(4096, 4096, device="cuda", dtype=torch.bfloat16)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        while not stop:
            for _ in range(20): a @ a
            s.synchronize()
th = threading.Thread(target=load); th.start()
time.sleep(1.0)
dec("busy ")
stop = True; th.join()
time.sleep(0.5)
dec("idle2")
