"""Rehearsal of `bench.py --gpus N` on a box with fewer GPUs (the ranks share device 0 and talk gloo): wall clock, the peak of
the summed resident memory of the bench and all its ranks, and the JSON line -- python scripts/rehearse_ranks.py N [comps per GPU]
[steps] > record.json.  The process guard of the GPU boxes allows six processes on the card: N <= 6 there."""
import json, os, subprocess, sys, threading, time
import psutil
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n = int(sys.argv[1]); comps = int(sys.argv[2]) if len(sys.argv) > 2 else 128; steps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n), "--share-device", "--comps-per-gpu", str(comps), "--steps", str(steps), "--warmup", "1"]
t0 = time.time()
p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=root)
peak = {"sum_rss": 0, "max_one": 0, "n_procs": 0}
def watch():
    me = psutil.Process(p.pid)
    while p.poll() is None:
        try:
            procs = [me] + me.children(recursive=True)
            rss = []
            for q in procs:
                try: rss.append(q.memory_info().rss)
                except psutil.Error: pass
            if rss:
                peak["sum_rss"] = max(peak["sum_rss"], sum(rss)); peak["max_one"] = max(peak["max_one"], max(rss)); peak["n_procs"] = max(peak["n_procs"], len(rss))
        except psutil.Error:
            pass
        time.sleep(0.25)
th = threading.Thread(target=watch); th.start()
out, err = p.communicate()
th.join()
wall = time.time() - t0
lines = [l for l in out.splitlines() if l.startswith("{")]
rec = {"command": " ".join(cmd[1:]), "rc": p.returncode, "wall_s": round(wall, 1), "peak_sum_rss_gb": round(peak["sum_rss"] / 2**30, 2), "peak_one_process_rss_gb": round(peak["max_one"] / 2**30, 2),
       "processes": peak["n_procs"], "cpus_granted": open("/sys/fs/cgroup/cpu.max").read().split()[0] if os.path.exists("/sys/fs/cgroup/cpu.max") else None,
       "line": json.loads(lines[-1]) if lines else None, "stderr_tail": err[-1500:] if p.returncode else ""}
print(json.dumps(rec))
