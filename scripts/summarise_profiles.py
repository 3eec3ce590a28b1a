"""Condenses the rocprofv3 CSV output of scripts/collect_profiles.sh into the small files committed under profiles/."""
import csv, glob, json, os, sys
from collections import defaultdict

out = sys.argv[1]


def first(pattern):
    fs = glob.glob(os.path.join(out, pattern), recursive=True)
    return fs[0] if fs else None


def short(name):
    name = name.split("(")[0]
    for pre in ("void hry::dev::", "hry::dev::"):
        if name.startswith(pre):
            name = name[len(pre):]
    return name


for tag in ("kstats",):
    f = first(f"{tag}/**/*kernel_stats.csv")
    if not f:
        print(tag, ": no kernel stats")
        continue
    rows = list(csv.DictReader(open(f)))
    with open(os.path.join(out, f"{tag}_kernel_stats.csv"), "w") as o:
        o.write("name,calls,total_ns,average_ns,percentage,min_ns,max_ns\n")
        for r in rows:
            o.write(f"\"{short(r['Name'])}\",{r['Calls']},{r['TotalDurationNs']},{float(r['AverageNs']):.0f},{r['Percentage']},{r['MinNs']},{r['MaxNs']}\n")
    print(f"== {tag}: top kernels")
    for r in rows[:12]:
        print(f"  {short(r['Name'])[:60]:60s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:10.1f} pct={r['Percentage']}")

# PMC: one row per dispatch and counter
traffic = defaultdict(lambda: defaultdict(list))
for tag in ("pmc_fetch", "pmc_write", "pmc_sq"):
    f = first(f"{tag}/**/*counter_collection.csv")
    if not f:
        print(tag, ": no counter file")
        continue
    for r in csv.DictReader(open(f)):
        traffic[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
summary = {}
for k, cs in sorted(traffic.items()):
    summary[k] = {c: {"launches": len(v), "mean": sum(v) / len(v)} for c, v in cs.items()}
with open(os.path.join(out, "pmc_summary.json"), "w") as o:
    json.dump(summary, o, indent=1, sort_keys=True)
# HBM-side bytes per PASS of the workload (one encode + one decode; a kernel that is launched once per slice or per level is
# summed over its launches): FETCH_SIZE / WRITE_SIZE are reported in KiB.  Raw counter values: the gfx950 correction (FETCH_SIZE
# tallies 128-byte requests at 64 bytes -> x2) is applied by the reader (bench.py), not here.
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3   # passes of scripts/quick_chunked.py
tj = {"passes": iters, "unit": "bytes per pass (KiB x 1024, uncorrected)", "kernels": {}}
for k, cs in sorted(traffic.items()):
    if "FETCH_SIZE" in cs or "WRITE_SIZE" in cs:
        tj["kernels"][k] = {"fetch_bytes": int(sum(cs.get("FETCH_SIZE", [])) * 1024 / iters), "write_bytes": int(sum(cs.get("WRITE_SIZE", [])) * 1024 / iters),
                            "launches_per_pass": len(cs.get("FETCH_SIZE", cs.get("WRITE_SIZE", []))) / iters}
with open(os.path.join(out, "traffic.json"), "w") as o:
    json.dump(tj, o, indent=1, sort_keys=True)
print("== PMC means per launch (FETCH_SIZE / WRITE_SIZE in KiB as reported; gfx950: double FETCH_SIZE for wide streaming reads)")
for k, cs in summary.items():
    print(f"  {k[:50]:50s} " + "  ".join(f"{c}={v['mean']:.4g}" for c, v in sorted(cs.items())))
