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


for tag in ("kstats", "kstats_compat"):
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
print("== PMC means per launch (FETCH_SIZE / WRITE_SIZE in KiB as reported; gfx950: double FETCH_SIZE for wide streaming reads)")
for k, cs in summary.items():
    print(f"  {k[:50]:50s} " + "  ".join(f"{c}={v['mean']:.4g}" for c, v in sorted(cs.items())))
