import csv, glob, sys
fs = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
if not fs:
    print("no stats under", sys.argv[1]); sys.exit(0)
for r in list(csv.DictReader(open(fs[0])))[:int(sys.argv[2]) if len(sys.argv) > 2 else 14]:
    print(f"{r['Name'][:56]:56s} calls={r['Calls']:>4s} avg_us={float(r['AverageNs'])/1e3:12.1f} pct={r['Percentage']}")
