"""Development: the device's time line of the last decode of a run recorded by `rocprofv3 --kernel-trace --memory-copy-trace`
(python scripts/trace_timeline.py DIR): kernels and copies with start / end relative to the decode's first kernel."""
import csv, sys, glob
rows = []
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'K', r['Kernel_Name'].split('(')[0].replace('hry::dev::', '').replace('void ', '')[:44], r.get('Queue_Id', '?')))
for f in glob.glob(sys.argv[1] + '/**/*memory_copy_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'C', r.get('Direction', '') + ' ' + r.get('Size', r.get('Bytes', '')), '-'))
rows.sort()
idx = [i for i, r in enumerate(rows) if r[2] == 'K' and 'k_chunk_decode' in r[3]]
i = idx[-1]
while i > 0 and rows[i][0] - rows[i - 1][1] < 2_000_000: i -= 1
t0 = rows[i][0]
for s, e, k, n, q in rows[i:]:
    print(f"{(s - t0) / 1e6:8.3f} -> {(e - t0) / 1e6:8.3f} ms  {k} q{q}  {n}")
