import csv, sys, glob
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# find the last decode: last k_chunk_decode launch group
idx = [i for i, r in enumerate(rows) if 'k_chunk_decode' in r['Kernel_Name']]
# group consecutive decodes: take the last conn decode (first of last cluster)
last = idx[-1]
t0 = None
# cluster start: walk back while gap < 5 ms
i = last
while i > 0 and int(rows[i]['Start_Timestamp']) - int(rows[i-1]['End_Timestamp']) < 3_000_000: i -= 1
t0 = int(rows[i]['Start_Timestamp'])
for r in rows[i:]:
    n = r['Kernel_Name'].split('(')[0].replace('hry::dev::', '').replace('void ', '')
    print(f"{(int(r['Start_Timestamp'])-t0)/1e6:8.3f} -> {(int(r['End_Timestamp'])-t0)/1e6:8.3f} ms  q{r.get('Queue_Id','?')}  {n[:50]}  grid {r.get('Grid_Size','')}")
