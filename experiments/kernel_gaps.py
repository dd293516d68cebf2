import csv, glob, sys, re, collections
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = lambda r: re.search(r'(\w+_kernel)', r['Kernel_Name']).group(1) if re.search(r'(\w+_kernel)', r['Kernel_Name']) else r['Kernel_Name'][:30]
gaps = collections.defaultdict(list)
for a, b in zip(rows, rows[1:]):
    g = int(b['Start_Timestamp']) - int(a['End_Timestamp'])
    gaps[(names(a), names(b))].append(g)
for k, v in sorted(gaps.items(), key=lambda kv: -len(kv[1]))[:8]:
    v.sort()
    print(k, len(v), 'median gap ns', v[len(v)//2])
dur = collections.defaultdict(list)
for r in rows:
    dur[names(r)].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for k, v in dur.items():
    v.sort(); print(k, len(v), 'median dur ns', v[len(v)//2])
