import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
dur = collections.defaultdict(list); gap = collections.defaultdict(list)
prev = None
for r in rows:
    nm = r['Kernel_Name'].split('(')[0].replace('void mi355ntt::','')
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    dur[nm].append(e - s)
    if prev: gap[(prev[0], nm)].append(s - prev[1])
    prev = (nm, e)
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    v2 = sorted(v); print("%-40s n=%5d  median %7.2f us  min %7.2f" % (k, len(v), v2[len(v)//2] / 1e3, v2[0] / 1e3))
print("gaps (end of previous -> start of next), median us:")
for k, v in sorted(gap.items(), key=lambda kv: -len(kv[1]))[:14]:
    v2 = sorted(v); print("  %-34s -> %-34s n=%5d median %6.2f min %6.2f" % (k[0], k[1], len(v), v2[len(v)//2] / 1e3, v2[0] / 1e3))
