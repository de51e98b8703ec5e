"""kernel durations and gaps between consecutive kernels from a rocprofv3 --kernel-trace --output-format csv directory"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-int(sys.argv[2]) if len(sys.argv) > 2 else -1200:]
gaps, dur = collections.defaultdict(list), collections.defaultdict(list)
short = lambda s: s.replace("mi355ntt::", "").replace("(anonymous namespace)::", "")[:34]
for p, c in zip(rows, rows[1:]):
    gaps[(short(p["Kernel_Name"]), short(c["Kernel_Name"]))].append(int(c["Start_Timestamp"]) - int(p["End_Timestamp"]))
for r in rows:
    dur[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in dur.items():
    print("dur %-36s n=%4d avg %8.2f us" % (k, len(v), sum(v) / len(v) / 1e3))
for k, v in gaps.items():
    print("gap %-36s -> %-36s n=%4d avg %6.2f us" % (k[0], k[1], len(v), sum(v) / len(v) / 1e3))
