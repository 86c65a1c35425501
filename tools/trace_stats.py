"""Per-kernel statistics of the batch-sized launches in a `rocprofv3 --kernel-trace --output-format csv` trace
(the plain --stats table also averages the handful of tiny setup-time launches of the same kernels).
usage: python tools/trace_stats.py <kernel_trace.csv> <batch> <out.csv>"""
import collections
import csv
import re
import sys

path, batch, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
d = collections.defaultdict(list)
for r in csv.DictReader(open(path)):
    if int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) < batch:
        continue
    m = re.search(r"mp::(k_\w+?)<", r["Kernel_Name"])
    d[m.group(1) if m else r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
tot = sum(sum(v) for v in d.values())
with open(out, "w") as f:
    f.write("# launches with grid >= %d threads (one lane per proof x jobs); durations in ns\n" % batch)
    f.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs\n")
    for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
        f.write("%s,%d,%d,%.1f,%.2f,%d,%d\n" % (k, len(v), sum(v), sum(v) / len(v), 100.0 * sum(v) / tot, min(v), max(v)))
print(open(out).read())
