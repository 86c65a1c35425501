#!/usr/bin/env python3
"""tools/trace_overlap.py -- how much of a run the GPU spent with 0 / 1 / 2+ kernels in flight, from a rocprofv3 kernel trace.

  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tr -- python tools/plan_sweep.py ...
  python tools/trace_overlap.py gpurun_out/tr [--last-ms 200]

Prints the union of kernel intervals (busy time), the time with at least two kernels running, per-queue busy time and the longest
kernels -- the numbers behind "does the second lane of streams actually overlap" (small batches).
"""
import csv
import glob
import os
import sys


def main():
    d = sys.argv[1]
    last_ms = None
    if "--last-ms" in sys.argv:
        last_ms = float(sys.argv[sys.argv.index("--last-ms") + 1])
    only = sys.argv[sys.argv.index("--only") + 1] if "--only" in sys.argv else "mp::"      # the engine's kernels (not torch's)
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        raise SystemExit("no *kernel_trace.csv under " + d)
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            if only and only not in r["Kernel_Name"]:
                continue
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"].split("(")[0][:48]))
    rows.sort()
    t_end = max(r[1] for r in rows)
    if last_ms is not None:
        rows = [r for r in rows if r[0] >= t_end - last_ms * 1e6]
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    ev = []
    for s, e, q, n in rows:
        ev.append((s, 1))
        ev.append((e, -1))
    ev.sort()
    depth, prev, hist = 0, t0, {}
    for t, dlt in ev:
        hist[depth] = hist.get(depth, 0) + (t - prev)
        prev = t
        depth += dlt
    wall = t1 - t0
    print("kernels %d  wall %.3f ms  sum of kernel times %.3f ms" % (len(rows), wall / 1e6, sum(e - s for s, e, _, _ in rows) / 1e6))
    for k in sorted(hist):
        print("  %d kernel(s) in flight: %8.3f ms  %5.1f %%" % (k, hist[k] / 1e6, 100.0 * hist[k] / wall))
    perq = {}
    for s, e, q, n in rows:
        perq[q] = perq.get(q, 0) + (e - s)
    for q in sorted(perq):
        print("  queue %s: %.3f ms of kernels" % (q, perq[q] / 1e6))
    byname = {}
    for s, e, q, n in rows:
        c, t = byname.get(n, (0, 0))
        byname[n] = (c + 1, t + e - s)
    for n, (c, t) in sorted(byname.items(), key=lambda kv: -kv[1][1])[:14]:
        print("  %-48s %5d x %8.1f us" % (n, c, t / c / 1e3))


if __name__ == "__main__":
    main()
