#!/usr/bin/env python3
"""tools/slot_trace.py -- where the empty wave slots of ONE small-batch step are, from a rocprofv3 kernel trace.

  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tr1024 -- python tools/plan_sweep.py --batches 1024 --configs default --seconds 0.05
  python tools/slot_trace.py gpurun_out/tr1024 > profiles/r06_slot_trace_1024.txt

Takes the last complete prove + verify step of the trace (from the end of one k_verdict* launch to the end of the next) and prints every
launch in start order: offset and duration in microseconds, the waves it launched, and the share of the chip's wave slots (1 024 SIMDs
x the waves per SIMD the kernel's register count allows) those waves can fill at most.  Then per kernel the time it held the step's
critical path, weighted by the slots it left empty: the serial sections that keep a 1 024-proof step at 60 % of the issue slots.
"""
import csv
import glob
import os
import sys

SLOTS = 4096.0


def main():
    files = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        raise SystemExit("no *kernel_trace.csv under " + sys.argv[1])
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            if "mp::" not in r["Kernel_Name"]:
                continue
            name = r["Kernel_Name"].split("mp::")[1].split("<")[0]
            grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
            vgpr = int(r.get("VGPR_Count", 128) or 128) + int(r.get("Accum_VGPR_Count", 0) or 0)
            per_simd = max(1, min(8, 512 // max(vgpr, 64)))      # waves per SIMD the kernel's registers allow
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, grid, 1024 * per_simd, r.get("Queue_Id", "?")))
    rows.sort()
    ends = [e for s, e, n, g, w, q in rows if n.startswith("k_verdict")]
    if len(ends) < 3:
        raise SystemExit("fewer than three steps in the trace")
    t0, t1 = ends[-3], ends[-2]          # (the last step may be the profiler's own tail)
    step = [r for r in rows if r[0] >= t0 and r[1] <= t1 + 1]
    print("one step of the trace: %.1f us wall, %d launches on %d queue(s)" % ((t1 - t0) / 1e3, len(step), len({r[5] for r in step})))
    print("%9s %9s  %-22s %9s %7s  %s" % ("start us", "dur us", "kernel", "waves", "fill", "queue"))
    per = {}
    for s, e, n, g, w, q in step:
        waves = (g + 63) // 64
        fill = min(1.0, waves / float(w))
        print("%9.1f %9.1f  %-22s %9d %7.2f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, n, waves, fill, q))
        c, t, idle = per.get(n, (0, 0.0, 0.0))
        per[n] = (c + 1, t + (e - s) / 1e3, idle + (e - s) / 1e3 * (1.0 - fill))
    # union of busy intervals -> gaps
    ev = sorted([(s, 1) for s, e, *_ in step] + [(e, -1) for s, e, *_ in step])
    depth, prev, gap, two = 0, t0, 0, 0
    for t, d in ev:
        if depth == 0:
            gap += t - prev
        if depth >= 2:
            two += t - prev
        prev = t
        depth += d
    gap += t1 - prev
    wall = (t1 - t0) / 1e3
    print("\nno kernel in flight: %.1f us (%.1f %%); two or more in flight: %.1f us" % (gap / 1e3, 100 * gap / 1e3 / wall, two / 1e3))
    print("\n%-22s %6s %10s %14s" % ("kernel", "calls", "time us", "empty-slot us"))
    for n, (c, t, idle) in sorted(per.items(), key=lambda kv: -kv[1][2]):
        print("%-22s %6d %10.1f %14.1f" % (n, c, t, idle))
    tot_t = sum(v[1] for v in per.values())
    tot_i = sum(v[2] for v in per.values())
    print("%-22s %6s %10.1f %14.1f   (slots a launch of its size cannot fill: %.0f %% of the kernel time)" % ("sum", "", tot_t, tot_i, 100 * tot_i / tot_t))


if __name__ == "__main__":
    main()
