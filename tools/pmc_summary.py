"""Turn two rocprofv3 counter passes (--pmc FETCH_SIZE, --pmc WRITE_SIZE; separate runs of the same bench command,
`--output-format csv`) into profiles/<round>_pmc_summary.json: HBM bytes per proof per step for every kernel whose grid
covers the batch.  Correction per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): the counters are in KB and
FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950 -> bytes = (2 * FETCH + WRITE) * 1024.

usage: python tools/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <bench.json> <out.json>
(<bench.json> = the JSON line the profiled bench command printed: batch, steps and the launches of the untimed priming prove)
"""
import collections
import csv
import json
import re
import sys


def per_kernel(path, counter, batch, skip):
    """sum of `counter` per kernel over the batch-sized dispatches, in dispatch order, skipping the first skip[kernel]
    (the priming prove) -> (totals, dispatch counts)"""
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter and int(r["Grid_Size"]) >= batch]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))     # setup-time launches (tables, tiny grids) are filtered by the grid size
    tot, cnt, seen = collections.Counter(), collections.Counter(), collections.Counter()
    for row in rows:
        m = re.search(r"mp::(k_\w+?)<", row["Kernel_Name"]) or re.search(r"(k_\w+)", row["Kernel_Name"])
        name = m.group(1) if m else row["Kernel_Name"]
        seen[name] += 1
        if seen[name] <= skip.get(name, 0):
            continue
        tot[name] += float(row["Counter_Value"])
        cnt[name] += 1
    return tot, cnt


def main():
    fetch_csv, write_csv, bench_json, out = sys.argv[1:5]
    bench = json.loads(open(bench_json).read().strip().splitlines()[-1])
    batch, passes = bench["config"]["proofs_per_gpu_per_step"], bench["steps"] + bench["warmup"]
    skip = bench["roofline"]["priming_launches"]
    f, fc = per_kernel(fetch_csv, "FETCH_SIZE", batch, skip)
    w, _ = per_kernel(write_csv, "WRITE_SIZE", batch, skip)
    kernels = {}
    for k in sorted(f, key=lambda k: -(2 * f[k] + w[k])):
        kernels[k] = {
            "dispatches": fc[k],
            "fetch_kb_total": f[k], "write_kb_total": w[k],
            "hbm_bytes_per_proof_per_step_corrected": (2 * f[k] + w[k]) * 1024 / batch / passes,
            "hbm_bytes_per_proof_per_step_raw": (f[k] + w[k]) * 1024 / batch / passes,
        }
    json.dump({"_note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes of the same bench command; counters in KB; "
                        "corrected = (2*FETCH + WRITE) KB (gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x, "
                        "MI355X_MICROARCH.md); per proof per prove+verify step",
               "batch": batch, "steps_equivalent": passes, "kernels": kernels}, open(out, "w"), indent=1)
    for k, v in kernels.items():
        print("%-16s %8.1f KB/proof/step (corrected)" % (k, v["hbm_bytes_per_proof_per_step_corrected"] / 1024))


if __name__ == "__main__":
    main()
