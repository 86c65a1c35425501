#!/usr/bin/env python3
"""Recompute the roofline figures of a bench line from the files under profiles/ alone.

  python tools/roofline_report.py r02E_default

reads profiles/<TAG>_bench.json (the JSON line of `python bench.py <args>`), profiles/<TAG>_kernel_stats.csv (rocprofv3
--kernel-trace of the same command) and profiles/<TAG>_pmc_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ passes) and
prints, for the dominant kernel: algorithmic bytes per launch, average launch time from the bench's HIP events and from the trace,
achieved GB/s and its fraction of the 8 TB/s HBM peak, the HBM traffic per launch from the counters, and the multiply-add count of
one prove+verify rebuilt from the static plan x the per-operation counts (mental-poker_amd/mad_counts.json as recorded in the line).
"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tag = sys.argv[1]
    P = lambda n: os.path.join(ROOT, "profiles", tag + n)
    d = json.load(open(P("_bench.json")))
    r = d["roofline"]
    k = r["kernel"]
    print("bench line      : %.0f %s, %.1f ms/step, %d steps, workload %s" % (d["value"], d["unit"], d["ms_per_step"], d["steps"], d["config"]["workload"]))
    print("dominant kernel : %s, %d launches in the timed region, %.3f ms average (HIP events in bench.py)" % (k, r["launches"], r["avg_launch_ms"]))
    if os.path.exists(P("_kernel_stats.csv")):
        for row in csv.DictReader(l for l in open(P("_kernel_stats.csv")) if not l.startswith("#")):
            if row["Name"] == k:
                print("                  rocprofv3 kernel trace of the same command: %s calls, %.3f ms average" % (row["Calls"], float(row["AverageNs"]) / 1e6))
    ach = r["alg_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9
    print("algorithmic     : %.0f B per proof (scalars 32 B + points + results of this kernel class) -> %.3f GB per launch" % (r["alg_bytes_per_proof"], r["alg_bytes_per_launch"] / 1e9))
    print("achieved        : %.2f GB/s = %.5f of the %.0f GB/s HBM peak   (line says frac %.5f)" % (ach, ach / r["peak"], r["peak"], r["frac"]))
    if os.path.exists(P("_pmc_summary.json")):
        pmc = json.load(open(P("_pmc_summary.json")))
        e = pmc["kernels"].get(k)
        if e:
            per = e.get("hbm_bytes_per_proof_per_step", e["hbm_bytes_per_proof_per_step_corrected"])
            launches_per_step = r["launches"] / d["steps"]
            print("HBM traffic     : %.1f KB per proof and step from FETCH_SIZE x %.0f + WRITE_SIZE (batch %d) -> %.2f GB per launch, %.2f TB/s while the kernel runs"
                  % (per / 1024, e.get("fetch_calibration", 2.0), pmc["batch"], per * pmc["batch"] / launches_per_step / 1e9,
                     per * pmc["batch"] / launches_per_step / (r["avg_launch_ms"] * 1e-3) / 1e12))
            print("                  (summary taken from engine sources %s; the line was produced by %s)" % (pmc.get("engine_src"), r.get("engine_src")))
    im = r.get("int_mul")
    if im and "plan_stats" in im:
        st, mo, bw = im["plan_stats"], im["mads_per_op"], im.get("bucket_windows", 32)
        N, vw, fw = st["N"], st["var_windows"], st["fixed_windows"]
        m_guess = None
        tot = 0
        for side in ("prove", "verify"):
            s_ = st[side]
            fixed = s_["fixed_terms"] * fw
            if side == "prove":
                fixed += -N * (fw - 1) + 2 * N * (fw + 1)
            tot += (fixed + s_["var_terms"] * vw) * mo["madd"] + s_["var_jobs"] * (vw - 1) * 5 * mo["dbl"] + s_["table_bases"] * 15 * mo["aff"] + s_["combine_terms"] * mo["jac"]
        fm = im["mads_per_field_op"]
        tot += st.get("bucket_terms", 0) * bw * mo["madd"] + st.get("bucket_jobs", 0) * bw * (14 * 64 * (12 * fm["mul"] + 2 * fm["sqr"]) + 8 * mo["dbl"] + mo["jac"])
        print("multiply-adds   : %.2f M per prove+verify from the plan (line: %.2f M incl. normalisation of the outputs); %.0f Gmad/s = %.3f of the %.0f Gmad/s issue peak"
              % (tot / 1e6, im["mads_per_proof"] / 1e6, im["achieved"], im["frac"], im["peak"]))
    print("kernel time     : " + ", ".join("%s %.1f" % (a, b / d["steps"]) for a, b in list(r["kernels_ms"].items())[:6]) + "  (ms per step)")


if __name__ == "__main__":
    main()
