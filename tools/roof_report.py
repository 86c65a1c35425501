#!/usr/bin/env python3
"""tools/microbench/roof (gpurun_out/<TAG>/roof.txt) -> profiles/<name>.json + a table on stdout.

Per variant: instructions (or field / group operations) per second from the HIP-event wall time of a >= 200 ms kernel that fills the
chip, the same figure as lanes per clock per SIMD at the clock measured inside the kernel (s_memtime / s_memrealtime), SIMD cycles per
wave-level instruction, the sysfs shader clock and the board power during the kernel's second half.  The VALU issue classes the
engine's roofline uses (bench.py roofline.compute, tools/gen_mad_counts.py) are read off the `c_*` rows: >= 24 lanes/clk/SIMD =
full rate (2 cycles per wave64 instruction), otherwise half rate (4 cycles)."""
import json
import sys


def main():
    src, out = sys.argv[1], sys.argv[2]
    rows = []
    hdr = []
    for line in open(src):
        if line.startswith("JSON "):
            d = json.loads(line[5:])
            if "discard" in d["variant"]:
                continue
            clk = d["clock_mhz_memtime"] * 1e6
            lanes = d["G_units_per_s"] * 1e9 / (1024 * clk)
            d["lanes_per_clk_per_simd"] = round(lanes, 2)
            d["simd_cycles_per_wave_unit_wall"] = round(64.0 / lanes, 2) if lanes > 0 else None
            if d["variant"].startswith("c_") or d["variant"] == "v_fma_f64":
                d["issue_class"] = "full_rate (2 cycles)" if lanes >= 24 else "half_rate (4 cycles)"
            rows.append(d)
        elif line.startswith(("device:", "telemetry:")):
            hdr.append(line.strip())
    full = sorted(r["variant"][2:] for r in rows if r.get("issue_class", "").startswith("full"))
    half = sorted(r["variant"][2:] if r["variant"].startswith("c_") else r["variant"] for r in rows if r.get("issue_class", "").startswith("half"))
    doc = {"_note": "tools/microbench/roof.hip on one MI355X (sustained kernels, occ = waves per SIMD the grid is sized for); G_units_per_s from "
                    "HIP-event wall time; lanes_per_clk_per_simd = units/s / (1024 SIMDs x clock measured in the kernel); "
                    "simd_cycles_per_unit = per-wave s_memtime cycles / occ (the waves of a SIMD do not all live for the whole kernel, so the "
                    "wall-based column is the one to use for rates); power = board power from the device's own hwmon",
           "header": hdr, "full_rate_instructions": full, "half_rate_instructions": half, "rows": rows}
    json.dump(doc, open(out, "w"), indent=1)
    print("%-34s %3s %9s %12s %8s %8s %7s %7s" % ("variant", "occ", "ms", "G units/s", "lanes/clk", "cyc/unit", "MHz", "W"))
    for r in rows:
        print("%-34s %3d %9.1f %12.2f %8.2f %8.2f %7.0f %7.0f" % (r["variant"], r["occ"], r["ms"], r["G_units_per_s"], r["lanes_per_clk_per_simd"],
                                                                   r["simd_cycles_per_wave_unit_wall"] or 0, r["clock_mhz_memtime"], r["power_w"]))
    print("full rate:", ", ".join(full))
    print("half rate:", ", ".join(half))


if __name__ == "__main__":
    main()
