#!/usr/bin/env python3
"""Build an experimental variant of libmpshuffle.so for a back-to-back A/B on one GPU box (tools/ab.sh).

usage: tools/ab_build.py NAME [-DMACRO=...]... [--units curve_stark_msm.hip,curve_stark.hip]

Compiles the listed translation units (default: the STARK group-arithmetic unit) with the extra macros into tools/ab/obj_NAME/ and
links them with the product build's other objects into tools/ab/lib_NAME.so.  tools/ab/ is git-ignored but travels with gpurun.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import importlib
nat = importlib.import_module("mental-poker_amd._native")


def main():
    name = sys.argv[1]
    extra = [a for a in sys.argv[2:] if a.startswith("-D") or a.startswith("-mllvm") or a.startswith("-amdgpu")]
    units = ["curve_stark_msm.hip"]
    for a in sys.argv[2:]:
        if a.startswith("--units="):
            units = a.split("=", 1)[1].split(",")
    if "--no-product" not in sys.argv:
        nat.build()                              # product objects up to date
    csrc = os.path.join(ROOT, "mental-poker_amd", "csrc")
    objdir = os.path.join(ROOT, "tools", "ab", "obj_" + name)
    os.makedirs(objdir, exist_ok=True)
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-mllvm", "-amdgpu-sched-strategy=max-ilp",
             "-mllvm", "-pragma-unroll-threshold=1000000"] + extra    # same flags as mental-poker_amd/_native.py
    for a in sys.argv[2:]:
        if a.startswith("--sched="):      # another scheduling strategy for the variant's units (default: the product's max-ilp; "default" = LLVM's)
            i = flags.index("-amdgpu-sched-strategy=max-ilp")
            if a == "--sched=default":
                del flags[i - 1:i + 1]
            else:
                flags[i] = "-amdgpu-sched-strategy=" + a.split("=", 1)[1]

    def cc(u):
        obj = os.path.join(objdir, u.replace(".hip", ".o"))
        subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + flags + ["-c", os.path.join(csrc, u), "-o", obj])
        return obj
    with ThreadPoolExecutor(len(units)) as ex:
        mine = list(ex.map(cc, units))
    others = [os.path.join(csrc, "_obj", s.replace(".hip", ".o")) for s in nat.SOURCES if s not in units]
    out = os.path.join(ROOT, "tools", "ab", "lib_%s.so" % name)
    subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + mine + others)
    print(out)


if __name__ == "__main__":
    main()
