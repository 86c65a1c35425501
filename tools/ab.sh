#!/bin/bash
# A/B of two builds of libmpshuffle.so on ONE MI355X box (boxes of the pool differ by up to 13 %, so versions must be compared
# back to back inside a single gpurun call).  Usage, from the repo root:
#   1. build variant A, `cp mental-poker_amd/libmpshuffle.so tools/ab/lib_A.so`; same for B (tools/ab/ is git-ignored but
#      travels with the snapshot);
#   2. gpurun -- 'bash tools/ab.sh A B'
# Alternates the variants twice and prints proofs/s and the main kernel times of bench.py's default run.
set -e
for round in 1 2; do
  for v in "$@"; do
    cp tools/ab/lib_$v.so mental-poker_amd/libmpshuffle.so
    python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); k = d['roofline']['kernels_ms']
print('%-10s %7d proofs/s %7.1f ms/step  var %.1f  table %.1f  fixed %.1f  remask %.1f' % (sys.argv[1], d['value'], d['ms_per_step'], k['k_var_msm'], k['k_table'], k['k_fixed_msm'], k['k_remask']))" $v
  done
done
