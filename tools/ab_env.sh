#!/bin/bash
# A/B of run-time variants (environment hooks) and builds on ONE MI355X box, back to back.  Usage, from the repo root:
#   gpurun -- 'bash tools/ab_env.sh "<bench args>" "name|lib|ENV=.. ENV=.." ...'
# lib = a name under tools/ab/lib_<name>.so.  Two rounds, alternating; prints proofs/s and the bucket kernel's ms per launch.
set -e
ARGS="$1"; shift
cp mental-poker_amd/libmpshuffle.so /tmp/lib_keep.so
for round in 1 2; do
  for v in "$@"; do
    name="${v%%|*}"; rest="${v#*|}"; lib="${rest%%|*}"; envs="${rest#*|}"
    cp tools/ab/lib_$lib.so mental-poker_amd/libmpshuffle.so
    env $envs python bench.py --no-cpu-baseline --no-extras $ARGS 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); k = d['roofline']['kernels_ms']; s = d['steps']
print('%-22s %8d proofs/s %7.1f ms/step  bucket %.1f  var %.1f  fixed %.1f  tile %.1f  recode %.1f' % (sys.argv[1], d['value'], d['ms_per_step'], k.get('k_bucket_msm', 0) / s, k['k_var_msm'] / s, k['k_fixed_msm'] / s, k.get('k_group_tile', 0) / s, k.get('k_bucket_recode', 0) / s))" "$name"
  done
done
cp /tmp/lib_keep.so mental-poker_amd/libmpshuffle.so
