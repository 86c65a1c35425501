#!/bin/bash
# round 5: window width x group size of the verifier's group equation on the re-laid-out bucket kernel, one box, back to back;
# then the kernel's phases (MP_EXP_BK_TIMING build) for the default and the 11-bit configuration
cp mental-poker_amd/libmpshuffle.so /tmp/lib_keep.so
one() {
  python bench.py --no-cpu-baseline --no-extras --steps 3 --group-points $1 --bucket-bits $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]);k=d['roofline']['kernels_ms'];s=d['steps']
print('points %s bits %s: %d proofs/s  bucket %.1f tile %.1f recode %.1f chainscal %.1f fold %.1f' % (sys.argv[1],sys.argv[2],int(d['value']),k.get('k_bucket_msm',0)/s,k.get('k_group_tile',0)/s,k.get('k_bucket_recode',0)/s,k.get('k_chain_scalars',0)/s,(k.get('k_bucket_fold',0)+k.get('k_bucket_fold_q',0))/s))" $1 $2
}
for cfg in "30464 10" "60928 11" "60928 10" "15232 9" "15232 10" "30464 10" "60928 11"; do one $cfg; done
if [ -f tools/ab/lib_timing.so ]; then
  cp tools/ab/lib_timing.so mental-poker_amd/libmpshuffle.so
  for cfg in "30464 10" "60928 11"; do
    set -- $cfg
    python bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 --group-points $1 --bucket-bits $2 2>&1 >/dev/null | grep k_bucket_msm | tail -1
  done
fi
cp /tmp/lib_keep.so mental-poker_amd/libmpshuffle.so
