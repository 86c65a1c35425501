#!/bin/bash
# A/B of builds of the split bucket pipeline on ONE box: verify-only steps, two rounds over the variants
cd "$GRAFT_REPO_ROOT" || exit 1
cp mental-poker_amd/libmpshuffle.so /tmp/lib_keep.so
for round in 1 2; do
  for v in "$@"; do
    cp tools/ab/lib_$v.so mental-poker_amd/libmpshuffle.so
    python tools/r06_group_sweep.py --verify-only --steps 3 --rounds 2 --configs "243712:13,121856:12,487424:13:3072" 2>/dev/null | python -c "
import json, sys
for l in sys.stdin:
    r = json.loads(l); k = r['kernels_ms']
    print('%-6s L=%-5d c=%d  verify %.1f ms  acc %.2f sort %.2f reduce %.2f' % (sys.argv[1], r['group_size'], r['bits'], r['ms_per_step'], k.get('k_bucket_acc', 0), k.get('k_bucket_sort', 0), k.get('k_bucket_reduce', 0)))" $v
  done
done
cp /tmp/lib_keep.so mental-poker_amd/libmpshuffle.so
