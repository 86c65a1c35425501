X="--no-extras --no-cpu-baseline --steps 3 --warmup 1"
for cfg in "" "--bucket-bits 11" "--workload chain32" "--workload chain32 --bucket-bits 11" "--workload chain32 --chain-group 12 --chain-slice 21504" "" "--bucket-bits 11"; do
  echo "== $cfg"
  python bench.py $X $cfg 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); ks=d['roofline']['kernels_ms']; n=d['steps']
print('   %.1f k proofs/s, %.1f ms per step; bucket %.1f ms' % (d['value'] / 1e3, d['ms_per_step'], ks.get('k_bucket_msm',0)/n))"
done
