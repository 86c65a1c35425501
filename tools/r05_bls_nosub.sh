#!/bin/bash
# BLS12-377 (row f3), 300 cards (10,30): with and without the per-point subgroup test, groups on / off, proofs in flight
X="--no-extras --no-cpu-baseline --steps 2 --warmup 1 --curve bls12_377 --m 10 --n 30"
for cfg in "--batch 4096" "--batch 4096 --no-subgroup-check" "--batch 4096 --no-subgroup-check --group-points 30464" "--batch 8192 --no-subgroup-check" "--batch 8192 --no-subgroup-check --group-points 30464" "--batch 16384 --no-subgroup-check --group-points 30464" "--batch 16384"; do
  echo "== $cfg"
  timeout 900 python bench.py $X $cfg 2>&1 | python -c "
import sys, json
for line in sys.stdin:
    line = line.strip()
    if not line.startswith('{'):
        if 'rror' in line or 'memory' in line: print('   ', line[:300])
        continue
    d = json.loads(line)
    ks = d['roofline']['kernels_ms']; n = d['steps']
    print('   %.2f k pairs/s, %.1f ms per step, %s GB of HBM; ms per step: %s' % (d['value'] / 1e3, d['ms_per_step'], d['config'].get('hbm_per_rank_gb'),
          ', '.join('%s %.1f' % (k, v / n) for k, v in sorted(ks.items(), key=lambda x: -x[1])[:8])))
"
done
