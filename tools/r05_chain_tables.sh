#!/bin/bash
# chain32 (BASELINE config 3): card tables per GPU against links per chain equation (the chain workspace is 68 KB per link in flight)
# usage (on the GPU box): bash tools/r05_chain_tables.sh "<tables> <links per equation> [extra bench flags]" ... > gpurun_out/r05q_chain_tables.txt
[ $# -eq 0 ] && set -- "49152 32" "65536 16" "81920 8" "98304 8"
for cfg in "$@"; do
  set -- $cfg
  T=$1; LK=$2; shift 2
  echo "== tables $T, links per chain equation $LK $*"
  timeout 600 python bench.py --workload chain32 --batch $T --chain-max-links $LK --steps 2 --warmup 1 --no-cpu-baseline "$@" 2>&1 | python -c "
import sys, json
for line in sys.stdin:
    line = line.strip()
    if not line.startswith('{'):
        if 'rror' in line or 'memory' in line: print('   ', line[:300])
        continue
    d = json.loads(line)
    ks = d['roofline']['kernels_ms']; n = d['config']['proofs_per_gpu_per_step'] * d['steps']
    print('   %.1f k proofs/s, %.1f ms per step, %s GB of HBM; us per proof: %s' % (d['value'] / 1e3, d['ms_per_step'], d['config'].get('hbm_per_rank_gb'),
          ', '.join('%s %.3f' % (k, 1e3 * v / n) for k, v in sorted(ks.items(), key=lambda x: -x[1])[:8])))
"
done
