#!/bin/bash
# Which of a table's four static work splits should a batch of B proofs take?  Runs bench.py over batch sizes x mp_set_latency_batch
# settings inside ONE gpurun call (boxes differ) and prints proofs/s per combination; the crossovers are the defaults of
# engine_core.hpp (latency_batch / medium_batch / tiny_batch).  Output of the round-3 run: profiles/r03k_plan_sweep.txt.
#   usage (repo root):  gpurun -- 'bash tools/plan_sweep.sh [bench.py arguments, e.g. --m 8 --n 128]'
O=gpurun_out/plan_sweep; mkdir -p $O
for B in 256 1024 1536 2048 3072 4096 8192 16384 24576 32768 49152 65536; do
for lb in 512 1024 2048 4096 8192 16384 0; do
python bench.py --no-extras --no-cpu-baseline --batch $B --steps 8 --warmup 2 --latency-batch $lb "$@" > $O/b${B}_lb$lb.json 2> $O/b${B}_lb$lb.err
python - <<PY
import json
try:
    d = json.load(open('$O/b${B}_lb$lb.json'))
    print('B=%6d latency_batch=%6d  %9.0f proofs/s  %7.2f ms/step' % ($B, $lb, d['value'], d['ms_per_step']))
except Exception as e:
    print('B=$B latency_batch=$lb FAILED', e)
PY
done
done
