#!/bin/bash
# One reproducible profiling pass of a bench.py command on the GPU box; everything lands in gpurun_out/<TAG>/ and the
# summaries the judge reads are then copied to profiles/ by tools/profile_collect.py.
#   usage (repo root, inside one gpurun call):  bash tools/profile_round.sh TAG [MODE] -- [bench.py arguments]
#   MODE: all (default) | trace | pmc | bench
# Passes (separate runs of the SAME command, as MI355X_MICROARCH.md prescribes for the TCC counters):
#   bench        plain run, the JSON line -> bench.json
#   trace        rocprofv3 --kernel-trace --stats
#   pmc_FETCH    rocprofv3 --pmc FETCH_SIZE                      (--steps 1 --warmup 0)
#   pmc_WRITE    rocprofv3 --pmc WRITE_SIZE
#   pmc_TCC      rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum        (L2 hit rate per kernel)
#   pmc_SQ       rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE
TAG=$1; shift
MODE=all
if [ "$1" != "--" ]; then MODE=$1; shift; fi
shift
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
echo "bench.py $*" > "$OUT/command.txt"
git rev-parse HEAD 2>/dev/null >> "$OUT/command.txt"
sha256sum mental-poker_amd/libmpshuffle.so | cut -c1-16 >> "$OUT/command.txt"
if [ "$MODE" = all ] || [ "$MODE" = bench ]; then
  python bench.py "$@" > "$OUT/bench.json" 2> "$OUT/bench.err"
  tail -c 600 "$OUT/bench.json"
fi
if [ "$MODE" = all ] || [ "$MODE" = trace ]; then
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python bench.py --no-cpu-baseline "$@" > "$OUT/trace_bench.json" 2> "$OUT/trace.err"
fi
if [ "$MODE" = all ] || [ "$MODE" = pmc ]; then
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_FETCH" -- python bench.py --no-cpu-baseline --no-extras "$@" --steps 1 --warmup 0 > "$OUT/pmc_FETCH_bench.json" 2> "$OUT/pmc_FETCH.err"
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_WRITE" -- python bench.py --no-cpu-baseline --no-extras "$@" --steps 1 --warmup 0 > "$OUT/pmc_WRITE_bench.json" 2> "$OUT/pmc_WRITE.err"
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_SQ" -- python bench.py --no-cpu-baseline --no-extras "$@" --steps 1 --warmup 0 > "$OUT/pmc_SQ_bench.json" 2> "$OUT/pmc_SQ.err"
  # L2 hit rate (round 5: the bucket kernel's XCD-affine items and contiguous point runs are judged by it)
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/pmc_TCC" -- python bench.py --no-cpu-baseline --no-extras "$@" --steps 1 --warmup 0 > "$OUT/pmc_TCC_bench.json" 2> "$OUT/pmc_TCC.err"
fi
# keep only the CSVs (the raw rocprofv3 output directories also hold agent info etc.)
find "$OUT" -name '*.csv' | head -50
du -sh "$OUT"
