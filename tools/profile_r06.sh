#!/bin/bash
# The profiling pass of round 6 (final sources): every number DESIGN.md's front section quotes comes from this call.
#   /usr/local/graft/bin/gpurun --timeout 3000 -- 'bash tools/profile_r06.sh'
# Headline: the kernel trace and the PMC passes run `--no-extras --steps 3 --warmup 1`, so that r06_default_kernel_stats.csv's AverageNs IS
# the per-launch time of each hot kernel (round 5's trace averaged 689 calls of every batch size: VERDICT r05 item 8).
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1
T=r06
bash tools/profile_round.sh ${T}_default bench -- > gpurun_out/${T}_default.log 2>&1
bash tools/profile_round.sh ${T}_default trace -- --no-extras --steps 3 --warmup 1 >> gpurun_out/${T}_default.log 2>&1
bash tools/profile_round.sh ${T}_default pmc -- --no-extras --steps 3 --warmup 1 >> gpurun_out/${T}_default.log 2>&1
bash tools/profile_round.sh ${T}_default_steps20 bench -- --steps 20 --warmup 3 --no-extras >> gpurun_out/${T}_default.log 2>&1
bash tools/profile_round.sh ${T}_s8_128 all -- --m 8 --n 128 --batch 16384 --steps 2 --warmup 1 --no-extras > gpurun_out/${T}_s8_128.log 2>&1
bash tools/profile_round.sh ${T}_chain32 all -- --workload chain32 --steps 1 --warmup 1 --no-extras > gpurun_out/${T}_chain32.log 2>&1
python bench.py --pipeline 1 --no-extras --no-cpu-baseline > gpurun_out/${T}_default_pipelined_bench.json 2> /dev/null
MP_CONFIGS_OUT=gpurun_out/${T}f bash tools/all_configs.sh > gpurun_out/${T}_all_configs.txt 2>&1
python bench.py --no-extras --no-cpu-baseline --steps 2 --warmup 1 --curve bls12_377 --m 10 --n 30 --batch 4096 --validated-once > gpurun_out/${T}f/bls_10_30_validated_once.json 2>> gpurun_out/${T}f/err.txt
{ python tools/pcie_inclusive.py 262144; python tools/pcie_inclusive.py 16384; } 2>&1 | grep -v amdgpu > gpurun_out/${T}_pcie_inclusive.txt
python tools/r05_small.py 1 64 1024 4096 2>&1 | grep -v amdgpu > gpurun_out/${T}_small_batches.txt
python tools/r06k_one_bad.py 2>&1 | grep -v amdgpu > gpurun_out/${T}_one_bad_strategies.txt
for f in gpurun_out/${T}_*.log; do tail -n 3 "$f"; done; cat gpurun_out/${T}_all_configs.txt gpurun_out/${T}_pcie_inclusive.txt | head -40
