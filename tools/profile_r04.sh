#!/bin/bash
# The one profiling pass of round 4 (final sources): every number DESIGN.md section 6 "Round 4" quotes comes from this call.
#   /usr/local/graft/bin/gpurun --timeout 3000 -- 'bash tools/profile_r04.sh'
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1
T=r04
bash tools/profile_round.sh ${T}_default all -- --steps 3 --warmup 1 > gpurun_out/${T}_default.log 2>&1
bash tools/profile_round.sh ${T}_default_steps20 bench -- --steps 20 --warmup 3 >> gpurun_out/${T}_default.log 2>&1
bash tools/profile_round.sh ${T}_d4096 all -- --batch 4096 --steps 10 --warmup 2 --no-extras > gpurun_out/${T}_d4096.log 2>&1
bash tools/profile_round.sh ${T}_s8_128 all -- --m 8 --n 128 --batch 16384 --steps 2 --warmup 1 --no-extras > gpurun_out/${T}_s8_128.log 2>&1
bash tools/all_configs.sh > gpurun_out/${T}_all_configs.txt 2>&1
{
  for P in 0 1; do echo "pipeline depth $P"; python tools/plan_sweep.py --pipeline $P --batches 1,16,64,256,512,1024,2048,4096,8192,16384,32768,65536 --configs default 2>&1 >/dev/null | grep -v "amdgpu\|table build"; done
} > gpurun_out/${T}_plan_sweep.txt
{
  for c in stark bn254 secp256k1; do python tools/decompress_bench.py $c; done; python tools/decompress_bench.py bls12_377 16384
} 2>&1 | grep -v amdgpu > gpurun_out/${T}_decompress.txt
{ python tools/pcie_inclusive.py 262144; python tools/pcie_inclusive.py 16384; } 2>&1 | grep -v amdgpu > gpurun_out/${T}_pcie_inclusive.txt
python tools/host_cost.py 1 64 1024 4096 2>&1 | grep -v amdgpu > gpurun_out/${T}_host_cost.txt
tail -3 gpurun_out/${T}_*.log; cat gpurun_out/${T}_plan_sweep.txt gpurun_out/${T}_decompress.txt gpurun_out/${T}_pcie_inclusive.txt
