#!/bin/bash
# The one profiling pass of round 5 (final sources): every number DESIGN.md section 6 "Round 5" quotes comes from this call.
#   /usr/local/graft/bin/gpurun --timeout 3000 -- 'bash tools/profile_r05.sh'
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1
T=r05
bash tools/profile_round.sh ${T}_default all -- --steps 3 --warmup 1 > gpurun_out/${T}_default.log 2>&1
bash tools/profile_round.sh ${T}_default_steps20 bench -- --steps 20 --warmup 3 >> gpurun_out/${T}_default.log 2>&1
bash tools/profile_round.sh ${T}_s8_128 all -- --m 8 --n 128 --batch 16384 --steps 2 --warmup 1 --no-extras > gpurun_out/${T}_s8_128.log 2>&1
# BASELINE config 3 (chain32: 65 536 tables x 32 links, 8 tables per chain equation, passes of ~22 000 tables)
bash tools/profile_round.sh ${T}_chain32 all -- --workload chain32 --steps 1 --warmup 1 --no-extras > gpurun_out/${T}_chain32.log 2>&1
# the default step with the verify calls on the second lane (mp_set_pipeline): what filling the tails of the large launches is worth
python bench.py --pipeline 1 --no-extras --no-cpu-baseline > gpurun_out/${T}_default_pipelined_bench.json 2> /dev/null
# the bucket kernel's L2 hit rate without the XCD-affine items / without the contiguous point runs (same box, same pass)
for v in "MP_BK_XCD=0" "MP_BK_TILE=0 MP_BK_XCD=0"; do
  tag=$(echo "$v" | tr ' =' '__')
  env $v rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/${T}_tcc_$tag -- python bench.py --no-cpu-baseline --no-extras --steps 1 --warmup 0 > /dev/null 2> gpurun_out/${T}_tcc_$tag.err
  python - "$v" gpurun_out/${T}_tcc_$tag <<'PY'
import csv, glob, os, sys
f = sorted(glob.glob(sys.argv[2] + "/**/*counter_collection.csv", recursive=True), key=os.path.getmtime, reverse=True)
acc = {}
for r in csv.DictReader(open(f[0])) if f else []:
    if "k_bucket_msm" in r["Kernel_Name"]:
        acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
if acc:
    h, m = acc["TCC_HIT_sum"][-1], acc["TCC_MISS_sum"][-1]      # (the last launch: the timed step's, not the priming pass's)
    print("k_bucket_msm %-28s TCC_HIT %.4g TCC_MISS %.4g  L2 hit rate %.3f" % (sys.argv[1], h, m, h / (h + m)))
PY
done > gpurun_out/${T}_bucket_l2.txt 2>&1
MP_CONFIGS_OUT=gpurun_out/${T}f bash tools/all_configs.sh > gpurun_out/${T}_all_configs.txt 2>&1
{ python tools/pcie_inclusive.py 262144; python tools/pcie_inclusive.py 16384; } 2>&1 | grep -v amdgpu > gpurun_out/${T}_pcie_inclusive.txt
python tools/r05_small.py 1 64 1024 4096 2>&1 | grep -v amdgpu > gpurun_out/${T}_small_batches.txt
tail -3 gpurun_out/${T}_*.log; cat gpurun_out/${T}_bucket_l2.txt gpurun_out/${T}_all_configs.txt gpurun_out/${T}_pcie_inclusive.txt
