#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -6
X="--no-extras --no-cpu-baseline --steps 2 --warmup 1"
for v in "" "--validated-once" "--no-subgroup-check"; do
  python bench.py $X --curve bls12_377 --m 10 --n 30 --batch 4096 $v 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels_ms']; s=d['steps']
print('bls (10,30) B=4096 %-22s %8.0f pairs/s %7.1f ms  subgroup %.1f var %.1f acc %.1f validated_once=%s' % (sys.argv[1] if len(sys.argv)>1 else 'default', d['value'], d['ms_per_step'], k.get('k_subgroup_check',0)/s, k.get('k_var_msm',0)/s, k.get('k_bucket_acc',0)/s, d['config'].get('validated_once')))" "$v"
done
python bench.py > gpurun_out/r06s_bench.json 2> gpurun_out/r06s_bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r06s_bench.json").read().strip().splitlines()[-1]); c=d["config"]
print("headline", round(d["value"]), round(d["ms_per_step"],2), c["verification"][:90])
for k in ("one_bad_value","pct1_bad_value","pct1_bad_first_value","per_rank_sclk_mhz","per_rank_power_w","per_rank_host_enqueue_ms_per_step","batch_1024_serial","batch_1024_pipelined","api_host_pinned_value"):
    print(" ", k, c.get(k))
PY
