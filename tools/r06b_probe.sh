cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1
cp mental-poker_amd/libmpshuffle.so /tmp/lib_keep.so
cp tools/ab/lib_timing.so mental-poker_amd/libmpshuffle.so
python tools/r06_group_sweep.py --verify-only --steps 1 --configs "30464:0,243712:13,121856:12,60928:12:1536" > gpurun_out/r06b_timing.json 2> gpurun_out/r06b_timing.txt
cp /tmp/lib_keep.so mental-poker_amd/libmpshuffle.so
for set in "SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --output-format csv -d gpurun_out/r06b_pmc_$tag -- python tools/r06_group_sweep.py --verify-only --steps 1 --configs "30464:0,243712:13,121856:12" > /dev/null 2> gpurun_out/r06b_pmc_$tag.err
done
python - <<'PY'
import csv, glob, os
for d in sorted(glob.glob("gpurun_out/r06b_pmc_*")):
    if not os.path.isdir(d): continue
    f = sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True))
    if not f: continue
    acc = {}
    for r in csv.DictReader(open(f[0])):
        if "k_bucket_msm" in r["Kernel_Name"]:
            acc.setdefault((r["Kernel_Name"][:40], r["Counter_Name"]), []).append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print(d, k, ["%.4g" % x for x in v])
PY
