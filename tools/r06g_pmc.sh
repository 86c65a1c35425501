#!/bin/bash
# PMC counters of the split bucket pipeline's kernels for a few configurations (one pass per counter set, as the guide prescribes)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1
cp mental-poker_amd/libmpshuffle.so /tmp/lib_keep.so
for v in "$@"; do
  cp tools/ab/lib_$v.so mental-poker_amd/libmpshuffle.so
  for set in "SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_LDS" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
    tag=$(echo $set | cut -d' ' -f1)
    rocprofv3 --pmc $set --output-format csv -d gpurun_out/r06g_${v}_$tag -- python tools/r06_group_sweep.py --verify-only --steps 1 --configs "243712:13,121856:12,487424:13:3072" > /dev/null 2> gpurun_out/r06g_${v}_$tag.err
  done
done
cp /tmp/lib_keep.so mental-poker_amd/libmpshuffle.so
python - <<'PY'
import csv, glob, os
for d in sorted(glob.glob("gpurun_out/r06g_*")):
    if not os.path.isdir(d): continue
    f = sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True))
    if not f: continue
    acc = {}
    for r in csv.DictReader(open(f[0])):
        if "k_bucket_" in r["Kernel_Name"] and "recode" not in r["Kernel_Name"] and "fold" not in r["Kernel_Name"]:
            acc.setdefault((r["Kernel_Name"].split("<")[0].split("::")[-1], r["Counter_Name"]), []).append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print(os.path.basename(d), k, ["%.4g" % x for x in v])
PY
