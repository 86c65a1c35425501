set +x
O=${MP_CONFIGS_OUT:-gpurun_out/r05f}; mkdir -p $O
X="--no-extras --no-cpu-baseline --steps 2 --warmup 1"
python bench.py $X --curve secp256k1 > $O/secp256k1.json 2>$O/err.txt
python bench.py $X --curve bn254 > $O/bn254.json 2>>$O/err.txt
python bench.py $X --workload mixed > $O/mixed.json 2>>$O/err.txt
python bench.py $X --workload chain32 > $O/chain32.json 2>>$O/err.txt
python bench.py $X --m 4 --n 13 > $O/s4_13.json 2>>$O/err.txt
python bench.py $X --m 8 --n 128 --batch 16384 > $O/s8_128.json 2>>$O/err.txt
python bench.py $X --m 8 --n 128 --batch 32768 > $O/s8_128_B32768.json 2>>$O/err.txt
python bench.py $X --m 16 --n 64 --batch 16384 > $O/s16_64.json 2>>$O/err.txt
python bench.py $X --m 32 --n 32 --batch 8192 > $O/s32_32.json 2>>$O/err.txt
python bench.py $X --curve bls12_377 --m 10 --n 30 --batch 4096 > $O/bls_10_30.json 2>>$O/err.txt
python bench.py $X --curve bls12_377 --m 10 --n 30 --batch 4096 --pipeline 1 > $O/bls_10_30_p1.json 2>>$O/err.txt
# (the reference's like-for-like: its verify_shuffle takes typed points, validated when they were deserialised -- no subgroup test in the call)
python bench.py $X --curve bls12_377 --m 10 --n 30 --batch 4096 --no-subgroup-check > $O/bls_10_30_nosub.json 2>>$O/err.txt
python bench.py $X --curve bls12_377 --m 10 --n 30 --batch 4096 --no-subgroup-check --group-points 30464 > $O/bls_10_30_nosub_groups.json 2>>$O/err.txt
python tools/latency.py > $O/latency.txt 2>>$O/err.txt
python examples/parameter_selection.py > $O/parameter_selection.txt 2>>$O/err.txt
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/*.json")):
    try:
        d=json.load(open(f)); print(f.split("/")[-1], round(d["value"]), round(d["ms_per_step"],1))
    except Exception as e: print(f, "ERR", e)
PY
tail -5 $O/latency.txt; tail -12 $O/parameter_selection.txt; tail -3 $O/err.txt
