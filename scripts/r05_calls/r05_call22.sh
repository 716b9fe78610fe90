#!/bin/bash
# r05 closing set, second edition (final tree: keep_output, IPC env): whole GPU suite, smoke, profile_round (kernel trace + PMC traffic / MFMA passes; refreshes the traffic figure for the
# final GEMM sources), the driver's bench command with the by-shape table, configs D and E
set -u
R=$GRAFT_REPO_ROOT; cd $R; OUT=$R/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
export MPV_PARITY_REPORT=$OUT/r05_parity.txt; rm -f $MPV_PARITY_REPORT
timeout 1500 python -m pytest tests -m gpu -q --durations=5 -p no:cacheprovider > $OUT/r05_final2_gpu_tests.log 2>&1
echo "pytest rc=$?" >> $OUT/r05_final2_gpu_tests.log
grep -v "^E   \|^    \|^$" $OUT/r05_final2_gpu_tests.log | tail -10
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/r05_final2_smoke.log 2>&1; tail -1 $OUT/r05_final2_smoke.log
bash tools/profile_round.sh r05_final2 > $OUT/r05_final2_profile_round.log 2>&1
cp $OUT/pmc_gemm_latest.json $R/profiles/pmc_gemm_latest.json
cd $R
MPV_BENCH_BY_SHAPE=$OUT/r05_final2_gemm_in_step_by_shape.md timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/r05_final2_bench_B_1gpu.json 2> $OUT/r05_final2_bench_B_1gpu.err
python - <<PY
import json
r=json.load(open("$OUT/r05_final2_bench_B_1gpu.json")); ro=r["roofline"]
print("final bench: ms/step", r["ms_per_step"], "value", r["value"], "frac", ro["frac"], "gemm_frac", ro["gemm_frac"], "gemm ms", ro["gemm_ms_per_step"], "traffic", ro["traffic"], "alg bytes", ro["algorithmic_bytes_per_launch"], "sclk", ro.get("sclk_mhz"), "W", ro.get("power_w"), "cpu", r["cpu_baseline"]["value"])
PY
for c in D E; do
  timeout 600 python bench.py --config $c --steps 10 --warmup 3 --no-cpu-baseline > $OUT/r05_final2_bench_$c.json 2> $OUT/r05_final2_bench_$c.err
  python -c "import json;r=json.load(open('$OUT/r05_final2_bench_$c.json'));ro=r['roofline'];print('config $c: ms/step', r['ms_per_step'], 'value', r['value'], 'frac', ro['frac'], 'gemm_frac', ro['gemm_frac'])"
done
