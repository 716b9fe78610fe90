#!/bin/bash
# r04 closing set, second edition (after calls 18-25): whole GPU suite, smoke, decode timing + trace, the driver's bench command, profile_round (refreshes
# the PMC traffic figure for the final GEMM sources)
set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
python -m pytest tests -m gpu -q --durations=5 -p no:cacheprovider > $OUT/r04_final_gpu_tests.log 2>&1
echo "pytest rc=$?" >> $OUT/r04_final_gpu_tests.log
grep -v "^E   \|^    \|^$" $OUT/r04_final_gpu_tests.log | tail -12
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/r04_final_smoke.log 2>&1; tail -1 $OUT/r04_final_smoke.log
timeout 200 python tools/bench_decode.py > $OUT/r04_c9_decode_timing.log 2>&1; tail -4 $OUT/r04_c9_decode_timing.log
cd /tmp; rm -rf /tmp/kd
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kd -o dec -- python $R/tools/bench_decode.py > $OUT/r04_c9_decode_trace.log 2>&1
python $R/tools/rocpd_stats.py $(find /tmp/kd -name "*.db" | head -1) $OUT/r04_c9_decode_kernel_trace.md > /dev/null 2>&1
cd $R
bash tools/profile_round.sh r04_final > $OUT/r04_final_profile_round.log 2>&1
cp $OUT/pmc_gemm_latest.json $R/profiles/pmc_gemm_latest.json
MPV_BENCH_BY_SHAPE=$OUT/r04_final_gemm_in_step_by_shape.md timeout 600 python bench.py > $OUT/r04_final_bench_B_1gpu.json 2> $OUT/r04_final_bench_B_1gpu.err
python - <<PY
import json
r=json.load(open("$OUT/r04_final_bench_B_1gpu.json"))
print("final bench: ms/step", r["ms_per_step"], "value", r["value"], "frac", r["roofline"]["frac"], "traffic", r["roofline"]["traffic"], "alg bytes", r["roofline"]["algorithmic_bytes_per_launch"], "step_frac", r["roofline"]["step_frac"], "cpu", r["cpu_baseline"]["value"])
PY
