#!/bin/bash
# r04 GPU call 3: persistent walk v2 (last-tile counted waits fixed, counted end-of-epilogue wait) -- bitwise tests, then the step A/B
set -u
R=$PWD
OUT=$R/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gemm256_gpu.py -m gpu -q -k "persistent" -p no:cacheprovider > $OUT/r04_c3_gemm_tests.log 2>&1
echo "pytest rc=$?" >> $OUT/r04_c3_gemm_tests.log
grep -v "^E   \|^    \|^$" $OUT/r04_c3_gemm_tests.log | tail -15
for P in 0 1 0 1; do
  MPV_GEMM_PERSIST=$P MPV_BENCH_BY_SHAPE=$OUT/r04_c3_by_shape_persist$P.md timeout 300 python bench.py --no-cpu-baseline > $OUT/r04_c3_bench_persist$P.json 2> $OUT/r04_c3_bench_persist$P.err
  python - <<PY
import json
r=json.load(open("$OUT/r04_c3_bench_persist$P.json"))
print("persist=$P ms/step", r["ms_per_step"], "gemm ms", r["roofline"]["gemm_ms_per_step"], "frac", r["roofline"]["frac"], "loss", r["config"]["final_loss"])
PY
done
grep "| 768 |" $OUT/r04_c3_by_shape_persist0.md | grep "gemm<0,0>" | head -8
echo ---
grep "| 768 |" $OUT/r04_c3_by_shape_persist1.md | grep "gemm<0,0>" | head -8
timeout 120 python tools/bench_decode.py > $OUT/r04_c3_decode.log 2>&1; cat $OUT/r04_c3_decode.log
timeout 300 python -m pytest tests/test_model_gpu.py tests/test_kernels_gpu.py -m gpu -q -k "decode or caption or generat" -p no:cacheprovider 2>&1 | tail -3
