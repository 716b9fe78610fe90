#!/bin/bash
# r04 GPU call 2: the persistent tile walk of the 256x256 GEMM (bitwise vs the per-tile launch, race screens), the two re-gated true-dims
# parity cases, and a same-box A/B of the step with the walk off / on, each with its in-step by-shape table.
set -u
R=$PWD
OUT=$R/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gemm256_gpu.py -m gpu -q -x -k "persistent" -p no:cacheprovider > $OUT/r04_c2_gemm_tests.log 2>&1
echo "pytest rc=$?" >> $OUT/r04_c2_gemm_tests.log
tail -30 $OUT/r04_c2_gemm_tests.log
timeout 600 python -m pytest tests/test_parity_fullsize_gpu.py -m gpu -q -k "itm or caption" -p no:cacheprovider > $OUT/r04_c2_parity_tests.log 2>&1
tail -5 $OUT/r04_c2_parity_tests.log
for P in 0 1 0 1; do
  MPV_GEMM_PERSIST=$P MPV_BENCH_BY_SHAPE=$OUT/r04_c2_by_shape_persist$P.md timeout 300 python bench.py --no-cpu-baseline > $OUT/r04_c2_bench_persist$P.json 2> $OUT/r04_c2_bench_persist$P.err
  python - <<PY
import json
r=json.load(open("$OUT/r04_c2_bench_persist$P.json"))
print("persist=$P ms/step", r["ms_per_step"], "gemm ms", r["roofline"]["gemm_ms_per_step"], "frac", r["roofline"]["frac"], "loss", r["config"]["final_loss"])
PY
done
grep "| 768 |" $OUT/r04_c2_by_shape_persist0.md | head -12
echo ---
grep "| 768 |" $OUT/r04_c2_by_shape_persist1.md | head -12
