#!/bin/bash
# r04 GPU call 4: GELU' parked by the forward epilogue (preact_deriv / MPV_ACT_DERIV): kernel + model tests, same-box step A/B
set -u
R=$PWD
OUT=$R/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gemm256_gpu.py tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -q -k "derivative or epilogue or golden or properties or loss_window or trainable" -p no:cacheprovider > $OUT/r04_c4_tests.log 2>&1
echo "pytest rc=$?" >> $OUT/r04_c4_tests.log
grep -v "^E   \|^    \|^$" $OUT/r04_c4_tests.log | tail -12
for P in 0 1 0 1; do
  MPV_GELU_DERIV=$P MPV_BENCH_BY_SHAPE=$OUT/r04_c4_by_shape_deriv$P.md timeout 300 python bench.py --no-cpu-baseline > $OUT/r04_c4_bench_deriv$P.json 2> $OUT/r04_c4_bench_deriv$P.err
  python - <<PY
import json
r=json.load(open("$OUT/r04_c4_bench_deriv$P.json"))
print("deriv=$P ms/step", r["ms_per_step"], "gemm ms", r["roofline"]["gemm_ms_per_step"], "frac", r["roofline"]["frac"], "loss", r["config"]["final_loss"])
PY
done
grep "act_bwd_z\|preact_out" $OUT/r04_c4_by_shape_deriv0.md | head -8
echo ---
grep "act_bwd_z\|preact_out" $OUT/r04_c4_by_shape_deriv1.md | head -8
