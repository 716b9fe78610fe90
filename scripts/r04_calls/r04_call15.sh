#!/bin/bash
set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_kernels_gpu.py tests/test_parity_fullsize_gpu.py -m gpu -q -k "dropout or own_masks or train_mode or ln_stream or graph" -p no:cacheprovider 2>&1 | tail -5
for P in 0 1 0 1; do
  MPV_DROPOUT_IN_LN=$P MPV_BENCH_BY_SHAPE=$OUT/r04_c15_by_shape_dropln$P.md timeout 300 python bench.py --no-cpu-baseline > $OUT/r04_c15_bench_dropln$P.json 2> $OUT/r04_c15_bench_dropln$P.err
  python - <<PY
import json
r=json.load(open("$OUT/r04_c15_bench_dropln$P.json"))
print("dropout_in_ln=$P ms/step", r["ms_per_step"], "gemm ms", r["roofline"]["gemm_ms_per_step"], "frac", r["roofline"]["frac"], "loss", r["config"]["final_loss"])
PY
done
