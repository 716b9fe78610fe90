#!/bin/bash
set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gemm256_gpu.py tests/test_kernels_gpu.py -m gpu -q -k "split or wgrad or epilogue or dropout" -p no:cacheprovider 2>&1 | tail -4
MPV_BENCH_BY_SHAPE=$OUT/r04_c12_by_shape_D.md timeout 300 python bench.py --config D --no-cpu-baseline > $OUT/r04_c12_bench_D.json 2> $OUT/r04_c12_bench_D.err
python - <<PY
import json
r=json.load(open("$OUT/r04_c12_bench_D.json"))
print("D ms/step", r["ms_per_step"], "gemm ms", r["roofline"]["gemm_ms_per_step"], "frac", r["roofline"]["frac"], r["roofline"]["step_frac"], "loss", r["config"]["final_loss"])
PY
head -8 $OUT/r04_c12_by_shape_D.md
timeout 300 python bench.py --no-cpu-baseline --no-roofline | cut -c1-200
