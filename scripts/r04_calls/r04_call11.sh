#!/bin/bash
set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
MPV_BENCH_BY_SHAPE=$OUT/r04_c11_by_shape_D.md timeout 300 python bench.py --config D --no-cpu-baseline > $OUT/r04_c11_bench_D.json 2> $OUT/r04_c11_bench_D.err
python - <<PY
import json
r=json.load(open("$OUT/r04_c11_bench_D.json"))
print("D ms/step", r["ms_per_step"], "gemm ms", r["roofline"]["gemm_ms_per_step"], "frac", r["roofline"]["frac"], r["roofline"]["step_frac"])
PY
head -40 $OUT/r04_c11_by_shape_D.md
