#!/bin/bash
# grouped AdamW with two tiles per wave-iteration and prefetched group ids; decode attention with q requested beside the first K batch:
# kernel / model tests, step alternation (MPV_LIB_PATH = previous build), decode timing
set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -q -p no:cacheprovider -k "adamw or optim or decode or kv_cache or beam or caption or engine or graph" > $OUT/r04_c27_tests.log 2>&1; tail -3 $OUT/r04_c27_tests.log | grep -E "passed|failed"
for L in base new base new; do
  if [ $L = base ]; then export MPV_LIB_PATH=$R/gpurun_ab/libmpv_hip_base.so; else unset MPV_LIB_PATH; fi
  timeout 300 python bench.py --no-cpu-baseline > $OUT/r04_c27_bench_$L.json 2> $OUT/r04_c27_bench_$L.err
  python - <<PY
import json
r=json.load(open("$OUT/r04_c27_bench_$L.json"))
print("$L ms/step", r["ms_per_step"], "gemm ms", r["roofline"]["gemm_ms_per_step"], "loss", r["config"]["final_loss"])
PY
  timeout 200 python tools/bench_decode.py 2>/dev/null | grep "^decode" | sed "s/^/$L: /"
done
