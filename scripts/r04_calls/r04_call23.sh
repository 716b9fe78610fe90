#!/bin/bash
# decoder-side LayerNorms: stream forward without stores pending at its barriers (LDS-only barriers, gamma / beta requested with the row),
# backward on branch-free buffer accesses (residual-gradient chunks requested together).  Tests, alternation against the previous norm.hip
# (gpurun_ab/libmpv_hip_base.so), serialised traces of both.
set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -q -p no:cacheprovider -k "layernorm or ln_ or norm or stream or train_mode or dropout_in" 2>&1 | tail -3
for L in base new base new; do
  if [ $L = base ]; then export MPV_LIB_PATH=$R/gpurun_ab/libmpv_hip_base.so; else unset MPV_LIB_PATH; fi
  timeout 300 python bench.py --no-cpu-baseline > $OUT/r04_c23_bench_$L.json 2> $OUT/r04_c23_bench_$L.err
  python - <<PY
import json
r=json.load(open("$OUT/r04_c23_bench_$L.json"))
print("$L ms/step", r["ms_per_step"], "gemm ms", r["roofline"]["gemm_ms_per_step"], "loss", r["config"]["final_loss"])
PY
done
cd /tmp && export TMPDIR=/tmp
for L in base new; do
  if [ $L = base ]; then export MPV_LIB_PATH=$R/gpurun_ab/libmpv_hip_base.so; else unset MPV_LIB_PATH; fi
  rm -rf /tmp/kp_$L
  MPV_WGRAD_STREAM=0 timeout 600 rocprofv3 --kernel-trace -d /tmp/kp_$L -o t -- python $R/bench.py --no-cpu-baseline --no-roofline --steps 3 --warmup 2 > /dev/null 2>&1
  python $R/tools/rocpd_stats.py $(find /tmp/kp_$L -name "*.db" | head -1) $OUT/r04_c23_trace_$L.md > /dev/null 2>&1
  echo "== $L"; grep -E "ln_" $OUT/r04_c23_trace_$L.md | cut -c1-150
done
