#!/bin/bash
# attention: every request of a block in one round trip (branch-free row fragments; Q, dO, O and the statistic of a dQ block together;
# statistics / K rows of a dK/dV item before its image DMAs; delta stored with the dQ rows).  Attention tests + full-size parity, then
# alternation against the previous attention sources (gpurun_ab/libmpv_hip_base.so) and serialised traces of both.
set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_attention_gpu.py tests/test_parity_fullsize_gpu.py tests/test_model_gpu.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -4
for L in base new base new; do
  if [ $L = base ]; then export MPV_LIB_PATH=$R/gpurun_ab/libmpv_hip_base.so; else unset MPV_LIB_PATH; fi
  timeout 300 python bench.py --no-cpu-baseline > $OUT/r04_c24_bench_$L.json 2> $OUT/r04_c24_bench_$L.err
  python - <<PY
import json
r=json.load(open("$OUT/r04_c24_bench_$L.json"))
print("$L ms/step", r["ms_per_step"], "gemm ms", r["roofline"]["gemm_ms_per_step"], "loss", r["config"]["final_loss"])
PY
done
cd /tmp && export TMPDIR=/tmp
for L in base new; do
  if [ $L = base ]; then export MPV_LIB_PATH=$R/gpurun_ab/libmpv_hip_base.so; else unset MPV_LIB_PATH; fi
  rm -rf /tmp/kp_$L
  MPV_WGRAD_STREAM=0 timeout 600 rocprofv3 --kernel-trace -d /tmp/kp_$L -o t -- python $R/bench.py --no-cpu-baseline --no-roofline --steps 3 --warmup 2 > /dev/null 2>&1
  python $R/tools/rocpd_stats.py $(find /tmp/kp_$L -name "*.db" | head -1) $OUT/r04_c24_trace_$L.md > /dev/null 2>&1
  echo "== $L"; grep -E "attn_" $OUT/r04_c24_trace_$L.md | cut -c1-150
done
