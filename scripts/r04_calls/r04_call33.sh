#!/bin/bash
# decode attention: K and V rows of the first two chunks requested together, up front: decode / caption tests + timing alternation
set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
python -m pytest tests/test_model_gpu.py tests/test_parity_fullsize_gpu.py tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -k "kv_cache or beam or caption or generate or decode" 2>&1 | grep -E "passed|failed" | tail -2
for L in base new base new; do
  if [ $L = base ]; then export MPV_LIB_PATH=$R/gpurun_ab/libmpv_hip_base.so; else unset MPV_LIB_PATH; fi
  timeout 200 python tools/bench_decode.py 2>/dev/null | grep "^decode" | sed "s/^/$L: /" | tee -a $OUT/r04_c33_decode_attention_ab.log
done
