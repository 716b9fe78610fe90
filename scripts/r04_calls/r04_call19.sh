#!/bin/bash
# decode regime: host-side search state + shared-prefix beam re-order (tests, timing), and weights streamed non-temporally (variant library)
set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_parity_fullsize_gpu.py -m gpu -q -p no:cacheprovider -k "kv_cache or beam_reorder or caption or generate" 2>&1 | tail -4
for L in base nt base nt; do
  if [ $L = nt ]; then export MPV_LIB_PATH=$R/gpurun_ab/libmpv_hip_nt.so; else unset MPV_LIB_PATH; fi
  timeout 300 python tools/bench_decode.py 2>/dev/null | grep -E "^decode|^prefill" | sed "s/^/$L: /" | tee -a $OUT/r04_c19_decode_nt_ab.log
done
