#!/bin/bash
set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_parity_fullsize_gpu.py -m gpu -q -k "decode or caption or generat or small_m or topk" -p no:cacheprovider 2>&1 | tail -4
timeout 200 python tools/bench_decode.py > $OUT/r04_c9_decode.log 2>&1; cat $OUT/r04_c9_decode.log
cd /tmp; rm -rf /tmp/kd
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kd -o dec -- python $R/tools/bench_decode.py > $OUT/r04_c9_decode_trace.log 2>&1
python $R/tools/rocpd_stats.py $(find /tmp/kd -name "*.db" | head -1) $OUT/r04_c9_decode_kernel_trace.md > /dev/null 2>&1
head -8 $OUT/r04_c9_decode_kernel_trace.md
