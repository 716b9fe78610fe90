#!/bin/bash
# attention changes of call 24: the GPU tests that did not run there (wrong file name), and a second serialised trace (outlier check)
set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_parity_fullsize_gpu.py tests/test_model_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -4
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kp_new
MPV_WGRAD_STREAM=0 timeout 600 rocprofv3 --kernel-trace -d /tmp/kp_new -o t -- python $R/bench.py --no-cpu-baseline --no-roofline --steps 5 --warmup 2 > /dev/null 2>&1
python $R/tools/rocpd_stats.py $(find /tmp/kp_new -name "*.db" | head -1) $OUT/r04_c25_trace_new.md > /dev/null 2>&1
grep -E "attn_|kernel \|" $OUT/r04_c25_trace_new.md | cut -c1-150
