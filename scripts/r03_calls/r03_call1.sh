#!/bin/bash
# Round 3, GPU call 1: new glue kernels + persistent attention: kernel tests, model tests, same-box A/B of the persistent attention
# knob, and a kernel trace.  Everything lands in gpurun_out/.
set -u
R=$PWD
OUT=$R/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_gemm256_gpu.py -x -q -m gpu 2>&1 | tail -25) > $OUT/c1_kernel_tests.log
tail -5 $OUT/c1_kernel_tests.log
(timeout 900 python -m pytest tests/test_model_gpu.py tests/test_entrypoint_gpu.py -x -q -m gpu 2>&1 | tail -25) > $OUT/c1_model_tests.log
tail -5 $OUT/c1_model_tests.log
run() { "$@" python bench.py --no-cpu-baseline --no-roofline --steps 30 2>&1 | grep -E "timed region|host enqueue" | sed 's/.*\] //' | tr '\n' ' '; echo; }
for round in 1 2; do for v in 0 1; do echo -n "MPV_ATTN_PERSIST=$v  "; run env MPV_ATTN_PERSIST=$v; done; done > $OUT/c1_ab_attn_persist.log 2>&1
cat $OUT/c1_ab_attn_persist.log
cd /tmp
export MPV_WGRAD_STREAM=0
rm -rf /tmp/kt
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt -o c1 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/c1_trace_bench.log 2>&1
python $R/tools/rocpd_stats.py $(find /tmp/kt -name "*.db" | head -1) $OUT/c1_kernel_trace.md > /dev/null
head -45 $OUT/c1_kernel_trace.md | cut -c1-200
tail -2 $OUT/c1_kernel_trace.md
