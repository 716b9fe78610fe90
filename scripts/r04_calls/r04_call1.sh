#!/bin/bash
# r04 GPU call 1: the whole GPU suite on the new tree (bf16-row temporal attention, true-dims goldens of the 8(f) rows, graph-DP test),
# the aten-op tracer at config B, a baseline bench line with the in-step by-shape table, the graph-replay line (host CPU per step with
# the bounded queue), decode-regime evidence (timing + kernel trace) and a kernel trace of the step.
set -u
R=$PWD
OUT=$R/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
python -m pytest tests -m gpu -q --durations=12 -p no:cacheprovider > $OUT/r04_c1_gpu_tests.log 2>&1
echo "pytest rc=$?" >> $OUT/r04_c1_gpu_tests.log
tail -25 $OUT/r04_c1_gpu_tests.log
timeout 300 python tools/trace_framework_ops.py --gpu > $OUT/r04_c1_framework_ops.log 2>&1
tail -40 $OUT/r04_c1_framework_ops.log
MPV_BENCH_BY_SHAPE=$OUT/r04_c1_gemm_in_step_by_shape.md timeout 400 python bench.py > $OUT/r04_c1_bench_B.json 2> $OUT/r04_c1_bench_B.err
cat $OUT/r04_c1_bench_B.json; grep "host\|timed\|memory" $OUT/r04_c1_bench_B.err
MPV_GRAPH=1 timeout 300 python bench.py --no-cpu-baseline --no-roofline > $OUT/r04_c1_bench_graph.json 2> $OUT/r04_c1_bench_graph.err
cat $OUT/r04_c1_bench_graph.json; grep "host\|timed" $OUT/r04_c1_bench_graph.err
MPV_TEMPORAL_ROWS=fp32 timeout 300 python bench.py --no-cpu-baseline --no-roofline > $OUT/r04_c1_bench_temporal_fp32rows.json 2> $OUT/r04_c1_bench_temporal_fp32rows.err
cat $OUT/r04_c1_bench_temporal_fp32rows.json
timeout 300 python tools/bench_decode.py > $OUT/r04_c1_decode.log 2>&1
cat $OUT/r04_c1_decode.log
cd /tmp
rm -rf /tmp/kd /tmp/kt
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kd -o dec -- python $R/tools/bench_decode.py > $OUT/r04_c1_decode_trace.log 2>&1
python $R/tools/rocpd_stats.py $(find /tmp/kd -name "*.db" | head -1) $OUT/r04_c1_decode_kernel_trace.md > /dev/null 2>&1
head -20 $OUT/r04_c1_decode_kernel_trace.md
MPV_WGRAD_STREAM=0 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt -o st -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/r04_c1_trace_bench.log 2>&1
python $R/tools/rocpd_stats.py $(find /tmp/kt -name "*.db" | head -1) $OUT/r04_c1_kernel_trace.md > /dev/null 2>&1
head -45 $OUT/r04_c1_kernel_trace.md
