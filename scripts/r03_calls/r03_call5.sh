#!/bin/bash
set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
rm -f $OUT/r03_parity.txt
(time timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -12) > $OUT/c5_gpu_tests.log 2>&1
tail -8 $OUT/c5_gpu_tests.log | cut -c1-400
(timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2) > $OUT/c5_smoke.log; cat $OUT/c5_smoke.log
python bench.py --steps 50 --warmup 5 --no-cpu-baseline > $OUT/c5_bench_B.json 2> $OUT/c5_bench_B.err
tail -1 $OUT/c5_bench_B.json | cut -c1-330
cd /tmp; export MPV_WGRAD_STREAM=0; rm -rf /tmp/kt
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt -o c5 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/c5_trace_bench.log 2>&1
python $R/tools/rocpd_stats.py $(find /tmp/kt -name "*.db" | head -1) $OUT/c5_kernel_trace.md > /dev/null
grep -E "temporal_attn|attn_fwd_pres|compose_finish|ln_stream" $OUT/c5_kernel_trace.md | cut -c1-150
