#!/bin/bash
# Closing validation of a tree whose hot kernels are the ones of the last profile set (scripts/r03_calls/r03_final.sh): the GPU suite, smoke, the
# default bench line exactly as the driver runs it (cpu baseline included), the graph-mode bench and the two side configurations.
# Every command under its own timeout, stdin closed.   Usage (GPU box): bash scripts/r03_calls/r03_close.sh <tag>
set -u
TAG=${1:-r03_final5}
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
rm -f $OUT/r03_parity.txt
(timeout 1200 python -m pytest tests -q -m gpu --durations=8 < /dev/null 2>&1 | grep -vE "Warning|warnings.warn|^$|return func" | tail -22 | cut -c1-200) > $OUT/${TAG}_gpu_tests.log 2>&1
grep -E "passed|failed|error" $OUT/${TAG}_gpu_tests.log | tail -3
(timeout 300 python __graft_entry__.py smoke < /dev/null 2>&1 | tail -2) > $OUT/${TAG}_smoke.log; cat $OUT/${TAG}_smoke.log
(timeout 600 python bench.py < /dev/null > $OUT/${TAG}_bench_B_1gpu.json 2> $OUT/${TAG}_bench_B_1gpu.err); grep "bench +" $OUT/${TAG}_bench_B_1gpu.err | cut -c1-220; cut -c1-400 $OUT/${TAG}_bench_B_1gpu.json
(MPV_GRAPH=1 timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 30 < /dev/null 2>&1 | grep -E "timed region|host |rror" | cut -c1-300) > $OUT/${TAG}_bench_graph.log; cat $OUT/${TAG}_bench_graph.log
for c in D E; do
  (timeout 400 python bench.py --config $c --no-cpu-baseline --steps 20 < /dev/null > $OUT/${TAG}_bench_${c}_1gpu.json 2> $OUT/${TAG}_bench_${c}_1gpu.err); grep -E "timed region|device memory" $OUT/${TAG}_bench_${c}_1gpu.err | cut -c1-220; cut -c1-300 $OUT/${TAG}_bench_${c}_1gpu.json
done
