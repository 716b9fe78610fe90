#!/bin/bash
# Round-3 closing measurement set on one box: the GPU suite, smoke, the four profile passes (tools/profile_round.sh), the default
# bench line with the in-step GEMM table, and the graph-mode bench.  Every command under its own timeout, stdin closed.
set -u
TAG=${1:-r03_final3}
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
rm -f $OUT/r03_parity.txt
(timeout 900 python -m pytest tests -q -m gpu < /dev/null 2>&1 | tail -15 | cut -c1-300) > $OUT/f_gpu_tests.log 2>&1
grep -E "passed|failed" $OUT/f_gpu_tests.log
(timeout 300 python __graft_entry__.py smoke < /dev/null 2>&1 | tail -2) > $OUT/f_smoke.log; cat $OUT/f_smoke.log
(timeout 1500 bash tools/profile_round.sh $TAG < /dev/null) > $OUT/f_profile.log 2>&1
tail -1 $OUT/f_profile.log | cut -c1-300
(MPV_BENCH_BY_SHAPE=$OUT/${TAG}_gemm_in_step_by_shape.md timeout 400 python bench.py --no-cpu-baseline < /dev/null > $OUT/f_bench_B.json 2> $OUT/f_bench_B.err); tail -4 $OUT/f_bench_B.err | cut -c1-300; cut -c1-400 $OUT/f_bench_B.json
(MPV_GRAPH=1 timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 30 < /dev/null 2>&1 | grep -E "timed region|host |rror" | cut -c1-300) > $OUT/f_bench_graph.log; cat $OUT/f_bench_graph.log
for c in D E; do
  (timeout 400 python bench.py --config $c --no-cpu-baseline --steps 20 < /dev/null > $OUT/f_bench_$c.json 2> $OUT/f_bench_$c.err); tail -2 $OUT/f_bench_$c.err | cut -c1-200; cut -c1-300 $OUT/f_bench_$c.json
done
