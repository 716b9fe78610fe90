#!/bin/bash
# Round 3, GPU call 3: the whole GPU suite + smoke on the final tree, same-box A/B of the composed temporal projection, the default
# bench line with the in-step GEMM table, and the four profile passes (tools/profile_round.sh r03_final).
set -u
R=$PWD
OUT=$R/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
rm -f $OUT/r03_parity.txt
(time timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -15) > $OUT/c3_gpu_tests.log 2>&1
tail -6 $OUT/c3_gpu_tests.log
(timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3) > $OUT/c3_smoke.log
cat $OUT/c3_smoke.log
run() { "$@" python bench.py --no-cpu-baseline --no-roofline --steps 30 2>&1 | grep -E "timed region" | sed 's/.*\] //' | tr '\n' ' '; echo; }
for round in 1 2; do for v in 0 1; do echo -n "MPV_VIT_COMPOSE=$v  "; run env MPV_VIT_COMPOSE=$v; done; done > $OUT/c3_ab_compose.log 2>&1
cat $OUT/c3_ab_compose.log
MPV_BENCH_BY_SHAPE=$OUT/r03_gemm_in_step_by_shape.md python bench.py --steps 50 --warmup 5 > $OUT/c3_bench_B.json 2> $OUT/c3_bench_B.err
tail -1 $OUT/c3_bench_B.json | cut -c1-300; grep -E "\[bench" $OUT/c3_bench_B.err | tail -4
bash tools/profile_round.sh r03_final > $OUT/c3_profile_round.log 2>&1
tail -40 $OUT/c3_profile_round.log | cut -c1-220
