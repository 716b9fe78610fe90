#!/bin/bash
# r04 closing measurement set on the final tree: the driver's command (default bench line incl. the CPU baseline), the by-shape table,
# then tools/profile_round.sh (kernel trace + FETCH / WRITE / MFMA-busy PMC passes, weight-gradient lane off)
set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
MPV_BENCH_BY_SHAPE=$OUT/r04_final_gemm_in_step_by_shape.md timeout 600 python bench.py > $OUT/r04_final_bench_B_1gpu.json 2> $OUT/r04_final_bench_B_1gpu.err
cat $OUT/r04_final_bench_B_1gpu.json | cut -c1-400; grep "host\|timed\|memory" $OUT/r04_final_bench_B_1gpu.err
bash tools/profile_round.sh r04_final > $OUT/r04_final_profile_round.log 2>&1
tail -40 $OUT/r04_final_profile_round.log
