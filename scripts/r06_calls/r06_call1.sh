#!/bin/bash
# r06 call 1: the untouched round-5 tree on this round's first box: the driver's command, then config D and a config-B by-shape table
cd $GRAFT_REPO_ROOT
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_c1_bench_B.json 2> gpurun_out/r06_c1_bench_B.err
MPV_BENCH_BY_SHAPE=gpurun_out/r06_c1_gemm_by_shape.md python bench.py --no-cpu-baseline --steps 20 > gpurun_out/r06_c1_bench_B2.json 2> gpurun_out/r06_c1_bench_B2.err
python bench.py --config D --no-cpu-baseline --steps 20 > gpurun_out/r06_c1_bench_D.json 2> gpurun_out/r06_c1_bench_D.err
tail -3 gpurun_out/r06_c1_bench_B.err; cat gpurun_out/r06_c1_bench_B.json gpurun_out/r06_c1_bench_B2.json gpurun_out/r06_c1_bench_D.json
