#!/bin/bash
# r05 call 7: (a) graph_step of a data-parallel step as a chain of segments with eager collectives (in-process, once) + the other graph
# tests + the cross-kernel derivative test; (b) same-box alternation tree / LayerNorm-backward grid of round 4; (c) the bench line in
# graph mode with the distributed branch forced (1-rank RCCL): host cost of 16 segment launches + 14 all-reduce calls per step
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest -q -x -m gpu -p no:cacheprovider "tests/test_model_gpu.py::test_graph_step_data_parallel_segments_forced_world1" \
  "tests/test_model_gpu.py::test_graph_step_bit_identical_to_eager" "tests/test_gemm256_gpu.py::test_parked_gelu_derivative_does_not_depend_on_the_tile_kernel" \
  "tests/test_gemm256_gpu.py::test_gemm_gelu_derivative_parked_by_the_forward" "tests/test_entrypoint_gpu.py" > gpurun_out/r05_c7_tests.log 2>&1
tail -8 gpurun_out/r05_c7_tests.log
for gm in 2 8 16; do echo -n "MPV_GEMM_GM=$gm "; MPV_GEMM_GM=$gm python bench.py --no-cpu-baseline --no-roofline --steps 30 2>&1 | grep -E "timed region" | sed "s/.*timed region done: //"; done | tee gpurun_out/r05_c7_gm.log; echo -n "default "; python bench.py --no-cpu-baseline --no-roofline --steps 30 2>&1 | grep -E "timed region" | sed "s/.*timed region done: //"
for mode in eager graph; do
  if [ $mode = graph ]; then export MPV_GRAPH=1; else unset MPV_GRAPH; fi
  MPV_BENCH_FORCE_DIST=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/r05_c7_bench_dist_$mode.json 2> gpurun_out/r05_c7_bench_dist_$mode.err
  grep -E "host enqueue|idle queue|timed region" gpurun_out/r05_c7_bench_dist_$mode.err | sed "s/^/$mode: /"
  python -c "import json;r=json.load(open('gpurun_out/r05_c7_bench_dist_$mode.json'));print('$mode', r['ms_per_step'], r['config']['final_loss'])"
done
