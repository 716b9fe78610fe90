#!/bin/bash
# r05 call 9: (a) where does graph replay first differ from the eager step at config B? (b) the fp32 (split-K partial) tile stored straight from
# the accumulators: wgrad tests + same-box alternation
cd $GRAFT_REPO_ROOT
timeout 600 python tools/graph_vs_eager_configB.py 5 2>&1 | grep -v Warn | tee gpurun_out/r05_c9_graph_vs_eager_configB.log
MPV_GEMM_DIRECT_EPI=2 timeout 900 python -m pytest -q -x -m gpu -p no:cacheprovider tests/test_gemm256_gpu.py tests/test_kernels_gpu.py -k "gemm" > gpurun_out/r05_c9_gemm_tests_direct_f32.log 2>&1
tail -3 gpurun_out/r05_c9_gemm_tests_direct_f32.log
for round in 1 2; do for d in 0 2; do
  MPV_GEMM_DIRECT_EPI=$d MPV_BENCH_BY_SHAPE=gpurun_out/r05_c9_by_shape_direct$d.md timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r05_c9_bench_direct$d.json 2> gpurun_out/r05_c9_bench_direct$d.err
  python -c "
import json;r=json.load(open('gpurun_out/r05_c9_bench_direct$d.json'));ro=r['roofline']
print('direct=$d ms/step', r['ms_per_step'], 'gemm ms', ro['gemm_ms_per_step'], 'loss', r['config']['final_loss'], 'sclk', ro.get('sclk_mhz'), 'W', ro.get('power_w'), ro['by_kernel']['gemm<1,1> wgrad'])"
done; done 2>&1 | tee gpurun_out/r05_c9_direct_f32_ab.log
