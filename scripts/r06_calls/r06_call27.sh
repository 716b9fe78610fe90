#!/bin/bash
# r06 call 27: bench.py as two ranks on the box's one GPU (gloo on device tensors): the whole N > 1 flow on real kernels
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest -q -x -m gpu -p no:cacheprovider tests/test_entrypoint_gpu.py -k "two_ranks" 2>&1 | grep -v "Warning\|amdgpu.ids" | tail -25 | cut -c1-400 | tee gpurun_out/r06_c27_bench_two_ranks.log
