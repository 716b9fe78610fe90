#!/bin/bash
# r06 call 24: two ranks on the box's one GPU (gloo on device tensors): eager vs segmented replay, self-check, replicas identical across ranks
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest -q -x -m gpu -p no:cacheprovider tests/test_model_gpu.py -k "two_ranks" 2>&1 | grep -v "Warning\|amdgpu.ids" | tail -30 | tee gpurun_out/r06_c24_two_ranks.log
