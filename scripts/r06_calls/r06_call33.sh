#!/bin/bash
# r06 call 33: the ITC retrieval step on two ranks (one GPU, gloo on device tensors) against the fp32 oracle on the global batch
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest -q -x -s -m gpu -p no:cacheprovider tests/test_model_gpu.py -k "two_ranks" 2>&1 | grep -v "Warning\|amdgpu.ids\|socket.cpp\|Gloo" | tail -25 | cut -c1-300 | tee gpurun_out/r06_c33_two_rank_itc.log
