#!/bin/bash
# r06 call 37: run_pretrain_distributed_gpt3.py as two ranks on the box's one GPU (gloo swapped in by the test worker): ZeRO-1 default, and MPV_GRAPH=1
cd $GRAFT_REPO_ROOT
timeout 500 python -m pytest -q -x -m gpu -p no:cacheprovider tests/test_entrypoint_gpu.py -k "entrypoint_two_ranks" 2>&1 | grep -v "Warning\|amdgpu.ids\|socket.cpp\|Gloo" | tail -30 | cut -c1-300 | tee gpurun_out/r06_c37_entrypoint_two_ranks.log
