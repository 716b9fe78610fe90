#!/bin/bash
# r06 call 25: two ranks on one GPU: eager (autograd) vs forward_backward vs graph_step's first call -- which reduced gradient stage differs?
cd $GRAFT_REPO_ROOT
timeout 600 python tools/probe/two_rank_debug.py 2>&1 | grep -v "Warning\|amdgpu.ids\|socket.cpp\|Gloo" | tee gpurun_out/r06_c25_two_rank_debug.log | cut -c1-400
