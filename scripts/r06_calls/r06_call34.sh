#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 300 python tools/probe/two_rank_itc_debug.py 2>&1 | grep -v "Warning\|amdgpu.ids\|socket.cpp\|Gloo" | tail -60 | cut -c1-200 | tee gpurun_out/r06_c34_itc_hang.log
