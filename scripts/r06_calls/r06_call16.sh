#!/bin/bash
# r06 call 16: few-tile products (top decoder layer on the loss window, abstractor): 256x256 kernel vs the 128x128 kernel with its tail split
cd $GRAFT_REPO_ROOT
python tools/small_tile_ab.py 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/r06_c16_small_tile_ab.log
