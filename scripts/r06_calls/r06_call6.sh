#!/bin/bash
# r06 call 6: is config D's run-to-run loss difference an uninitialised read?  Fresh allocations poisoned with NaN bit patterns.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out
(echo "== config D, poisoned allocator, train"; MPV_WGRAD_STREAM=0 timeout 900 python tools/determinism_bisect.py --config D --steps 1 --poison 100
 echo "== config B, poisoned allocator, train"; MPV_WGRAD_STREAM=0 timeout 900 python tools/determinism_bisect.py --config B --steps 1 --poison 100
) 2>&1 | grep -v "Warning\|warn\|amdgpu.ids" | tee $OUT/r06_c6_poisoned_allocator.log | cut -c1-200 | head -150
