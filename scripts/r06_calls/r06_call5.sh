#!/bin/bash
# r06 call 5: config D's final loss differs from run to run on one tree (config B's does not): bisect to the first entry point whose output bits differ
cd $GRAFT_REPO_ROOT
OUT=gpurun_out
(echo "== lane off, train"; MPV_WGRAD_STREAM=0 timeout 600 python tools/determinism_bisect.py --config D --steps 3
 echo "== lane off, eval (no dropout)"; MPV_WGRAD_STREAM=0 timeout 600 python tools/determinism_bisect.py --config D --eval --steps 3
 echo "== lane off, train, config B"; MPV_WGRAD_STREAM=0 timeout 600 python tools/determinism_bisect.py --config B --steps 2
 echo "== lane ON, train, config D (losses only are meaningful)"; timeout 600 python tools/determinism_bisect.py --config D --steps 3 | grep -E "^run [0-9]: [0-9]+ calls, loss"
) 2>&1 | grep -v "Warning\|warn" | tee $OUT/r06_c5_determinism_bisect.log
