#!/bin/bash
# r06 call 7: config D across PROCESSES: does the loss still vary when every torch.empty is zero-filled?  and where do two processes first differ?
cd $GRAFT_REPO_ROOT
OUT=gpurun_out
export MPV_WGRAD_STREAM=0
(for i in 1 2 3; do echo -n "plain        process $i: "; timeout 600 python tools/determinism_bisect.py --config D --steps 1 --dump /tmp/p$i.json 2>&1 | grep "^run 0"; done
 for i in 1 2 3; do echo -n "zero-filled  process $i: "; timeout 600 python tools/determinism_bisect.py --config D --steps 1 --zero-empty --dump /tmp/z$i.json 2>&1 | grep "^run 0"; done
 for i in 1 2; do echo -n "eval, zero-filled process $i: "; timeout 600 python tools/determinism_bisect.py --config D --steps 1 --zero-empty --eval 2>&1 | grep "^run 0"; done
 echo "== zero-filled processes 1 vs 2"; python tools/determinism_bisect.py --compare /tmp/z1.json /tmp/z2.json
 echo "== zero-filled processes 1 vs 3"; python tools/determinism_bisect.py --compare /tmp/z1.json /tmp/z3.json
) 2>&1 | grep -v "Warning\|warn\|amdgpu.ids" | tee $OUT/r06_c7_across_processes.log | cut -c1-220
