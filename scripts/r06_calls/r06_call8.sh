#!/bin/bash
# r06 call 8: two PROCESSES from the seeded initial weights (no warm-up, torch.empty zero-filled): the first call whose output bits differ
cd $GRAFT_REPO_ROOT
OUT=gpurun_out
export MPV_WGRAD_STREAM=0
(for i in 1 2 3; do echo -n "process $i: "; timeout 600 python tools/determinism_bisect.py --config D --steps 1 --warmup 0 --zero-empty --dump /tmp/w$i.json 2>&1 | grep "^run 0"; done
 echo "== 1 vs 2"; python tools/determinism_bisect.py --compare /tmp/w1.json /tmp/w2.json
 echo "== 1 vs 3"; python tools/determinism_bisect.py --compare /tmp/w1.json /tmp/w3.json
) 2>&1 | grep -v "Warning\|warn\|amdgpu.ids" | tee $OUT/r06_c8_first_step_across_processes.log | cut -c1-260
