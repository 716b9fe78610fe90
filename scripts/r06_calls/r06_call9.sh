#!/bin/bash
# r06 call 9: the gradient norm reduced in a fixed order: config D's final loss across three processes, the optimizer test, config B unchanged
cd $GRAFT_REPO_ROOT
OUT=gpurun_out
timeout 600 python -m pytest -q -m gpu -p no:cacheprovider tests/test_kernels_gpu.py -k "adamw" 2>&1 | tail -2
run() { python bench.py --no-cpu-baseline --steps 10 "$@" 2>/dev/null | python -c "import json,sys;r=json.loads(sys.stdin.readline());print(r['ms_per_step'],'final loss',r['config']['final_loss'])"; }
(for i in 1 2 3; do echo -n "config D process $i: "; run --config D; done
 for i in 1 2; do echo -n "config B process $i: "; run; done) 2>&1 | tee $OUT/r06_c9_loss_across_processes.log
