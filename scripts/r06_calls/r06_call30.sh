#!/bin/bash
# r06 call 30: where the two-rank-on-one-GPU bench run spends its time
cd $GRAFT_REPO_ROOT
python -c "import torch" 2>/dev/null
for r in 0 1; do
  RANK=$r LOCAL_RANK=$r WORLD_SIZE=2 MASTER_ADDR=127.0.0.1 MASTER_PORT=29411 python bench.py --gpus 2 --steps 2 --warmup 1 --batch 4 --no-cpu-baseline --_test-one-gpu > gpurun_out/r06_c30_rank$r.out 2> gpurun_out/r06_c30_rank$r.err &
done
wait
grep "^\[bench" gpurun_out/r06_c30_rank0.err | cut -c1-200
