#!/bin/bash
# r06 call 20: the temporal-embedding gradient on 128 row lanes (was 32): its tests, the model goldens, its time in a trace
cd $GRAFT_REPO_ROOT
OUT=gpurun_out
timeout 900 python -m pytest -q -m gpu -p no:cacheprovider tests/test_kernels_gpu.py -k "im2col_and_assemble" tests/test_model_gpu.py 2>&1 | tail -2
cd /tmp; export TMPDIR=/tmp MPV_WGRAD_STREAM=0
rm -rf /tmp/kt
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $(find /tmp/kt -name "*.db" | head -1) /tmp/kt.md > /dev/null
grep -E "embed_bwd" /tmp/kt.md | tee $OUT/r06_c20_embed_bwd_kernels.log
