#!/bin/bash
set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "attention" 2>&1 | grep -E "^E |assert|passed|failed|Error" | cut -c1-600 | head -40) > $OUT/c6_attn_tests.log
cat $OUT/c6_attn_tests.log
(for m in 0 1 2; do MPV_ATTN_PAIR=$m timeout 120 python tools/attn_pair_ab.py; done; for m in 0 2; do AB_S=208 MPV_ATTN_PAIR=$m timeout 120 python tools/attn_pair_ab.py; done) 2>&1 | grep -E "MPV_ATTN_PAIR|Error|error" > $OUT/c6_attn_pair_ab.log
cat $OUT/c6_attn_pair_ab.log
(bash tools/ab_same_box.sh env MPV_ATTN_PAIR 0 2) > $OUT/c6_ab_attn_pair_step.log 2>&1
cat $OUT/c6_ab_attn_pair_step.log
(bash tools/ab_same_box.sh env MPV_VIT_COMPOSE_LANE 0 1) > $OUT/c6_ab_compose_lane.log 2>&1
cat $OUT/c6_ab_compose_lane.log
