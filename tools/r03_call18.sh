#!/bin/bash
set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
(MPV_ATTN_DUO=3 timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "attention" < /dev/null 2>&1 | grep -E "^E  *Assert|^FAILED|passed|failed|Error" | cut -c1-300 | head -20)
(timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "attention" < /dev/null 2>&1 | grep -E "^E  *Assert|^FAILED|passed|failed|Error" | cut -c1-300 | head -20)
(for i in 1 2; do for m in 2 3; do AB_VIT=1 MPV_ATTN_DUO=$m timeout 120 python tools/attn_pair_ab.py < /dev/null 2>&1 | grep -E "B=256|rror" | sed "s/^/DUO=$m /"; done; done) > $OUT/c18_fused_ab.log
cat $OUT/c18_fused_ab.log
(timeout 300 bash tools/ab_same_box.sh env MPV_ATTN_DUO 2 3 < /dev/null) >> $OUT/c18_fused_ab.log 2>&1
tail -4 $OUT/c18_fused_ab.log
