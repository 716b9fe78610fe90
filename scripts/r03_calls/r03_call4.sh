#!/bin/bash
set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
(timeout 600 python -m pytest tests/test_model_gpu.py -q -m gpu -k "generation_cls" 2>&1 | grep -E "^E |assert|passed|failed" | cut -c1-1500 | head -40) > $OUT/c4_itm.log
cat $OUT/c4_itm.log
(MPV_VIT_COMPOSE=0 timeout 600 python -m pytest tests/test_model_gpu.py -q -m gpu -k "generation_cls" 2>&1 | grep -E "^E |passed|failed" | cut -c1-600 | head -10) > $OUT/c4_itm_nocompose.log
cat $OUT/c4_itm_nocompose.log
rm -f $OUT/r03_parity.txt
(timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -12) > $OUT/c4_gpu_tests.log 2>&1
tail -8 $OUT/c4_gpu_tests.log | cut -c1-300
