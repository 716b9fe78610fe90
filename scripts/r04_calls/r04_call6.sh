#!/bin/bash
set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_parity_fullsize_gpu.py -m gpu -q -k "own_masks" -p no:cacheprovider > $OUT/r04_c6_tests.log 2>&1
echo "pytest rc=$?" >> $OUT/r04_c6_tests.log
grep -v "^    \|^$" $OUT/r04_c6_tests.log | tail -30; cat $OUT/r04_parity.txt
