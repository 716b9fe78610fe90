#!/bin/bash
# r06 call 13: parity AT the benchmarked batch for the shipped YAML's geometry too (48 x 4 frames x 80 tokens), beside config B's; checkpoint / entry-point tests after the layout fingerprint
cd $GRAFT_REPO_ROOT
OUT=gpurun_out
rm -f $OUT/r06_parity.txt
timeout 1200 python -m pytest -q -s -m gpu -p no:cacheprovider tests/test_parity_fullsize_gpu.py -k "benchmarked_batch" > $OUT/r06_c13_benchmarked_batch.log 2>&1; grep -E "passed|failed|Error" $OUT/r06_c13_benchmarked_batch.log | tail -3; cat $OUT/r06_parity.txt | cut -c1-400
timeout 900 python -m pytest -q -m gpu -p no:cacheprovider tests/test_entrypoint_gpu.py > $OUT/r06_c13_entrypoint_tests.log 2>&1; grep -E "passed|failed" $OUT/r06_c13_entrypoint_tests.log | tail -2
