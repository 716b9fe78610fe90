#!/bin/bash
# the round's last tree: whole GPU suite + smoke once more (insurance run before the driver's own)
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/r04_final3_gpu_tests.log 2>&1; echo "pytest rc=$?" >> $OUT/r04_final3_gpu_tests.log
grep -E "passed|failed|pytest rc" $OUT/r04_final3_gpu_tests.log | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
