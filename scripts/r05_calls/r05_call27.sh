#!/bin/bash
# r05 call 27: the whole GPU suite + smoke on the final tree (after the device-run oracle of the B = 32 test)
cd $GRAFT_REPO_ROOT
export MPV_PARITY_REPORT=$GRAFT_REPO_ROOT/gpurun_out/r05_parity.txt; rm -f $MPV_PARITY_REPORT
timeout 1500 python -m pytest tests -m gpu -q --durations=6 -p no:cacheprovider > gpurun_out/r05_final3_gpu_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r05_final3_gpu_tests.log
grep -v "^E   \|^    \|^$" gpurun_out/r05_final3_gpu_tests.log | tail -12
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
