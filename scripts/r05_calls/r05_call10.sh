#!/bin/bash
# r05 call 10: the whole GPU suite on the current tree (timing per test), smoke()
cd $GRAFT_REPO_ROOT
export MPV_PARITY_REPORT=$GRAFT_REPO_ROOT/gpurun_out/r05_parity.txt; rm -f $MPV_PARITY_REPORT
timeout 2400 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider --durations=15 > gpurun_out/r05_c10_gpu_tests.log 2>&1
tail -30 gpurun_out/r05_c10_gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee gpurun_out/r05_c10_smoke.log
