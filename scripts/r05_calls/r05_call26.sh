#!/bin/bash
# r05 call 26: the B = 32 parity test with its fp32 restatement run on the device (slice 0 also on the host: the two placements must agree)
cd $GRAFT_REPO_ROOT
export MPV_PARITY_REPORT=$GRAFT_REPO_ROOT/gpurun_out/r05_c26_parity_b32.txt; rm -f $MPV_PARITY_REPORT
timeout 900 python -m pytest -q -x -s -m gpu -p no:cacheprovider "tests/test_parity_fullsize_gpu.py::test_configB_at_the_benchmarked_batch_vs_oracle" --durations=3 2>&1 | tail -12
cat $MPV_PARITY_REPORT
