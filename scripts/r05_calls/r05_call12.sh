#!/bin/bash
# r05 call 12: which kernels does the vendor library pick for the three forward shapes where it is 2-10 % faster (macro tile in the kernel name)?
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r05_c12_vendor -o vendor -- python $GRAFT_REPO_ROOT/tools/vendor_kernel_names.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/r05_c12_vendor -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'cut -c1-400 {} | head -12' | tee gpurun_out/r05_c12_vendor_kernel_names.log
