#!/bin/bash
# r05 call 5: is the g256p loop bound by HALF-LINE requests?  (K-steps of 32 on k-contiguous rows = 64-byte pieces: every 128-byte line
# is requested by two K-steps.)  Timing-only build whose DMA instructions read whole lines, same bytes / instruction count per K-step.
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05_c5_g256p_full_lines.log; : > $OUT
cd tools/probe
for v in "-DABL_FULLLINE" ""; do
  n=g256p_$(echo "x$v" | sed 's/-D//g; s/ /_/g')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 $v -o /tmp/$n g256p_probe.hip 2>&1 | grep -A3 "error" | head -8
  for shape in "50432 768 768" "50432 2304 768" "50432 768 3072" "5120 2048 8192"; do
    echo "== $n $shape" | tee -a $OUT
    timeout 60 /tmp/$n $shape 2>&1 | tail -2 | tee -a $OUT
  done
done
