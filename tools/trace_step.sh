#!/bin/bash
# Kernel trace of a few bench steps only (no PMC passes): bash tools/trace_step.sh TAG  -> gpurun_out/TAG_kernel_trace.md
set -u
TAG=${1:-trace}
R=$PWD
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt -o $TAG -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/${TAG}_trace_bench.log 2>&1
python $R/tools/rocpd_stats.py $(find /tmp/kt -name "*.db" | head -1) $OUT/${TAG}_kernel_trace.md > /dev/null
