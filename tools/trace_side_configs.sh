#!/bin/bash
# Kernel traces of the two side configurations: bench.py --config D / E under rocprofv3 --kernel-trace --stats,
# weight-gradient lane off (one stream: per-kernel durations), summarised by tools/rocpd_stats.py.  Usage (GPU box): bash tools/trace_side_configs.sh <tag>
set -u
TAG=${1:-r03}
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp MPV_WGRAD_STREAM=0
for c in ${CONFIGS:-D E}; do
  rm -rf /tmp/kt_$c
  timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/kt_$c -o $c -- python $R/bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline --no-roofline < /dev/null > $OUT/${TAG}_trace_bench_$c.log 2>&1
  DB=$(find /tmp/kt_$c -name "*.db" | head -1)
  [ -n "$DB" ] && python $R/tools/rocpd_stats.py $DB $OUT/${TAG}_config${c}_kernel_trace.md > /dev/null
  head -8 $OUT/${TAG}_config${c}_kernel_trace.md | cut -c1-160; tail -1 $OUT/${TAG}_config${c}_kernel_trace.md
done
