#!/bin/bash
set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt
MPV_WGRAD_STREAM=0 timeout 300 rocprofv3 --kernel-trace -d /tmp/kt -o st -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $OUT/r04_c7_trace_bench.log 2>&1
DB=$(find /tmp/kt -name "*.db" | head -1)
python $R/tools/rocpd_gaps.py $DB $OUT/r04_c7_gaps_eager.md 8
rm -rf /tmp/kg
MPV_GRAPH=1 MPV_WGRAD_STREAM=0 timeout 300 rocprofv3 --kernel-trace -d /tmp/kg -o st -- python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-roofline > $OUT/r04_c7_trace_bench_graph.log 2>&1
DB=$(find /tmp/kg -name "*.db" | head -1)
python $R/tools/rocpd_gaps.py $DB $OUT/r04_c7_gaps_graph.md 8 | head -12
