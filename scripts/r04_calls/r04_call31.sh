#!/bin/bash
# power / clocks of the final tree while the default bench runs (0.5 s sampling), and a serialised kernel trace of the same tree
set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
( while true; do date +%s.%N; rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk"; sleep 0.5; done ) > $OUT/r04_power_clocks.log 2>&1 &
SMI=$!
python bench.py --steps 100 --warmup 5 --no-cpu-baseline > $OUT/r04_c31_bench_B.json 2> $OUT/r04_c31_bench_B.err
kill $SMI
python - <<PY
import re,json
t=open("$OUT/r04_power_clocks.log").read()
p=[float(x) for x in re.findall(r"Power \(W\): ([0-9.]+)",t)]; c=[int(x) for x in re.findall(r"sclk clock level: \d+: \((\d+)Mhz\)",t)]
busy=[(a,b) for a,b in zip(p,c) if a>1000]
print("samples",len(p),"busy",len(busy),"power W min/med/max",min(a for a,_ in busy),sorted(a for a,_ in busy)[len(busy)//2],max(a for a,_ in busy),"sclk MHz min/med/max",min(b for _,b in busy),sorted(b for _,b in busy)[len(busy)//2],max(b for _,b in busy))
print(json.load(open("$OUT/r04_c31_bench_B.json"))["ms_per_step"])
PY
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/kt
MPV_WGRAD_STREAM=0 timeout 600 rocprofv3 --kernel-trace -d /tmp/kt -o t -- python $R/bench.py --no-cpu-baseline --no-roofline --steps 5 --warmup 2 > /dev/null 2>&1
python $R/tools/rocpd_stats.py $(find /tmp/kt -name "*.db" | head -1) $OUT/r04_final2_kernel_trace.md > /dev/null 2>&1
head -24 $OUT/r04_final2_kernel_trace.md | cut -c1-150
