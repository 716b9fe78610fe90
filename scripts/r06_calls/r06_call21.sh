#!/bin/bash
# r06 call 21: the embedding-gradient sums and the cross entropy on more waves per workgroup: whole GPU suite, smoke, their times in a trace, the driver's command
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out
rm -f $OUT/r06_parity.txt
timeout 1800 python -m pytest tests -m gpu -q -s -p no:cacheprovider > $OUT/r06_last_gpu_tests.log 2>&1
grep -E "\[gate|not gated|passed|failed|FAILED" $OUT/r06_last_gpu_tests.log | sort | uniq > $OUT/r06_last_gates.txt
grep -E "passed|failed" $OUT/r06_last_gpu_tests.log | tail -2
python __graft_entry__.py smoke 2>&1 | tail -1
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/kt
MPV_WGRAD_STREAM=0 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt -o t -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > /dev/null 2>&1
python $R/tools/rocpd_stats.py $(find /tmp/kt -name "*.db" | head -1) $OUT/r06_c21_kernel_trace.md > /dev/null
grep -E "embed_bwd|cross_entropy|grad_sumsq" $OUT/r06_c21_kernel_trace.md | cut -c1-160
cd $R
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r06_last_bench_B_1gpu.json 2> $OUT/r06_last_bench_B_1gpu.err; python -c "import json;r=json.load(open('$OUT/r06_last_bench_B_1gpu.json'));print(r['ms_per_step'],r['value'],r['roofline']['frac'],r['roofline']['gemm_frac'],r['roofline']['sclk_mhz'],r['roofline']['power_w'],r['config']['final_loss'])"
