#!/bin/bash
# r06 call 14: the whole GPU suite + smoke on the last tree (after the layout fingerprint, the memory guard and the second benchmarked-batch parity case)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out
rm -f $OUT/r06_parity.txt
timeout 1800 python -m pytest tests -m gpu -q -s -p no:cacheprovider > $OUT/r06_last_gpu_tests.log 2>&1
grep -E "\[gate|not gated|passed|failed|FAILED" $OUT/r06_last_gpu_tests.log | sort | uniq > $OUT/r06_last_gates.txt
grep -E "passed|failed" $OUT/r06_last_gpu_tests.log | tail -2
python __graft_entry__.py smoke 2>&1 | tail -1
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r06_last_bench_B_1gpu.json 2> $OUT/r06_last_bench_B_1gpu.err; python -c "import json;r=json.load(open('$OUT/r06_last_bench_B_1gpu.json'));print(r['ms_per_step'],r['value'],r['roofline']['frac'],r['roofline']['gemm_frac'],r['roofline']['sclk_mhz'],r['roofline']['power_w'])"
