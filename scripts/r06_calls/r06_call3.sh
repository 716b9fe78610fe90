#!/bin/bash
# r06 call 3: new tests (batched launches, colscale epilogue, producer-scaled q), then same-box alternations: q scaled by the qkv product vs by the
# attention kernels; by-shape GEMM table (the kmapped weight gradients without a divide per K-tile); then the whole GPU suite
cd $GRAFT_REPO_ROOT
OUT=gpurun_out
timeout 900 python -m pytest -q -x -m gpu -p no:cacheprovider tests/test_kernels_gpu.py -k "batched or colscale or producer or gemm_row_maps or wgrad" 2>&1 | tail -5
run() { python bench.py --no-cpu-baseline --steps 30 2>/dev/null | python -c "import json,sys;r=json.loads(sys.stdin.readline());ro=r['roofline'];print(r['ms_per_step'],'gemm ms',ro['gemm_ms_per_step'],'launches',ro['launches_per_step'],'loss',r['config']['final_loss'],'sclk',ro.get('sclk_mhz'),'W',ro.get('power_w'))"; }
(for round in 1 2 3; do for v in 1 0; do echo -n "MPV_VIT_PRESCALE_Q=$v : "; MPV_VIT_PRESCALE_Q=$v run; done; done) 2>&1 | tee $OUT/r06_c3_prescale_q_ab.log
MPV_BENCH_BY_SHAPE=$OUT/r06_c3_gemm_by_shape.md python bench.py --no-cpu-baseline --steps 20 > $OUT/r06_c3_bench_B.json 2> $OUT/r06_c3_bench_B.err; grep -E "kmap|768 \| 50176|2304 \| 768 \| 50" $OUT/r06_c3_gemm_by_shape.md
rm -f gpurun_out/r06_parity.txt
timeout 1500 python -m pytest tests -m gpu -q -s -p no:cacheprovider > $OUT/r06_c3_gpu_tests_full.log 2>&1
grep -E "\[gate|not gated|passed|failed|FAILED|Error" $OUT/r06_c3_gpu_tests_full.log | sort | uniq > $OUT/r06_c3_gates.txt
tail -3 $OUT/r06_c3_gpu_tests_full.log
