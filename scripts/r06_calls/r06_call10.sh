#!/bin/bash
# r06 call 10: the step on a high-priority stream (the weight-gradient lane at normal priority) vs the default stream; bench with the distributed
# branch forced (self-check + measured mode choice); the graph tests incl. the self-check
cd $GRAFT_REPO_ROOT
OUT=gpurun_out
run() { python bench.py --no-cpu-baseline --steps 30 "$@" 2>/dev/null | python -c "import json,sys;r=json.loads(sys.stdin.readline());ro=r['roofline'];print(r['ms_per_step'],r['step_mode'],'gemm ms',ro['gemm_ms_per_step'],'loss',r['config']['final_loss'],'sclk',ro.get('sclk_mhz'),'W',ro.get('power_w'))"; }
(for round in 1 2 3; do for v in default high; do echo -n "MPV_BENCH_MAIN_PRIORITY=$v : "; MPV_BENCH_MAIN_PRIORITY=$v run; done; done) 2>&1 | tee $OUT/r06_c10_main_priority_ab.log
MPV_BENCH_FORCE_DIST=1 python bench.py --no-cpu-baseline --steps 20 > $OUT/r06_c10_bench_forced_dist_auto.json 2> $OUT/r06_c10_bench_forced_dist_auto.err; grep -E "self-check|host |timed" $OUT/r06_c10_bench_forced_dist_auto.err; python -c "import json;r=json.load(open('$OUT/r06_c10_bench_forced_dist_auto.json'));print(r['ms_per_step'],r['step_mode'],r['host'])"
timeout 900 python -m pytest -q -m gpu -p no:cacheprovider tests/test_model_gpu.py -k "graph" tests/test_entrypoint_gpu.py 2>&1 | tail -4
