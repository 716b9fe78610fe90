#!/bin/bash
# r06 call 2: (a) the GPU suite on the round-6 tree with every gate's measurement printed (plain-number gates, wide gradient samples of every
# tensor, batched composed-projection launches); (b) same-box alternation: composed projection grouped (5 launches) vs per block (60);
# (c) bench --config Y; (d) the distributed branch forced at world 1: graph self-check + segmented replay by default
cd $GRAFT_REPO_ROOT
OUT=gpurun_out
rm -f gpurun_out/r06_parity.txt
timeout 1500 python -m pytest tests -m gpu -q -s -p no:cacheprovider > $OUT/r06_c2_gpu_tests_full.log 2>&1
grep -E "\[gate|not gated|passed|failed|FAILED|Error" $OUT/r06_c2_gpu_tests_full.log | sort | uniq > $OUT/r06_c2_gates.txt
tail -3 $OUT/r06_c2_gpu_tests_full.log

run() { python bench.py --no-cpu-baseline --steps 30 2>/dev/null | python -c "import json,sys;r=json.loads(sys.stdin.readline());ro=r['roofline'];print(r['ms_per_step'],'gemm ms',ro['gemm_ms_per_step'],'launches',ro['launches_per_step'],'loss',r['config']['final_loss'],'sclk',ro.get('sclk_mhz'),'W',ro.get('power_w'))"; }
(for round in 1 2 3; do for v in 1 0; do echo -n "MPV_VIT_COMPOSE_GROUP=$v : "; MPV_VIT_COMPOSE_GROUP=$v run; done; done) 2>&1 | tee $OUT/r06_c2_compose_group_ab.log
python bench.py --config Y --no-cpu-baseline --steps 20 > $OUT/r06_c2_bench_Y.json 2> $OUT/r06_c2_bench_Y.err; tail -2 $OUT/r06_c2_bench_Y.err; cat $OUT/r06_c2_bench_Y.json
MPV_BENCH_FORCE_DIST=1 python bench.py --no-cpu-baseline --steps 20 > $OUT/r06_c2_bench_forced_dist_auto.json 2> $OUT/r06_c2_bench_forced_dist_auto.err; grep -E "self-check|host |timed" $OUT/r06_c2_bench_forced_dist_auto.err; cat $OUT/r06_c2_bench_forced_dist_auto.json
python bench.py --no-cpu-baseline --steps 20 --step-mode graph > $OUT/r06_c2_bench_graph_n1.json 2> $OUT/r06_c2_bench_graph_n1.err; grep -E "self-check|host |timed" $OUT/r06_c2_bench_graph_n1.err
python bench.py --no-cpu-baseline --steps 20 > $OUT/r06_c2_bench_eager_n1.json 2> $OUT/r06_c2_bench_eager_n1.err; grep -E "host |timed" $OUT/r06_c2_bench_eager_n1.err
