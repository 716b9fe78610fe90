#!/bin/bash
# r06 call 4: the producer-scaled-q test (fixed), eager vs graph replay at N = 1 (alternating), forward split-K threshold on config D, side lines D / E
cd $GRAFT_REPO_ROOT
OUT=gpurun_out
timeout 900 python -m pytest -q -m gpu -p no:cacheprovider tests/test_kernels_gpu.py -k "producer" 2>&1 | tail -4
run() { python bench.py --no-cpu-baseline --steps 30 "$@" 2>/dev/null | python -c "import json,sys;r=json.loads(sys.stdin.readline());ro=r['roofline'];print(r['ms_per_step'],r['step_mode'],'host cpu',r['host']['cpu_ms_per_step'],'gemm ms',ro['gemm_ms_per_step'],'loss',r['config']['final_loss'],'sclk',ro.get('sclk_mhz'),'W',ro.get('power_w'))"; }
(for round in 1 2 3; do for m in eager graph; do echo -n "--step-mode $m : "; run --step-mode $m; done; done) 2>&1 | tee $OUT/r06_c4_eager_vs_graph_n1.log
(for round in 1 2; do for k in 4096 2048; do echo -n "config D MPV_FWD_SPLIT_MINK=$k : "; MPV_FWD_SPLIT_MINK=$k run --config D; done; done) 2>&1 | tee $OUT/r06_c4_configD_fwd_split.log
python bench.py --config E --no-cpu-baseline --steps 10 > $OUT/r06_c4_bench_E.json 2> $OUT/r06_c4_bench_E.err; tail -2 $OUT/r06_c4_bench_E.err; cut -c1-400 $OUT/r06_c4_bench_E.json
