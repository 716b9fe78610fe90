#!/bin/bash
# r06 call 23: config B with the forward split-K threshold at K = 2048 (the abstractor's and the top decoder layer's few-tile products) vs 4096
cd $GRAFT_REPO_ROOT
OUT=gpurun_out
run() { python bench.py --no-cpu-baseline --steps 30 2>/dev/null | python -c "import json,sys;r=json.loads(sys.stdin.readline());ro=r['roofline'];print(r['ms_per_step'],'gemm ms',ro['gemm_ms_per_step'],'loss',r['config']['final_loss'],'sclk',ro.get('sclk_mhz'),'W',ro.get('power_w'))"; }
(for round in 1 2 3; do for k in 4096 2048; do echo -n "MPV_FWD_SPLIT_MINK=$k : "; MPV_FWD_SPLIT_MINK=$k run; done; done) 2>&1 | tee $OUT/r06_c23_fwd_split_configB.log
MPV_FWD_SPLIT_MINK=2048 MPV_BENCH_BY_SHAPE=$OUT/r06_c23_by_shape_mink2048.md python bench.py --no-cpu-baseline --steps 10 > /dev/null 2>&1
awk -F'|' 'NR>2 && $9+0 < 700 {print}' $OUT/r06_c23_by_shape_mink2048.md | cut -c1-140
