#!/bin/bash
# r05 call 17: non-temporal hint on the LayerNorm kernels' loads / stores / both (pure streaming kernels: the question is what they leave in
# L2 / the Infinity Cache for the GEMM behind them); variant libraries, same box
cd $GRAFT_REPO_ROOT
run() { echo -n "$1 : "; if [ "$1" = tree ]; then env python bench.py --no-cpu-baseline --steps 30 2>/dev/null; else MPV_LIB_PATH=gpurun_ab/libmpv_hip_$1.so python bench.py --no-cpu-baseline --steps 30 2>/dev/null; fi | python -c "import json,sys;r=json.loads(sys.stdin.readline());ro=r['roofline'];print(r['ms_per_step'],'gemm ms',ro['gemm_ms_per_step'],'loss',r['config']['final_loss'],'sclk',ro.get('sclk_mhz'),'W',ro.get('power_w'))"; }
(for round in 1 2; do for v in tree ln_ld ln_st ln_both; do run $v; done; done) 2>&1 | tee gpurun_out/r05_c17_nt_layernorm_ab.log
