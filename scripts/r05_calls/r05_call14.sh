#!/bin/bash
# r05 call 14: the GEMM finish pass's output stores as non-temporal (`nt`) and as write-through (`sc1`) stores -- variant libraries, same box
cd $GRAFT_REPO_ROOT
run() { echo -n "$1 : "; if [ "$1" = tree ]; then env python bench.py --no-cpu-baseline --steps 30 2>/dev/null; else MPV_LIB_PATH=gpurun_ab/libmpv_hip_$1.so python bench.py --no-cpu-baseline --steps 30 2>/dev/null; fi | python -c "import json,sys;r=json.loads(sys.stdin.readline());ro=r['roofline'];print(r['ms_per_step'],'gemm ms',ro['gemm_ms_per_step'],'loss',r['config']['final_loss'],'sclk',ro.get('sclk_mhz'),'W',ro.get('power_w'))"; }
(run tree; run nt; run sc1; run tree; run nt; run sc1) 2>&1 | tee gpurun_out/r05_c14_store_flavours_ab.log
