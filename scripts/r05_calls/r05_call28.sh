#!/bin/bash
# r05 call 28: the split-K reduce (on the weight-gradient lane, beside the main stream's GEMMs) with non-temporal partial loads / gradient stores
cd $GRAFT_REPO_ROOT
run() { echo -n "$1 : "; if [ "$1" = tree ]; then env python bench.py --no-cpu-baseline --steps 30 2>/dev/null; else MPV_LIB_PATH=gpurun_ab/libmpv_hip_$1.so python bench.py --no-cpu-baseline --steps 30 2>/dev/null; fi | python -c "import json,sys;r=json.loads(sys.stdin.readline());ro=r['roofline'];print(r['ms_per_step'],'gemm ms',ro['gemm_ms_per_step'],'loss',r['config']['final_loss'],'sclk',ro.get('sclk_mhz'))"; }
(for round in 1 2 3; do for v in tree rednt; do run $v; done; done) 2>&1 | tee gpurun_out/r05_c28_reduce_nt_ab.log
