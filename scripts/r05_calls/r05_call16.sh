#!/bin/bash
# r05 call 16: non-temporal hint on the GEMM's operand DMA (A / B / both / the activation operands only); tree = nt output stores + nt epilogue reads
cd $GRAFT_REPO_ROOT
run() { echo -n "$1 : "; if [ "$1" = tree ]; then env python bench.py --no-cpu-baseline --steps 30 2>/dev/null; else MPV_LIB_PATH=gpurun_ab/libmpv_hip_$1.so python bench.py --no-cpu-baseline --steps 30 2>/dev/null; fi | python -c "import json,sys;r=json.loads(sys.stdin.readline());ro=r['roofline'];print(r['ms_per_step'],'gemm ms',ro['gemm_ms_per_step'],'loss',r['config']['final_loss'],'sclk',ro.get('sclk_mhz'),'W',ro.get('power_w'), {k[:9]: v['ms_per_step'] for k, v in ro['by_kernel'].items()})"; }
(for round in 1 2; do for v in tree dma_a dma_b dma_ab dma_act; do run $v; done; done) 2>&1 | tee gpurun_out/r05_c16_nt_dma_ab.log
