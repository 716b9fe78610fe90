#!/bin/bash
# r05 call 15: which of the GEMM's global accesses gain from the non-temporal hint?  tree = nt on C and on the second outputs; variant libraries
# (gpurun_ab/): plain (round 4), nt on C only, on the second outputs only, + nt on the fp32 split-K partials, + nt on the epilogue's residual / GELU'
# reads.  Same box, two rounds.
cd $GRAFT_REPO_ROOT
run() { echo -n "$1 : "; if [ "$1" = tree ]; then env python bench.py --no-cpu-baseline --steps 30 2>/dev/null; else MPV_LIB_PATH=gpurun_ab/libmpv_hip_$1.so python bench.py --no-cpu-baseline --steps 30 2>/dev/null; fi | python -c "import json,sys;r=json.loads(sys.stdin.readline());ro=r['roofline'];print(r['ms_per_step'],'gemm ms',ro['gemm_ms_per_step'],'loss',r['config']['final_loss'],'sclk',ro.get('sclk_mhz'),'W',ro.get('power_w'), {k[:9]: v['ms_per_step'] for k, v in ro['by_kernel'].items()})"; }
(for round in 1 2; do for v in tree plain nt_c_only nt_2nd_only nt_f32 nt_ext nt_f32_ext; do run $v; done; done) 2>&1 | tee gpurun_out/r05_c15_nt_variants_ab.log
