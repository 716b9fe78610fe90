#!/bin/bash
# r05 call 20: the decoder's qkv product keeps its output cacheable (keep_output) for the attention behind it: tree vs the plain-store library,
# alternating, + the decoder attention forward in a serialised trace of the tree
set -u
R=$GRAFT_REPO_ROOT; cd $R; OUT=$R/gpurun_out
run() { echo -n "$1 : "; if [ "$1" = tree ]; then env python bench.py --no-cpu-baseline --steps 30 2>/dev/null; else MPV_LIB_PATH=gpurun_ab/libmpv_hip_$1.so python bench.py --no-cpu-baseline --steps 30 2>/dev/null; fi | python -c "import json,sys;r=json.loads(sys.stdin.readline());ro=r['roofline'];print(r['ms_per_step'],'gemm ms',ro['gemm_ms_per_step'],'loss',r['config']['final_loss'],'sclk',ro.get('sclk_mhz'),'W',ro.get('power_w'))"; }
(for round in 1 2 3; do for v in tree plain; do run $v; done; done) 2>&1 | tee $OUT/r05_c20_keep_output_ab.log
(cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/kt; MPV_WGRAD_STREAM=0 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt -o t -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > /dev/null 2>&1; python $R/tools/rocpd_stats.py $(find /tmp/kt -name "*.db" | head -1) $OUT/r05_c20_kernel_trace_tree.md > /dev/null)
grep -E "attn_fwd_pair64|ln_bwd8|adamw|attn_bwd_dq_pair" $OUT/r05_c20_kernel_trace_tree.md | cut -c1-140 | tee -a $OUT/r05_c20_keep_output_ab.log
timeout 600 python -m pytest -q -x -m gpu -p no:cacheprovider tests/test_gemm256_gpu.py 2>&1 | tail -2
