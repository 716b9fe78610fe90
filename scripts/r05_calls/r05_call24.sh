#!/bin/bash
# r05 call 24: ViT dQ kernel with its LDS fragments requested a tile ahead (no MFMA behind its own ds_read + wait): attention tests, same-box
# alternation against the old scheduling (variant library), the kernel in a serialised trace
set -u
R=$GRAFT_REPO_ROOT; cd $R; OUT=$R/gpurun_out
timeout 900 python -m pytest -q -x -m gpu -p no:cacheprovider tests/test_kernels_gpu.py -k "attention or attn" 2>&1 | tail -2
timeout 600 python -m pytest -q -x -m gpu -p no:cacheprovider tests/test_model_gpu.py -k "tiny_vs_reference or configA" 2>&1 | tail -2
run() { echo -n "$1 : "; if [ "$1" = tree ]; then env python bench.py --no-cpu-baseline --no-roofline --steps 30 2>&1; else MPV_LIB_PATH=gpurun_ab/libmpv_hip_$1.so python bench.py --no-cpu-baseline --no-roofline --steps 30 2>&1; fi | grep -E "timed region" | sed 's/.*timed region done: //'; }
(for round in 1 2 3; do for v in tree dq_old; do run $v; done; done) 2>&1 | tee $OUT/r05_c24_dq_prefetch_ab.log
for arm in tree dq_old; do
  if [ $arm = dq_old ]; then export MPV_LIB_PATH=$R/gpurun_ab/libmpv_hip_dq_old.so; else unset MPV_LIB_PATH; fi
  (cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/kt_$arm; MPV_WGRAD_STREAM=0 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt_$arm -o t -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > /dev/null 2>&1; python $R/tools/rocpd_stats.py $(find /tmp/kt_$arm -name "*.db" | head -1) $OUT/r05_c24_kernel_trace_$arm.md > /dev/null)
  echo -n "$arm: "; grep -E "attn_bwd_dq_duo96" $OUT/r05_c24_kernel_trace_$arm.md | cut -c1-120
done 2>&1 | tee -a $OUT/r05_c24_dq_prefetch_ab.log
