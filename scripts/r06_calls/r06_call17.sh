#!/bin/bash
# r06 call 17: where the wave cycles of the attention and GEMM kernels go on the closing tree (issuing / dependency-stalled / parked), one --pmc pass
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out
cd /tmp; export TMPDIR=/tmp MPV_WGRAD_STREAM=0
rm -rf /tmp/p1
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS -d /tmp/p1 -o a -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/r06_c17_pmc1.log 2>&1
(python $R/tools/rocpd_counters.py $(find /tmp/p1 -name "*.db" | head -1) attn; python $R/tools/rocpd_counters.py $(find /tmp/p1 -name "*.db" | head -1) gemm256_kernel; python $R/tools/rocpd_counters.py $(find /tmp/p1 -name "*.db" | head -1) gemm_bf16_kernel) > $OUT/r06_c17_sq_wave_cycles_attention_gemm.txt 2>&1
head -40 $OUT/r06_c17_sq_wave_cycles_attention_gemm.txt
