#!/bin/bash
# r05 call 23: where do the wave cycles of the attention kernels go?  Two --pmc passes over one step (kernel-trace / stats off, counters only):
# (1) SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS   (2) SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out
cd /tmp; export TMPDIR=/tmp MPV_WGRAD_STREAM=0
rm -rf /tmp/p1 /tmp/p2
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS -d /tmp/p1 -o a -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/r05_c23_pmc1.log 2>&1
timeout 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES -d /tmp/p2 -o b -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/r05_c23_pmc2.log 2>&1
for p in p1 p2; do python $R/tools/rocpd_counters.py $(find /tmp/$p -name "*.db" | head -1) attn; python $R/tools/rocpd_counters.py $(find /tmp/$p -name "*.db" | head -1) gemm256_kernel; done > $OUT/r05_c23_sq_counters_attention_gemm.txt 2>&1
head -80 $OUT/r05_c23_sq_counters_attention_gemm.txt
