#!/bin/bash
# Round profile on the MI355X box: kernel trace + two PMC passes (FETCH_SIZE, WRITE_SIZE) of bench.py.
# Usage (from the repo root, on the GPU box): bash tools/profile_round.sh r01_final
set -u
TAG=${1:-r01}
R=$PWD
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# per-kernel passes run with the ViT weight-gradient lane off (one stream, kernels do not share the chip): durations and
# counters are then per-kernel properties; bench.py's timed region (the reported value) runs with the lane on
export MPV_WGRAD_STREAM=0
rm -rf /tmp/kt /tmp/pf /tmp/pw
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt -o $TAG -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/${TAG}_trace_bench.log 2>&1
python $R/tools/rocpd_stats.py $(find /tmp/kt -name "*.db" | head -1) $OUT/${TAG}_kernel_trace.md > /dev/null
timeout 400 rocprofv3 --pmc FETCH_SIZE -d /tmp/pf -o f -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/${TAG}_pmc_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE -d /tmp/pw -o w -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/${TAG}_pmc_write.log 2>&1
CALLS=$(grep -o "mpv_gemm_bf16 calls in this process: [0-9]*" $OUT/${TAG}_pmc_fetch.log | grep -o "[0-9]*$" | tail -1)
python $R/tools/rocpd_pmc.py $(find /tmp/pf -name "*.db" | head -1) $(find /tmp/pw -name "*.db" | head -1) $OUT/${TAG}_pmc_hbm_traffic.md $OUT/pmc_gemm_latest.json ${CALLS:-0} > /dev/null
# matrix-pipe utilisation (its own pass: SQ + GRBM counters only)
rm -rf /tmp/pm
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d /tmp/pm -o m -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/${TAG}_pmc_mfma.log 2>&1
python $R/tools/rocpd_mfma_util.py $(find /tmp/pm -name "*.db" | head -1) $OUT/${TAG}_pmc_mfma_util.md > /dev/null
# (tools/rocpd_overlap.py analyses an RCCL trace for all-reduce / backward overlap; on a 1-GPU box a 1-rank communicator launches no
# RCCL kernel at all -- measured: 0 launches under MPV_BENCH_FORCE_DIST=1 -- so that pass only makes sense on >= 2 GPUs)
head -12 $OUT/${TAG}_kernel_trace.md; head -14 $OUT/${TAG}_pmc_mfma_util.md; head -6 $OUT/${TAG}_pmc_hbm_traffic.md; cat $OUT/pmc_gemm_latest.json
