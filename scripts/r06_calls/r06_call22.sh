#!/bin/bash
# r06 call 22: the closing set of the LAST tree (after the glue kernels on more waves and the ragged-K weight gradient on the 256x256 kernel) -- GPU suite, smoke, kernel trace + PMC passes of the driver's command, instruction counters, the line of the driver's
# command with the in-step GEMM table, side lines D / E / Y
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out
rm -f $OUT/r06_parity.txt
timeout 1500 python -m pytest tests -m gpu -q -s -p no:cacheprovider > $OUT/r06_final2_gpu_tests.log 2>&1
grep -E "\[gate|not gated|passed|failed|FAILED" $OUT/r06_final2_gpu_tests.log | sort | uniq > $OUT/r06_final2_gates.txt
tail -2 $OUT/r06_final2_gpu_tests.log
python __graft_entry__.py smoke > $OUT/r06_final2_smoke.log 2>&1; tail -1 $OUT/r06_final2_smoke.log
bash tools/profile_round.sh r06_final2 > $OUT/r06_final2_profile_round.log 2>&1; tail -8 $OUT/r06_final2_profile_round.log | cut -c1-300
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/p2
MPV_WGRAD_STREAM=0 timeout 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES -d /tmp/p2 -o b -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/r06_final2_pmc_insts.log 2>&1
(python $R/tools/rocpd_counters.py $(find /tmp/p2 -name "*.db" | head -1) attn; python $R/tools/rocpd_counters.py $(find /tmp/p2 -name "*.db" | head -1) gemm256_kernel) > $OUT/r06_final2_sq_insts_attention_gemm.txt 2>&1
cd $R
cp $OUT/pmc_gemm_latest.json profiles/pmc_gemm_latest.json
MPV_BENCH_BY_SHAPE=$OUT/r06_final2_gemm_in_step_by_shape.md python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r06_final2_bench_B_1gpu.json 2> $OUT/r06_final2_bench_B_1gpu.err; tail -3 $OUT/r06_final2_bench_B_1gpu.err; cut -c1-700 $OUT/r06_final2_bench_B_1gpu.json
python bench.py --steps 50 --no-cpu-baseline > $OUT/r06_final2_bench_B_50steps.json 2> /dev/null; cut -c1-200 $OUT/r06_final2_bench_B_50steps.json
python bench.py --config D --no-cpu-baseline --steps 20 > $OUT/r06_final2_bench_D.json 2> /dev/null; cut -c1-250 $OUT/r06_final2_bench_D.json
python bench.py --config Y --no-cpu-baseline --steps 20 > $OUT/r06_final2_bench_Y.json 2> /dev/null; cut -c1-250 $OUT/r06_final2_bench_Y.json
python bench.py --config E --no-cpu-baseline --steps 10 > $OUT/r06_final2_bench_E.json 2> /dev/null; cut -c1-250 $OUT/r06_final2_bench_E.json
