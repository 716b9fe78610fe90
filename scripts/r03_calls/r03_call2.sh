#!/bin/bash
# Round 3, GPU call 2: fp32 residual stream + glue kernels: kernel / model / full-size parity tests, same-box A/B of the stream
# dtype, side-line bench lines (configs D and E), and a power / clock log of one bench run.
set -u
R=$PWD
OUT=$R/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu 2>&1 | tail -25) > $OUT/c2_kernel_tests.log
tail -3 $OUT/c2_kernel_tests.log
(timeout 1200 python -m pytest tests/test_model_gpu.py tests/test_entrypoint_gpu.py -x -q -m gpu 2>&1 | tail -25) > $OUT/c2_model_tests.log
tail -3 $OUT/c2_model_tests.log
rm -f $OUT/r03_parity.txt
(timeout 1500 python -m pytest tests/test_parity_fullsize_gpu.py -q -m gpu 2>&1 | tail -40) > $OUT/c2_parity_tests.log
tail -5 $OUT/c2_parity_tests.log
cat $OUT/r03_parity.txt
run() { "$@" python bench.py --no-cpu-baseline --no-roofline --steps 30 2>&1 | grep -E "timed region" | sed 's/.*\] //' | tr '\n' ' '; echo; }
for round in 1 2; do for v in bf16 fp32; do echo -n "MPV_DECODER_STREAM=$v  "; run env MPV_DECODER_STREAM=$v; done; done > $OUT/c2_ab_stream.log 2>&1
cat $OUT/c2_ab_stream.log
# power / clocks while the default bench runs (0.5 s sampling)
( while true; do date +%s.%N; rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk|fclk"; sleep 0.5; done ) > $OUT/r03_power_clocks.log 2>&1 &
SMI=$!
python bench.py --steps 50 --warmup 5 > $OUT/c2_bench_B.json 2> $OUT/c2_bench_B.err
kill $SMI
tail -1 $OUT/c2_bench_B.json | cut -c1-600
python bench.py --config D --steps 20 --warmup 3 --no-cpu-baseline > $OUT/c2_bench_D.json 2> $OUT/c2_bench_D.err
tail -1 $OUT/c2_bench_D.json | cut -c1-400
python bench.py --config E --steps 10 --warmup 2 --no-cpu-baseline > $OUT/c2_bench_E.json 2> $OUT/c2_bench_E.err
tail -1 $OUT/c2_bench_E.json | cut -c1-400; tail -3 $OUT/c2_bench_E.err
