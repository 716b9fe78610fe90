#!/bin/bash
# r04 GPU call 5: whole GPU suite on the tree with the parked GELU' (tanh: one polynomial for both), then the measurement set:
# default bench line (+ by-shape), deriv A/B, graph replay line (host CPU per step with the polling queue bound), configs D / E.
set -u
R=$PWD
OUT=$R/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
python -m pytest tests -m gpu -q --durations=5 -p no:cacheprovider > $OUT/r04_c5_gpu_tests.log 2>&1
echo "pytest rc=$?" >> $OUT/r04_c5_gpu_tests.log
grep -v "^E   \|^    \|^$" $OUT/r04_c5_gpu_tests.log | tail -14
for P in 0 1 0 1; do
  MPV_GELU_DERIV=$P MPV_BENCH_BY_SHAPE=$OUT/r04_c5_by_shape_deriv$P.md timeout 300 python bench.py --no-cpu-baseline > $OUT/r04_c5_bench_deriv$P.json 2> $OUT/r04_c5_bench_deriv$P.err
  python - <<PY
import json
r=json.load(open("$OUT/r04_c5_bench_deriv$P.json"))
print("deriv=$P ms/step", r["ms_per_step"], "gemm ms", r["roofline"]["gemm_ms_per_step"], "frac", r["roofline"]["frac"], "loss", r["config"]["final_loss"])
PY
done
grep "act_bwd_z\|preact_out" $OUT/r04_c5_by_shape_deriv0.md | head -4
grep "act_bwd_z\|preact_out" $OUT/r04_c5_by_shape_deriv1.md | head -4
MPV_GRAPH=1 timeout 300 python bench.py --no-cpu-baseline --no-roofline > $OUT/r04_c5_bench_graph.json 2> $OUT/r04_c5_bench_graph.err
cat $OUT/r04_c5_bench_graph.json | cut -c1-200; grep "host" $OUT/r04_c5_bench_graph.err
timeout 300 python bench.py --config D --no-cpu-baseline > $OUT/r04_c5_bench_D.json 2> $OUT/r04_c5_bench_D.err; cut -c1-330 $OUT/r04_c5_bench_D.json
timeout 400 python bench.py --config E --no-cpu-baseline --steps 10 --warmup 3 > $OUT/r04_c5_bench_E.json 2> $OUT/r04_c5_bench_E.err; cut -c1-330 $OUT/r04_c5_bench_E.json
