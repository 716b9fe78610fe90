#!/bin/bash
# two half-batch steps side by side on one GPU (two processes, each its own HIP queues) against one full-batch step: is there
# throughput to be had from running the LayerNorm / attention kernels of one half beside the GEMMs of the other?
set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
L=$OUT/r04_c21_two_half_batches.log; : > $L
one() { python - "$1" <<'PY'
import json,sys
r=json.load(open(sys.argv[1])); print(r["ms_per_step"], r["value"], r["config"]["global_batch"])
PY
}
timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 40 > $OUT/c21_full.json 2>/dev/null; echo "full batch 32 alone: $(one $OUT/c21_full.json)" | tee -a $L
timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 40 --batch 16 > $OUT/c21_half.json 2>/dev/null; echo "half batch 16 alone: $(one $OUT/c21_half.json)" | tee -a $L
for rep in 1 2; do
  timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 120 --warmup 10 --batch 16 > $OUT/c21_a.json 2>/dev/null &
  PA=$!
  timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 120 --warmup 10 --batch 16 > $OUT/c21_b.json 2>/dev/null &
  PB=$!
  wait $PA; wait $PB
  echo "two half batches side by side (rep $rep): A $(one $OUT/c21_a.json) | B $(one $OUT/c21_b.json)" | tee -a $L
done
timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 40 > $OUT/c21_full2.json 2>/dev/null; echo "full batch 32 alone (again): $(one $OUT/c21_full2.json)" | tee -a $L
