#!/bin/bash
# LayerNorm backward with parameter gradients: next-row prefetch (two register sets, branch-free buffer accesses, 3 waves per SIMD, 768 workgroups)
set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
MPV_LN_BWD_PF=1 timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -k "layernorm or ln_ or norm" 2>&1 | tail -3
for L in 0 1 0 1; do
  MPV_LN_BWD_PF=$L timeout 300 python bench.py --no-cpu-baseline > $OUT/r04_c20_bench_pf$L.json 2> $OUT/r04_c20_bench_pf$L.err
  python - <<PY
import json
r=json.load(open("$OUT/r04_c20_bench_pf$L.json"))
print("pf=$L ms/step", r["ms_per_step"], "gemm ms", r["roofline"]["gemm_ms_per_step"], "loss", r["config"]["final_loss"])
PY
done
cd /tmp && export TMPDIR=/tmp
MPV_LN_BWD_PF=1 MPV_WGRAD_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/r04_c20_prof -o pf1 -- python $R/bench.py --no-cpu-baseline --steps 3 --warmup 2 > /dev/null 2>&1
f=$(ls $OUT/r04_c20_prof/*/pf1_kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && grep -E "ln_bwd8|ln_fwd8|Name" $f | cut -c1-200
