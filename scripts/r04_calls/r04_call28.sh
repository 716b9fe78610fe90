#!/bin/bash
# GEMM finish branch-free: four staged chunks read, then stored through buffer stores over a descriptor of C (out-of-range offset = no store)
# GEMM tests on the new library, then same-box alternation against the previous one (gpurun_ab/libmpv_hip_base.so)
set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gemm256_gpu.py tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -k "gemm" 2>&1 | tail -3
for L in base new base new; do
  if [ $L = base ]; then export MPV_LIB_PATH=$R/gpurun_ab/libmpv_hip_base.so; else unset MPV_LIB_PATH; fi
  MPV_BENCH_BY_SHAPE=$OUT/r04_c28_by_shape_$L.md timeout 300 python bench.py --no-cpu-baseline > $OUT/r04_c28_bench_$L.json 2> $OUT/r04_c28_bench_$L.err
  python - <<PY
import json
r=json.load(open("$OUT/r04_c28_bench_$L.json"))
print("$L ms/step", r["ms_per_step"], "gemm ms", r["roofline"]["gemm_ms_per_step"], "frac", r["roofline"]["frac"], "loss", r["config"]["final_loss"])
PY
done
unset MPV_LIB_PATH
python - <<PY
def load(f):
    d={}
    for l in open(f):
        c=[x.strip() for x in l.split('|')]
        if len(c)>9 and c[1].startswith('gemm'):
            d[(c[1],c[2],c[3],c[4],c[5])]=(float(c[7]),float(c[6]))
    return d
a=load("$OUT/r04_c28_by_shape_base.md"); b=load("$OUT/r04_c28_by_shape_new.md")
for k in a:
    if k in b and a[k][1]>=12: print(k, "base", a[k][0], "new", b[k][0], f"{(b[k][0]/a[k][0]-1)*100:+.1f}%")
PY
