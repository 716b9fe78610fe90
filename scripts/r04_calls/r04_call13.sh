#!/bin/bash
set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
for P in 1 0 1 0; do
  MPV_GEMM_BANDS=$P MPV_BENCH_BY_SHAPE=$OUT/r04_c13_by_shape_bands$P.md timeout 300 python bench.py --no-cpu-baseline > $OUT/r04_c13_bench_bands$P.json 2> $OUT/r04_c13_bench_bands$P.err
  python - <<PY
import json
r=json.load(open("$OUT/r04_c13_bench_bands$P.json"))
print("bands=$P ms/step", r["ms_per_step"], "gemm ms", r["roofline"]["gemm_ms_per_step"], "frac", r["roofline"]["frac"])
PY
done
python - <<PY
def load(f):
    d={}
    for l in open(f):
        c=[x.strip() for x in l.split('|')]
        if len(c)>9 and c[1].startswith('gemm'):
            d[(c[1],c[2],c[3],c[4],c[5])]=(float(c[7]),float(c[6]))
    return d
a=load("$OUT/r04_c13_by_shape_bands1.md"); b=load("$OUT/r04_c13_by_shape_bands0.md")
for k in a:
    if k in b and a[k][1]>=1 and abs(a[k][0]-b[k][0])/a[k][0]>0.02:
        print(k, "bands", a[k][0], "single", b[k][0], "x", a[k][1])
PY
