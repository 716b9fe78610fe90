#!/bin/bash
# r05 call 21: the n-tiles walked in column PARTS, part-major (an XCD's contiguous tile range then needs only its part's B panels: B of the
# N = 3072 products is 4.7 MB against a 4 MiB L2) -- GEMM tests under the knob, then same-box runs with the by-shape table
set -u
R=$GRAFT_REPO_ROOT; cd $R; OUT=$R/gpurun_out
for np in 2 3; do MPV_GEMM_NPARTS=$np timeout 600 python -m pytest -q -x -m gpu -p no:cacheprovider tests/test_gemm256_gpu.py 2>&1 | tail -1; done
run() { tag=$1; shift; env "$@" MPV_BENCH_BY_SHAPE=$OUT/r05_c21_by_shape_$tag.md python bench.py --no-cpu-baseline --steps 30 2>/dev/null | python -c "import json,sys;r=json.loads(sys.stdin.readline());ro=r['roofline'];print('$tag', r['ms_per_step'],'gemm ms',ro['gemm_ms_per_step'],'loss',r['config']['final_loss'],'sclk',ro.get('sclk_mhz'))"; }
(run base A=1; run np2 MPV_GEMM_NPARTS=2; run np3 MPV_GEMM_NPARTS=3; run np4 MPV_GEMM_NPARTS=4; run np2vit MPV_GEMM_NPARTS=2 MPV_GEMM_NPARTS_MINM=20000; run np3vit MPV_GEMM_NPARTS=3 MPV_GEMM_NPARTS_MINM=20000; run np4vit MPV_GEMM_NPARTS=4 MPV_GEMM_NPARTS_MINM=20000; run base2 A=1) 2>&1 | tee $OUT/r05_c21_nparts.log
python - <<'PY' | tee -a $OUT/r05_c21_nparts.log
def shapes(f):
    d={}
    for l in open(f):
        c=[x.strip() for x in l.split('|')]
        if len(c)>9 and c[1].startswith('gemm'):
            d[(c[1],c[2],c[3],c[4],c[5])]=(float(c[7]),float(c[6]),float(c[9]))
    return d
tabs={t: shapes(f"gpurun_out/r05_c21_by_shape_{t}.md") for t in ("base","np2","np3","np4","base2")}
for k in sorted(tabs["base"], key=lambda k:-tabs["base"][k][2]):
    if tabs["base"][k][2] >= 0.8: print(k, " ".join(f"{t}={tabs[t][k][0]:.1f}" for t in tabs if k in tabs[t]))
PY
