#!/bin/bash
# r05 call 8: (a) the 256x256 GEMM's DIRECT epilogue (plain / bias kind: accumulators -> permlane-merged 16-byte stores, no LDS staging, no
# barrier): every GEMM test under MPV_GEMM_DIRECT_EPI=1, then the step with / without it (same box, alternating) with the by-shape table;
# (b) is graph replay bit-identical to the eager step in bench.py, with and without the distributed branch forced?
cd $GRAFT_REPO_ROOT
MPV_GEMM_DIRECT_EPI=1 timeout 900 python -m pytest -q -x -m gpu -p no:cacheprovider tests/test_gemm256_gpu.py tests/test_kernels_gpu.py -k "gemm" > gpurun_out/r05_c8_gemm_tests_direct.log 2>&1
tail -4 gpurun_out/r05_c8_gemm_tests_direct.log
for round in 1 2; do for d in 0 1; do
  MPV_GEMM_DIRECT_EPI=$d MPV_BENCH_BY_SHAPE=gpurun_out/r05_c8_by_shape_direct$d.md timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r05_c8_bench_direct$d.json 2> gpurun_out/r05_c8_bench_direct$d.err
  python -c "
import json;r=json.load(open('gpurun_out/r05_c8_bench_direct$d.json'));ro=r['roofline']
print('direct=$d ms/step', r['ms_per_step'], 'gemm ms', ro['gemm_ms_per_step'], 'gemm_frac', ro['gemm_frac'], 'loss', r['config']['final_loss'], 'sclk', ro.get('sclk_mhz'), 'W', ro.get('power_w'))"
done; done 2>&1 | tee gpurun_out/r05_c8_direct_ab.log
python - <<'PY' | tee -a gpurun_out/r05_c8_direct_ab.log
def load(f):
    d={}
    for l in open(f):
        c=[x.strip() for x in l.split('|')]
        if len(c)>9 and c[1].startswith('gemm'):
            d[(c[1],c[2],c[3],c[4],c[5])]=(float(c[7]),float(c[6]),float(c[9]))
    return d
a=load("gpurun_out/r05_c8_by_shape_direct0.md"); b=load("gpurun_out/r05_c8_by_shape_direct1.md")
for k in a:
    if k in b and a[k][2]>=0.3: print(k, "staged us", a[k][0], "direct us", b[k][0], f"{(b[k][0]/a[k][0]-1)*100:+.1f}%")
PY
for mode in eager graph; do for dist in 0 1; do
  if [ $mode = graph ]; then export MPV_GRAPH=1; else unset MPV_GRAPH; fi
  if [ $dist = 1 ]; then export MPV_BENCH_FORCE_DIST=1; else unset MPV_BENCH_FORCE_DIST; fi
  timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys;r=json.loads(sys.stdin.readline());print('$mode dist=$dist', r['ms_per_step'], 'final_loss', r['config']['final_loss'])"
done; done 2>&1 | tee gpurun_out/r05_c8_graph_vs_eager_loss.log
