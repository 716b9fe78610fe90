#!/bin/bash
# r05 call 19: WHERE does the non-temporal hint on the GEMM's output stores pay, and which consumer kernels pay for it?  tree vs the plain-store
# library on one box: in-step by-shape tables + serialised kernel traces (weight-gradient lane off) of both arms
set -u
R=$GRAFT_REPO_ROOT; cd $R; OUT=$R/gpurun_out
for arm in tree plain; do
  if [ $arm = plain ]; then export MPV_LIB_PATH=$R/gpurun_ab/libmpv_hip_plain.so; else unset MPV_LIB_PATH; fi
  MPV_BENCH_BY_SHAPE=$OUT/r05_c19_by_shape_$arm.md timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/r05_c19_bench_$arm.json 2>/dev/null
  python -c "import json;r=json.load(open('$OUT/r05_c19_bench_$arm.json'));print('$arm', r['ms_per_step'], r['roofline']['gemm_ms_per_step'])"
  (cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/kt_$arm; MPV_WGRAD_STREAM=0 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt_$arm -o t -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > /dev/null 2>&1; python $R/tools/rocpd_stats.py $(find /tmp/kt_$arm -name "*.db" | head -1) $OUT/r05_c19_kernel_trace_$arm.md > /dev/null)
done
python - <<'PY' | tee $OUT/r05_c19_nt_where.log
import re
def shapes(f):
    d={}
    for l in open(f):
        c=[x.strip() for x in l.split('|')]
        if len(c)>9 and c[1].startswith('gemm'):
            d[(c[1],c[2],c[3],c[4],c[5])]=(float(c[7]),float(c[6]),float(c[9]))
    return d
a=shapes("gpurun_out/r05_c19_by_shape_plain.md"); b=shapes("gpurun_out/r05_c19_by_shape_tree.md")
print("GEMM shapes in the step: plain stores -> non-temporal (us per launch, ms per step)")
for k in sorted(a, key=lambda k:-a[k][2]):
    if k in b and a[k][2]>=0.1: print(f"  {k}: {a[k][0]:.1f} -> {b[k][0]:.1f} us ({(b[k][0]/a[k][0]-1)*100:+.1f}%)  {a[k][2]:.2f} -> {b[k][2]:.2f} ms")
def trace(f):
    d={}
    for l in open(f):
        c=[x.strip() for x in l.split('|')]
        if len(c)>7 and c[2].isdigit(): d[c[1]]=(int(c[2]),float(c[3]),float(c[4]))
    return d
a=trace("gpurun_out/r05_c19_kernel_trace_plain.md"); b=trace("gpurun_out/r05_c19_kernel_trace_tree.md")
print("kernels (serialised trace, 4 steps): plain -> non-temporal, avg us; total ms")
for k in sorted(a, key=lambda k:-a[k][1]):
    if k in b and a[k][1]>=0.5 and "at::native" not in k: print(f"  {k[:70]}: {a[k][2]:.1f} -> {b[k][2]:.1f} us  total {a[k][1]:.2f} -> {b[k][1]:.2f} ms ({b[k][1]-a[k][1]:+.2f})")
PY
