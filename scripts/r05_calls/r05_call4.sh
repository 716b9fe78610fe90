#!/bin/bash
# r05 call 4: g256p with the XCD-aware grouped walk (each XCD's L2 fetches a panel once) against the flat walk and the production kernel
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05_c4_g256p_xcd_walk.log; : > $OUT
cd tools/probe
for v in "" "-DABL_FLATWALK"; do
  n=g256p_$(echo "x$v" | sed 's/-D//g; s/ /_/g')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 $v -o /tmp/$n g256p_probe.hip 2>/dev/null
  shapes=("256 256 768" "1024 768 768" "2048 2304 768" "50432 768 768" "50432 2304 768" "50432 3072 768" "50432 768 3072" "5120 2048 8192" "5120 8192 2048" "5120 6144 2048")
  for shape in "${shapes[@]}"; do
    echo "== $n $shape" | tee -a $OUT
    timeout 60 /tmp/$n $shape 2>&1 | tail -2 | tee -a $OUT
  done
done
for gm in 1 2 8 16; do
  for shape in "50432 768 768" "50432 2304 768" "50432 768 3072"; do
    echo "== g256p_x GM=$gm $shape" | tee -a $OUT
    timeout 60 /tmp/g256p_x $shape 1 $gm 2>&1 | tail -1 | tee -a $OUT
  done
done
cd $GRAFT_REPO_ROOT; python - <<'P' 2>/dev/null | tee -a $OUT
import os, sys
sys.path.insert(0, os.getcwd())
import torch, youku_mplug_amd
from youku_mplug_amd import ops
from tools.bench_kernels import rnd, dev, timeit
for M, N, K in [(50432, 768, 768), (50432, 2304, 768), (50432, 3072, 768), (50432, 768, 3072), (5120, 2048, 8192), (5120, 8192, 2048), (5120, 6144, 2048)]:
    a, b, bias = rnd(M, K), rnd(N, K), rnd(N)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    t = timeit(lambda: ops.gemm(a, b, M, N, K, bias=bias, out=out))
    print(f"production  M={M} N={N} K={K}: {t * 1e6:.1f} us  {2 * M * N * K / t / 1e12:.1f} TFLOP/s")
P
