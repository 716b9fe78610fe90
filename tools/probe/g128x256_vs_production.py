"""The production GEMM (bias epilogue) on the shapes tools/probe/g128x256_probe.hip is run on -- same box, same call."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, youku_mplug_amd
from youku_mplug_amd import ops
from tools.bench_kernels import rnd, dev, timeit
for M, N, K in [(50432, 2304, 768), (50432, 3072, 768), (50432, 768, 768), (50432, 768, 3072), (5120, 2048, 8192)]:
    a, b, bias = rnd(M, K), rnd(N, K), rnd(N)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    t = timeit(lambda: ops.gemm(a, b, M, N, K, bias=bias, out=out))
    print(f"production  M={M} N={N} K={K}: {t * 1e6:.1f} us  {2 * M * N * K / t / 1e12:.1f} TFLOP/s")
