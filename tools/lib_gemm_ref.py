"""What the vendor library (torch.matmul -> hipBLASLt/rocBLAS) reaches on the path's GEMM shapes.
Measurement aid only: the product never calls it."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, youku_mplug_amd
from youku_mplug_amd import ops
from tools.bench_kernels import timeit, rnd, dev
shapes = [(50432, 2304, 768), (50432, 768, 768), (50432, 3072, 768), (50432, 768, 3072), (50432, 2304, 6144),
          (5120, 6144, 2048), (5120, 2048, 2048), (5120, 8192, 2048), (5120, 2048, 8192), (8192, 8192, 8192)]
for M, N, K in shapes:
    a, b = rnd(M, K), rnd(N, K)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    t0 = timeit(lambda: ops.gemm(a, b, M, N, K, out=out))
    bt = b.t()
    t1 = timeit(lambda: torch.matmul(a, bt, out=out))
    fl = 2.0 * M * N * K / 1e12
    print(f"M={M:6d} N={N:5d} K={K:5d}  mpv {t0*1e6:8.1f} us {fl/t0:7.1f} TF/s   lib {t1*1e6:8.1f} us {fl/t1:7.1f} TF/s")
