"""MEASUREMENT ONLY (nothing in the product calls a vendor BLAS): the step's plain / bias GEMM shapes through torch.matmul / F.linear (hipBLASLt /
rocBLAS under PyTorch-ROCm) beside mpv_gemm_bf16 on the same box -- how far is the hand-written kernel from what the vendor library reaches on
this part at these shapes?   python tools/gemm_vs_vendor.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
import youku_mplug_amd  # noqa: F401
from youku_mplug_amd import ops
from tools.bench_kernels import rnd, dev, timeit

print(f"{'form':8s} {'M':>6s} {'N':>6s} {'K':>6s} | {'mpv us':>8s} {'TF/s':>7s} | {'vendor us':>9s} {'TF/s':>7s} | vendor / mpv time")
for form, M, N, K in [("fwd+b", 50432, 2304, 768), ("fwd+b", 50432, 3072, 768), ("fwd+b", 50432, 768, 768), ("fwd+b", 50432, 768, 3072),
                      ("fwd+b", 5120, 6144, 2048), ("fwd+b", 5120, 2048, 8192), ("fwd+b", 5120, 8192, 2048), ("fwd+b", 5120, 2048, 2048),
                      ("dgrad", 50432, 768, 3072), ("dgrad", 50432, 768, 2304), ("dgrad", 50432, 768, 768), ("dgrad", 50432, 3072, 768),
                      ("wgrad", 768, 3072, 50432), ("wgrad", 2304, 768, 50432), ("wgrad", 768, 768, 50432)]:
    if form == "fwd+b":
        a, w, bias = rnd(M, K), rnd(N, K), rnd(N)
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        t0 = timeit(lambda: ops.gemm(a, w, M, N, K, bias=bias, out=out))
        t1 = timeit(lambda: F.linear(a, w, bias))
    elif form == "dgrad":          # dX[M, N] = dY[M, K] @ W[K, N]
        a, w = rnd(M, K), rnd(K, N)
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        t0 = timeit(lambda: ops.gemm(a, w, M, N, K, trans_b=True, out=out))
        t1 = timeit(lambda: torch.matmul(a, w))
    else:                          # dW[M, N] = dY[K, M]^T @ X[K, N]
        a, w = rnd(K, M), rnd(K, N)
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        t0 = timeit(lambda: ops.gemm(a, w, M, N, K, trans_a=True, trans_b=True, out=out))
        t1 = timeit(lambda: torch.matmul(a.t(), w))
    fl = 2.0 * M * N * K
    print(f"{form:8s} {M:6d} {N:6d} {K:6d} | {t0 * 1e6:8.1f} {fl / t0 / 1e12:7.0f} | {t1 * 1e6:9.1f} {fl / t1 / 1e12:7.0f} | {t1 / t0:.2f}")
