"""Print which vendor kernels torch.matmul dispatches for the path's GEMM shapes (run under rocprofv3 --kernel-trace)."""
import torch
dev = torch.device("cuda:0")
for M, N, K in [(50432, 2304, 768), (50432, 768, 3072), (5120, 8192, 2048), (8192, 8192, 8192)]:
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    b = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        c = a @ b.t()
    torch.cuda.synchronize()
