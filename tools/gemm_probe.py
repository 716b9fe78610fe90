import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, youku_mplug_amd
from youku_mplug_amd import ops
from tools.bench_kernels import rnd, dev
for (M, N, K, ta, tb) in [(8192, 8192, 8192, 0, 0)]:
    a = rnd(K, M) if ta else rnd(M, K)
    b = rnd(K, N) if tb else rnd(N, K)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    for _ in range(3):
        ops.gemm(a, b, M, N, K, out=out, trans_a=bool(ta), trans_b=bool(tb))
torch.cuda.synchronize()
