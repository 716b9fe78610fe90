import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, youku_mplug_amd
from youku_mplug_amd import ops
from tools.bench_kernels import rnd, dev, timeit
M = N = K = 8192
a, b = rnd(M, K), rnd(N, K)
out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
t = timeit(lambda: ops.gemm(a, b, M, N, K, out=out))
print(f"ablate={os.environ.get('MPV_GEMM_ABLATE','0')} pp={os.environ.get('MPV_GEMM_PP','0')}: {t*1e6:8.1f} us {2*M*N*K/t/1e12:7.1f} TF/s")
