"""Weight-streaming rate of the small-M GEMM (decode regime) per decoder shape."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, youku_mplug_amd
from youku_mplug_amd import ops
from tools.bench_kernels import timeit, rnd, dev
for M, N, K in [(5, 6144, 2048), (5, 2048, 2048), (5, 8192, 2048), (5, 2048, 8192), (5, 51200, 2048), (1, 8192, 2048), (16, 8192, 2048)]:
    ws = [rnd(N, K) for _ in range(12)]          # rotate weights so they are not L2/MALL-resident
    a = rnd(M, K)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    i = [0]
    def f():
        i[0] = (i[0] + 1) % len(ws)
        ops.gemm(a, ws[i[0]], M, N, K, out=out)
    t = timeit(f)
    print(f"M={M:2d} N={N:5d} K={K:5d}: {t*1e6:7.1f} us  {N*K*2/t/1e12:5.2f} TB/s")
