import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, youku_mplug_amd
from youku_mplug_amd import ops
from tools.bench_kernels import timeit, rnd, dev
print("K sweep, M=50432 N=2304 (NT)")
for K in (128, 256, 512, 768, 1536, 3072, 6144):
    a, b = rnd(50432, K), rnd(2304, K)
    out = torch.empty(50432, 2304, dtype=torch.bfloat16, device=dev)
    t = timeit(lambda: ops.gemm(a, b, 50432, 2304, K, out=out))
    tiles = 394 * 18
    print(f"K={K:5d} {t*1e6:8.1f} us  {2*50432*2304*K/t/1e12:7.1f} TF/s   per-tile-slot {t*1e6/(tiles/512):6.2f} us  per-kstep {t*1e6/(tiles/512)/(K/64):5.2f} us")
print("out_f32 (no epilogue math) K=768")
a, b = rnd(50432, 768), rnd(2304, 768)
o32 = torch.empty(50432, 2304, dtype=torch.float32, device=dev)
t = timeit(lambda: ops.gemm(a, b, 50432, 2304, 768, out=o32, out_f32=True))
print(f"f32 out: {t*1e6:8.1f} us")
