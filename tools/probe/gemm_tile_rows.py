"""256- / 192- / 160-row tiles of the eight-phase kernel on the shapes whose 256-row tiling underfills the chip."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, youku_mplug_amd
from youku_mplug_amd import ops
from tools.bench_kernels import rnd, timeit
for (M, N, K) in [(5120, 2048, 2048), (5120, 2048, 6144), (5120, 2048, 8192), (5120, 6144, 2048), (5120, 8192, 2048), (50432, 768, 768),
                  (50432, 768, 3072), (50432, 3072, 768)]:
    a, w = rnd(M, K), rnd(N, K)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=a.device)
    line = f"{M:6d} {N:6d} {K:6d} |"
    for hint in (256, 192, 160, 0):
        t = timeit(lambda: ops.gemm(a, w, M, N, K, out=out, tile_hint=hint), iters=30, warm=5)
        line += f"  {hint or 'auto':>4}: {t * 1e6:7.1f} us {2.0 * M * N * K / t / 1e12:7.1f} TF/s"
    print(line, flush=True)
