"""Is the store-bound epilogue of the 256x256 GEMM a per-CU limit or a chip-wide one?  One launch of W workgroups (M = 256 W,
N = 256) at K = 64 (prologue + one K-tile + epilogue only) and K = 768, W = 8 ... 256 (256 = every CU storing at once).
Also the launch-to-launch gap of back-to-back dependent launches on one stream (what an extra row band costs).
Usage (GPU box): python tools/probe/epi_contention.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import youku_mplug_amd
from youku_mplug_amd import ops

dev = torch.device("cuda:0")


def t_us(fn, iters=50, rounds=5):
    best = 1e9
    for _ in range(rounds):
        fn(); fn()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            fn()
        e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / iters * 1e3)
    return best


def main():
    N = 256
    print("W = workgroups (one 256x256 tile each); us per launch (back-to-back launches, includes the launch gap)")
    for K in (64, 768, 3072):
        row = []
        for W in (8, 32, 64, 128, 256, 512):
            M = 256 * W
            a = (torch.rand(M, K, device=dev) * 2 - 1).bfloat16()
            b = (torch.rand(N, K, device=dev) * 2 - 1).bfloat16()
            out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
            res = (torch.rand(M, N, device=dev) * 2 - 1).bfloat16()
            plain = t_us(lambda: ops.gemm(a, b, M, N, K, out=out, tile_hint=256))
            withres = t_us(lambda: ops.gemm(a, b, M, N, K, out=out, residual=res, tile_hint=256))
            row.append(f"W={W}: {plain:6.1f} / +res {withres:6.1f}")
        print(f"K={K:5d}  " + "   ".join(row), flush=True)
    # launch gap: four full rounds of tiles as 1, 2 or 4 dependent launches of whole rounds
    K = 768
    for W in (1024,):
        M = 256 * W
        a = (torch.rand(M, K, device=dev) * 2 - 1).bfloat16()
        b = (torch.rand(N, K, device=dev) * 2 - 1).bfloat16()
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        one = t_us(lambda: ops.gemm(a, b, M, N, K, out=out, tile_hint=256))
        def halves(parts):
            step = M // parts
            for i in range(parts):
                ops.gemm(a[i * step:(i + 1) * step], b, step, N, K, out=out[i * step:(i + 1) * step], tile_hint=256)
        two, four = t_us(lambda: halves(2)), t_us(lambda: halves(4))
        print(f"launch split of a {W}-tile four-round problem (K={K}): 1 launch {one:.1f} us, 2 launches {two:.1f}, 4 launches {four:.1f}")


if __name__ == "__main__":
    main()
