"""Would tiles of different heights running side by side (their store-bound epilogues no longer coinciding on every CU)
beat the same launches run one after the other?  Two whole-round launches of the K = 768 ViT shapes, 256-row tiles and
192-row tiles, (a) back to back on one stream, (b) concurrently on two streams.  Usage: python tools/probe/mixed_tiles.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import youku_mplug_amd
from youku_mplug_amd import ops

dev = torch.device("cuda:0")
r = lambda *s: (torch.rand(*s, device=dev) * 2 - 1).bfloat16()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def t_us(fn, iters=10, rounds=4):
    best = 1e9
    for _ in range(rounds):
        fn(); fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            fn()
        e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / iters * 1e3)
    return best


for (N, K, rounds_each) in [(768, 768, 2), (2304, 768, 3), (768, 3072, 2), (3072, 768, 3)]:
    tn = N // 256
    mt = rounds_each * 256 // tn                      # m-tiles per launch: whole rounds of 256 workgroups
    Ma, Mb = mt * 256, mt * 192
    w = r(N, K)
    a1, a2 = r(Ma, K), r(Mb, K)
    o1 = torch.empty(Ma, N, dtype=torch.bfloat16, device=dev)
    o2 = torch.empty(Mb, N, dtype=torch.bfloat16, device=dev)
    r1, r2 = r(Ma, N), r(Mb, N)
    for res in (False, True):
        kw1 = dict(residual=r1) if res else {}
        kw2 = dict(residual=r2) if res else {}

        def seq():
            ops.gemm(a1, w, Ma, N, K, out=o1, tile_hint=256, **kw1)
            ops.gemm(a2, w, Mb, N, K, out=o2, tile_hint=192, **kw2)

        def par():
            cur = torch.cuda.current_stream()
            s1.wait_stream(cur); s2.wait_stream(cur)
            with torch.cuda.stream(s1):
                ops.gemm(a1, w, Ma, N, K, out=o1, tile_hint=256, **kw1)
            with torch.cuda.stream(s2):
                ops.gemm(a2, w, Mb, N, K, out=o2, tile_hint=192, **kw2)
            cur.wait_stream(s1); cur.wait_stream(s2)

        ta = t_us(lambda: ops.gemm(a1, w, Ma, N, K, out=o1, tile_hint=256, **kw1))
        tb = t_us(lambda: ops.gemm(a2, w, Mb, N, K, out=o2, tile_hint=192, **kw2))
        print(f"N={N} K={K} {'+res' if res else 'plain'}: 256-row x{mt * tn} wgs {ta:6.1f} us, 192-row x{mt * tn} wgs {tb:6.1f} us, "
              f"back to back {t_us(seq):6.1f}, two streams {t_us(par):6.1f}", flush=True)
