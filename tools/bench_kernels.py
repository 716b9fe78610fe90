"""Micro-benchmarks of the C-ABI kernels on the shapes of config B.
Usage (GPU box): python tools/bench_kernels.py [gemm|attn|ln|all]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import youku_mplug_amd
from youku_mplug_amd import ops

dev = torch.device("cuda:0")


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def rnd(*shape):
    return (torch.randn(*shape, device=dev) * 0.5).bfloat16()


def bench_gemm():
    shapes = [  # name, M, N, K, ta, tb
        ("vit qkv fwd", 50432, 2304, 768, 0, 0), ("vit proj fwd", 50432, 768, 768, 0, 0),
        ("vit fc1 fwd", 50432, 3072, 768, 0, 0), ("vit fc2 fwd", 50432, 768, 3072, 0, 0),
        ("vit qkv dgrad", 50432, 768, 2304, 0, 1), ("vit fc1 dgrad", 50432, 768, 3072, 0, 1), ("vit fc2 dgrad", 50432, 3072, 768, 0, 1),
        ("vit qkv wgrad", 2304, 768, 50432, 1, 1), ("vit proj wgrad", 768, 768, 50432, 1, 1), ("vit fc1 wgrad", 3072, 768, 50432, 1, 1),
        ("vit fc2 wgrad", 768, 3072, 50432, 1, 1),
        ("gpt qkv fwd", 5120, 6144, 2048, 0, 0), ("gpt dense fwd", 5120, 2048, 2048, 0, 0), ("gpt h4h fwd", 5120, 8192, 2048, 0, 0),
        ("gpt 4hh fwd", 5120, 2048, 8192, 0, 0), ("gpt qkv dgrad", 5120, 2048, 6144, 0, 1), ("gpt h4h dgrad", 5120, 2048, 8192, 0, 1),
        ("gpt 4hh dgrad", 5120, 8192, 2048, 0, 1), ("lm head fwd", 5120, 51200, 2048, 0, 0), ("lm head dgrad", 5120, 2048, 51200, 0, 1),
        ("square 4096", 4096, 4096, 4096, 0, 0), ("square 8192", 8192, 8192, 8192, 0, 0),
    ]
    for name, M, N, K, ta, tb in shapes:
        a = rnd(K, M) if ta else rnd(M, K)
        b = rnd(K, N) if tb else rnd(N, K)
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        t = timeit(lambda: ops.gemm(a, b, M, N, K, out=out, trans_a=bool(ta), trans_b=bool(tb)))
        print(f"{name:16s} M={M:6d} N={N:6d} K={K:6d}  {t*1e6:9.1f} us  {2*M*N*K/t/1e12:8.1f} TF/s", flush=True)


def bench_attn():
    cases = [("gpt causal", 32, 32, 160, 160, 64, True, False), ("vit spatial", 256, 8, 197, 197, 96, False, True),
             ("pool cross", 32, 8, 128, 1570, 96, False, False)]
    for name, B, H, Sq, Sk, hd, causal, sqb in cases:
        q, k, v = rnd(B, Sq, H, hd), rnd(B, Sk, H, hd), rnd(B, Sk, H, hd)
        o, do = torch.empty_like(q), rnd(B, Sq, H, hd)
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        lay = ops.AttnLayout((Sq * H * hd, hd, H * hd), (Sk * H * hd, hd, H * hd), (Sk * H * hd, hd, H * hd), (Sq * H * hd, hd, H * hd))
        lse = ops.attn_fwd(q, k, v, o, lay, B, H, Sq, Sk, hd, causal=causal, scale=hd ** -0.5, scale_q_bf16=sqb)
        tf = timeit(lambda: ops.attn_fwd(q, k, v, o, lay, B, H, Sq, Sk, hd, causal=causal, scale=hd ** -0.5, scale_q_bf16=sqb))
        tb = timeit(lambda: ops.attn_bwd(q, k, v, o, lse, do, dq, dk, dv, lay, B, H, Sq, Sk, hd, causal=causal, scale=hd ** -0.5, scale_q_bf16=sqb))
        fl = 4 * B * H * Sq * Sk * hd * (0.5 if causal else 1)
        print(f"{name:12s} fwd {tf*1e6:8.1f} us ({fl/tf/1e12:6.1f} TF/s)  bwd {tb*1e6:8.1f} us ({2.5*fl/tb/1e12:6.1f} TF/s)", flush=True)
    B, T, N, heads, hd = 32, 8, 196, 8, 96
    N1, D = N + 1, heads * hd
    qkv = rnd(B * T * N1, 3 * D)
    out, dout, dqkv = torch.empty(B * T * N1, D, dtype=torch.bfloat16, device=dev), rnd(B * T * N1, D), torch.empty_like(qkv)
    tf = timeit(lambda: ops.temporal_attn_fwd(qkv, out, B, T * N1, N, 1, N1, T, heads, hd, hd ** -0.5))
    tb = timeit(lambda: ops.temporal_attn_bwd(qkv, dout, dqkv, B, T * N1, N, 1, N1, T, heads, hd, hd ** -0.5))
    by = B * T * N * D * 2
    print(f"temporal     fwd {tf*1e6:8.1f} us ({4*by/tf/1e9:7.0f} GB/s)  bwd {tb*1e6:8.1f} us ({8*by/tb/1e9:7.0f} GB/s)", flush=True)


def bench_ln():
    for rows, cols in ((50432, 768), (5120, 2048)):
        x, g, b, dy = rnd(rows, cols), rnd(cols), rnd(cols), rnd(rows, cols)
        y = torch.empty_like(x)
        _, mean, rstd = ops.layernorm_fwd(x, g, b, 1e-5, rows, cols, out=y)
        tf = timeit(lambda: ops.layernorm_fwd(x, g, b, 1e-5, rows, cols, out=y))
        dx, dg, db = torch.empty_like(x), torch.empty_like(g), torch.empty_like(b)
        tb = timeit(lambda: ops.layernorm_bwd(dy, x, g, mean, rstd, rows, cols, dx=dx, dgamma=dg, dbeta=db))
        by = rows * cols * 2
        print(f"LN {rows}x{cols}: fwd {tf*1e6:7.1f} us ({2*by/tf/1e9:6.0f} GB/s)  bwd {tb*1e6:7.1f} us ({3*by/tb/1e9:6.0f} GB/s)", flush=True)
        # the forms the step runs: ViT = residual gradient + dgamma/dbeta; GPT (frozen) = residual gradient + dropout-masked copy
        dres, dxd = rnd(rows, cols), torch.empty_like(x)
        t1 = timeit(lambda: ops.layernorm_bwd(dy, x, g, mean, rstd, rows, cols, dres=dres, dx=dx, dgamma=dg, dbeta=db))
        t2 = timeit(lambda: ops.layernorm_bwd(dy, x, g, mean, rstd, rows, cols, dres=dres, dx=dx, dx_drop=dxd, dropout_p=0.1, seed=3, offset=11))
        print(f"   bwd + dres + dparams {t1*1e6:7.1f} us ({4*by/t1/1e9:6.0f} GB/s)   bwd + dres + dropped copy {t2*1e6:7.1f} us ({5*by/t2/1e9:6.0f} GB/s)", flush=True)
    n = 130_000_000 // 256 * 256
    p16, master, m, v, g = (torch.zeros(n, dtype=torch.bfloat16, device=dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev),
                            torch.zeros(n, device=dev), rnd(n))
    ss = torch.zeros((), device=dev)
    t = timeit(lambda: ops.adamw_step(p16, master, m, v, g, 1e-4, 0.9, 0.999, 1e-6, 0.05, 1, 1.0, ss, 3.0), iters=5)
    print(f"AdamW {n/1e6:.0f}M: {t*1e6:8.1f} us ({n*28/t/1e9:6.0f} GB/s)")
    rows, V = 5120, 51200
    logits, labels, w = rnd(rows, V), torch.randint(0, V, (rows,), device=dev), torch.full((rows,), 1.0 / rows, device=dev)
    dl = torch.empty_like(logits)
    t = timeit(lambda: ops.cross_entropy(logits, labels, w, rows, V, dlogits=dl), iters=5)
    print(f"CE {rows}x{V}: {t*1e6:8.1f} us ({rows*V*2*2/t/1e9:6.0f} GB/s algorithmic r+w)")


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("gemm", "all"):
        bench_gemm()
    if what in ("attn", "all"):
        bench_attn()
    if what in ("ln", "all"):
        bench_ln()
