"""A/B of the pinned 256-row launch, the pinned 192- / 160-row launches and the library's own choice (row bands) on the epilogue-heavy GEMMs of the step (GELU + pre-activation copy, GELU-backward multiply,
dropout + residual), random data.  Usage (GPU box): python tools/gemm_ab_epi.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import youku_mplug_amd
from youku_mplug_amd import ops

dev = torch.device("cuda:0")


def rnd(*shape):
    return (torch.rand(*shape, device=dev) * 2 - 1).bfloat16()


def timeit(fn, iters=8):
    fn(); fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


CASES = [  # name, M, N, K, tb, kind
    ("vit fc1 fwd gelu+pre", 50432, 3072, 768, 0, "gelu_erf"), ("vit fc2 dgrad gelu'", 50432, 3072, 768, 1, "bwd_erf"),
    ("vit proj fwd +res", 50432, 768, 768, 0, "res"), ("vit fc2 fwd +res", 50432, 768, 3072, 0, "res"),
    ("gpt h4h fwd gelu+pre", 5120, 8192, 2048, 0, "gelu_tanh"), ("gpt 4hh dgradT gelu'", 5120, 8192, 2048, 0, "bwd_tanh"),
    ("gpt dense drop+res", 5120, 2048, 2048, 0, "dropres"), ("gpt 4hh fwd drop+res", 5120, 2048, 8192, 0, "dropres"),
    ("plain fc1 shape", 50432, 3072, 768, 0, "plain"), ("vit proj dgrad plain", 50432, 768, 768, 1, "plain"),
    ("vit qkv dgrad plain", 50432, 768, 2304, 1, "plain"), ("vit fc1 dgrad plain", 50432, 768, 3072, 1, "plain"),
]
for name, M, N, K, tb, kind in CASES:
    a = rnd(M, K)
    b = rnd(K, N) if tb else rnd(N, K)
    bias, res, z = rnd(N), rnd(M, N), rnd(M, N)
    out, pre = torch.empty(M, N, dtype=torch.bfloat16, device=dev), torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    kw = dict(out=out, trans_b=bool(tb))
    if kind == "gelu_erf":
        kw.update(bias=bias, act=ops.ACT_GELU_ERF, preact_out=pre)
    elif kind == "gelu_tanh":
        kw.update(bias=bias, act=ops.ACT_GELU_TANH, preact_out=pre)
    elif kind == "bwd_erf":
        kw.update(act_bwd_z=z, act_bwd=ops.ACT_GELU_ERF)
    elif kind == "bwd_tanh":
        kw.update(act_bwd_z=z, act_bwd=ops.ACT_GELU_TANH)
    elif kind == "res":
        kw.update(bias=bias, residual=res)
    elif kind == "dropres":
        kw.update(bias=bias, residual=res, dropout_p=0.1, seed=1, offset=7)
    t = {}
    arms = (256, 192, 160, 0)
    for r in range(4):
        for h in arms:
            t[h] = min(t.get(h, 1e9), timeit(lambda: ops.gemm(a, b, M, N, K, tile_hint=h, **kw)))
    fl = 2.0 * M * N * K
    print(f"{name:24s} {M:6d} {N:5d} {K:5d} | " + " | ".join(f"{'auto' if h == 0 else h}: {t[h]*1e6:7.1f} us {fl/t[h]/1e12:7.1f}" for h in arms), flush=True)
