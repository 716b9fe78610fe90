"""A/B of the two tile kernels (tile_hint=128 / 256) and the vendor library (torch.matmul, measurement aid only: the
product never calls it) on the GEMM shapes of one config-B step, random bf16 data, interleaved rounds in one process.
Usage (GPU box): python tools/gemm_ab.py [rounds]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import youku_mplug_amd
from youku_mplug_amd import ops

dev = torch.device("cuda:0")
SHAPES = [  # name, M, N, K, ta, tb, launches per step
    ("vit qkv fwd", 50432, 2304, 768, 0, 0, 24), ("vit proj fwd", 50432, 768, 768, 0, 0, 36), ("vit fc1 fwd", 50432, 3072, 768, 0, 0, 12),
    ("vit fc2 fwd", 50432, 768, 3072, 0, 0, 12), ("vit qkv dgrad", 50432, 768, 2304, 0, 1, 24), ("vit proj dgrad", 50432, 768, 768, 0, 1, 36),
    ("vit fc1 dgrad", 50432, 768, 3072, 0, 1, 12), ("vit fc2 dgrad", 50432, 3072, 768, 0, 1, 12),
    ("vit qkv wgrad", 2304, 768, 50432, 1, 1, 24), ("vit proj wgrad", 768, 768, 50432, 1, 1, 36), ("vit fc1 wgrad", 3072, 768, 50432, 1, 1, 12),
    ("vit fc2 wgrad", 768, 3072, 50432, 1, 1, 12),
    ("gpt qkv fwd", 5120, 6144, 2048, 0, 0, 24), ("gpt dense fwd", 5120, 2048, 2048, 0, 0, 24), ("gpt h4h fwd", 5120, 8192, 2048, 0, 0, 24),
    ("gpt 4hh fwd", 5120, 2048, 8192, 0, 0, 24), ("gpt qkv dgradT", 5120, 2048, 6144, 0, 0, 24), ("gpt 4hh dgradT", 5120, 8192, 2048, 0, 0, 24),
    ("lm head fwd", 5120, 51200, 2048, 0, 0, 1), ("lm head dgrad", 5120, 2048, 51200, 0, 1, 1),
    ("square 4096", 4096, 4096, 4096, 0, 0, 0), ("square 8192", 8192, 8192, 8192, 0, 0, 0),
]


def rnd(*shape):
    return (torch.rand(*shape, device=dev) * 2 - 1).bfloat16()     # uniform [-1, 1): the guide's reference fill


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    only = sys.argv[2] if len(sys.argv) > 2 else ""
    tot = {"128": 0.0, "256": 0.0, "lib": 0.0, "best": 0.0}
    totfl = 0.0
    print(f"{'shape':16s} {'M':>6s} {'N':>6s} {'K':>6s} | {'mpv128':>8s} {'mpv256':>8s} {'vendor':>8s} TF/s | us 128/256/lib", flush=True)
    for name, M, N, K, ta, tb, cnt in SHAPES:
        if only and only not in name:
            continue
        a = rnd(K, M) if ta else rnd(M, K)
        b = rnd(K, N) if tb else rnd(N, K)
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        fns = {
            "128": lambda: ops.gemm(a, b, M, N, K, out=out, trans_a=bool(ta), trans_b=bool(tb), tile_hint=128),
            "256": lambda: ops.gemm(a, b, M, N, K, out=out, trans_a=bool(ta), trans_b=bool(tb), tile_hint=256),
        }
        am = a.t() if ta else a
        bm = b if tb else b.t()
        fns["lib"] = lambda: torch.matmul(am, bm, out=out)
        best = {k: 1e9 for k in fns}
        iters = 10 if M * N * K < 3e11 else 4
        for r in range(rounds):
            for k, fn in fns.items():
                fn(); fn()
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(iters):
                    fn()
                e.record()
                torch.cuda.synchronize()
                best[k] = min(best[k], s.elapsed_time(e) / iters * 1e-3)
        # never quote a time for a wrong result: both tile kernels against the vendor output, every element
        ref = torch.matmul(am, bm).float()
        scale = ref.abs().max().item()
        errs = []
        for k in ("128", "256"):
            fns[k]()
            errs.append(((out.float() - ref).abs().max().item()) / scale)
        del ref
        assert max(errs) < 1e-2, f"{name}: wrong result, rel err 128/256 = {errs}"
        fl = 2.0 * M * N * K
        print(f"{name:16s} {M:6d} {N:6d} {K:6d} | {fl/best['128']/1e12:8.1f} {fl/best['256']/1e12:8.1f} {fl/best['lib']/1e12:8.1f}      | "
              f"{best['128']*1e6:8.1f} {best['256']*1e6:8.1f} {best['lib']*1e6:8.1f}", flush=True)
        for k in ("128", "256", "lib"):
            tot[k] += best[k] * cnt
        tot["best"] += min(best["128"], best["256"]) * cnt
        totfl += fl * cnt
    print(f"step-weighted GEMM time (ms): mpv128 {tot['128']*1e3:.2f}  mpv256 {tot['256']*1e3:.2f}  best-of {tot['best']*1e3:.2f}  vendor {tot['lib']*1e3:.2f};"
          f"  {totfl/1e12:.1f} TFLOP -> best-of {totfl/tot['best']/1e12:.0f} TF/s, vendor {totfl/tot['lib']/1e12:.0f} TF/s")


if __name__ == "__main__":
    main()
