"""A/B of the 256x256 kernel pinned to one launch of 256-row tiles (tile_hint=256), the library's own choice (tile_hint=0:
row bands of 256- / 192- / 160-row tiles, split-K where it applies) and the vendor library (torch.matmul, measurement aid only: the
product never calls it) on the GEMM shapes of one config-B step, random bf16 data, interleaved rounds in one process.
MPV_AB_128=1 adds the round-1 128x128 kernel as a fourth arm.
Usage (GPU box): python tools/gemm_ab.py [rounds] [name filter]"""
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
    ("lm head fwd", 1024, 51200, 2048, 0, 0, 1), ("lm head dgrad", 1024, 2048, 51200, 0, 1, 1),      # loss window rows only
    ("square 4096", 4096, 4096, 4096, 0, 0, 0), ("square 8192", 8192, 8192, 8192, 0, 0, 0),
]


def rnd(*shape):
    return (torch.rand(*shape, device=dev) * 2 - 1).bfloat16()     # uniform [-1, 1): the guide's reference fill


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    only = sys.argv[2] if len(sys.argv) > 2 else ""
    arms = (["128"] if os.environ.get("MPV_AB_128") else []) + ["256", "auto", "lib"]
    tot = {k: 0.0 for k in arms}
    totfl = 0.0
    wins = n_shapes = 0
    print(f"{'shape':16s} {'M':>6s} {'N':>6s} {'K':>6s} | " + " ".join(f"{k:>8s}" for k in arms) + " TF/s | us " + "/".join(arms), flush=True)
    for name, M, N, K, ta, tb, cnt in SHAPES:
        if only and only not in name:
            continue
        a = rnd(K, M) if ta else rnd(M, K)
        b = rnd(K, N) if tb else rnd(N, K)
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        am = a.t() if ta else a
        bm = b if tb else b.t()
        hint = {"128": 128, "256": 256, "auto": 0}
        fns = {k: (lambda h=hint[k]: ops.gemm(a, b, M, N, K, out=out, trans_a=bool(ta), trans_b=bool(tb), tile_hint=h)) for k in arms if k != "lib"}
        fns["lib"] = lambda: torch.matmul(am, bm, out=out)
        best = {k: 1e9 for k in arms}
        iters = 10 if M * N * K < 3e11 else 4
        for r in range(rounds):
            for k in arms:
                fn = fns[k]
                fn(); fn()
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(iters):
                    fn()
                e.record()
                torch.cuda.synchronize()
                best[k] = min(best[k], s.elapsed_time(e) / iters * 1e-3)
        # never quote a time for a wrong result: every mpv arm against the vendor output, every element
        ref = torch.matmul(am, bm).float()
        scale = ref.abs().max().item()
        for k in arms:
            if k != "lib":
                fns[k]()
                err = (out.float() - ref).abs().max().item() / scale
                assert err < 1e-2, f"{name}: wrong result from arm {k}, rel err {err}"
        del ref
        fl = 2.0 * M * N * K
        print(f"{name:16s} {M:6d} {N:6d} {K:6d} | " + " ".join(f"{fl/best[k]/1e12:8.1f}" for k in arms) + "      | " +
              " ".join(f"{best[k]*1e6:8.1f}" for k in arms), flush=True)
        for k in arms:
            tot[k] += best[k] * cnt
        totfl += fl * cnt
        if cnt:
            n_shapes += 1
            wins += best["auto"] <= best["lib"]
    print("step-weighted GEMM time (ms): " + "  ".join(f"{k} {tot[k]*1e3:.2f}" for k in arms) +
          f";  {totfl/1e12:.1f} TFLOP -> auto {totfl/tot['auto']/1e12:.0f} TF/s, vendor {totfl/tot['lib']/1e12:.0f} TF/s;  "
          f"auto >= vendor on {wins} of {n_shapes} step shapes")


if __name__ == "__main__":
    main()
