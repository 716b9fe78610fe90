"""Weight-gradient products of the ViT block (dW[M,N] = dY[K,M]^T X[K,N], K = 50432 rows): time of GEMM + split-K reduce
against the split count, the XCD walk (gm) and the row pitch of dY.  Usage (GPU box): python tools/probe/wgrad_splits.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import youku_mplug_amd
from youku_mplug_amd import ops

dev = torch.device("cuda:0")


def t_us(fn, iters=10, rounds=3):
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
    K = 50432
    for (name, M, N) in [("qkv", 2304, 768), ("fc1", 3072, 768), ("fc2", 768, 3072), ("proj", 768, 768)]:
        dy = (torch.rand(K, M, device=dev) * 2 - 1).bfloat16()
        x = (torch.rand(K, N, device=dev) * 2 - 1).bfloat16()
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        fl = 2.0 * M * N * K
        base = t_us(lambda: ops.gemm(dy, x, M, N, K, trans_a=True, trans_b=True, out=out))
        ref = out.clone()
        row = [f"default {base:6.1f} us ({fl / base / 1e6:6.0f} TF/s)"]
        tiles = ((M + 255) // 256) * ((N + 255) // 256)
        for s in sorted({max(1, 128 // tiles), max(1, 192 // tiles), max(1, 256 // tiles), max(1, 384 // tiles), max(1, 512 // tiles), max(1, 768 // tiles)}):
            t = t_us(lambda: ops.gemm(dy, x, M, N, K, trans_a=True, trans_b=True, out=out, split_hint=s))
            assert (out.float() - ref.float()).abs().max().item() <= 2e-2 * ref.float().abs().max().item()
            row.append(f"s={s}: {t:6.1f}")
        for gm in (1, 2, 3, 9, 12):
            t = t_us(lambda: ops.gemm(dy, x, M, N, K, trans_a=True, trans_b=True, out=out, gm_hint=gm))
            row.append(f"gm={gm}: {t:6.1f}")
        if M % 256 == 0 and M < 4096:          # same product with dY rows at a 4096-element pitch
            big = torch.empty(K, 4096, dtype=torch.bfloat16, device=dev)
            big[:, :M] = dy
            t = t_us(lambda: ops.gemm(big, x, M, N, K, trans_a=True, trans_b=True, out=out, lda=4096))
            row.append(f"lda=4096: {t:6.1f}")
            big2 = torch.empty(K, M + 64, dtype=torch.bfloat16, device=dev)
            big2[:, :M] = dy
            t = t_us(lambda: ops.gemm(big2, x, M, N, K, trans_a=True, trans_b=True, out=out, lda=M + 64))
            row.append(f"lda=M+64: {t:6.1f}")
        print(f"{name:5s} M={M} N={N} tiles={tiles}: " + "  ".join(row), flush=True)


if __name__ == "__main__":
    main()
