"""Decoder attention at the config-B shape (32 x 32 heads x 160 x 64, causal, probability dropout 0.1, the packed qkv layout of
models/modeling_distributed_gpt3.py:895-902): forward / backward time per launch, 50 back-to-back launches between events.
MPV_ATTN_PAIR is read once per process: run it once per mode.
  for m in 0 1; do MPV_ATTN_PAIR=$m python tools/attn_pair_ab.py; done"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import youku_mplug_amd
from youku_mplug_amd import ops

dev = torch.device("cuda:0")
B, S, H, hn = int(os.environ.get("AB_B", 32)), int(os.environ.get("AB_S", 160)), 32, 64
Hh = H * hn
torch.manual_seed(0)
qkv = (torch.randn(B, S, H, 3 * hn, device=dev) * 0.5).to(torch.bfloat16)
do = torch.randn(B, S, Hh, device=dev).to(torch.bfloat16)
o = torch.empty(B, S, Hh, dtype=torch.bfloat16, device=dev)
dqkv = torch.empty_like(qkv)
st = (S * 3 * Hh, 3 * hn, 3 * Hh)
lay = ops.AttnLayout(st, st, st, (S * Hh, hn, Hh))
q, k, v = qkv[..., :hn], qkv[..., hn:2 * hn], qkv[..., 2 * hn:]
kw = dict(causal=True, scale=hn ** -0.5, dropout_p=0.1, seed=5, offset=9)


def fwd():
    return ops.attn_fwd(q, k, v, o, lay, B, H, S, S, hn, **kw)


lse = fwd()


def bwd():
    ops.attn_bwd(q, k, v, o, lse, do, dqkv[..., :hn], dqkv[..., hn:2 * hn], dqkv[..., 2 * hn:], lay, B, H, S, S, hn, **kw)


def t(fn, n=50, rep=5):
    best = 1e9
    for _ in range(rep):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            fn()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / n * 1e3)
    return best


if os.environ.get("AB_VIT"):      # ViT-B/16 spatial attention at config B: 256 images x 8 heads x 197 x 96, pre-scaled q, no dropout
    B, S, H, hn = 256, 197, 8, 96
    Hh = H * hn
    qkv = (torch.randn(B, S, 3, H, hn, device=dev) * 0.5).to(torch.bfloat16)
    do = torch.randn(B, S, Hh, device=dev).to(torch.bfloat16)
    o = torch.empty(B, S, Hh, dtype=torch.bfloat16, device=dev)
    dqkv = torch.empty_like(qkv)
    st = (S * 3 * Hh, hn, 3 * Hh)
    lay = ops.AttnLayout(st, st, st, (S * Hh, hn, Hh))
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    dq_, dk_, dv_ = dqkv[:, :, 0], dqkv[:, :, 1], dqkv[:, :, 2]
    kw = dict(causal=False, scale=hn ** -0.5, scale_q_bf16=True)

    def fwd():
        return ops.attn_fwd(q, k, v, o, lay, B, H, S, S, hn, **kw)

    lse = fwd()

    def bwd():
        ops.attn_bwd(q, k, v, o, lse, do, dq_, dk_, dv_, lay, B, H, S, S, hn, **kw)

for _ in range(3):
    fwd(); bwd()
print(f"MPV_ATTN_PAIR={os.environ.get('MPV_ATTN_PAIR', '(default)')} B={B} S={S}: fwd {t(fwd):.1f} us  bwd {t(bwd):.1f} us   "
      f"checksum o {o.float().abs().mean().item():.6f} dqkv {dqkv.float().abs().mean().item():.6f}")
