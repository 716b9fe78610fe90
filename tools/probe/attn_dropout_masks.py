"""Recover the dropout masks the attention forward, dQ and dK/dV kernels draw (q = k = 0 -> uniform probabilities,
one-hot V / dO) and compare them element by element."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, youku_mplug_amd
from youku_mplug_amd import ops
dev = torch.device("cuda:0")
B, H, S, hd = 1, 2, 64, 64
for causal in (False, True):
    q = torch.zeros(B, S, H, hd, dtype=torch.bfloat16, device=dev)
    k = torch.zeros_like(q)
    v = torch.zeros_like(q)
    eye = torch.eye(S, hd, dtype=torch.bfloat16, device=dev)
    v[:] = eye.view(1, S, 1, hd)
    o = torch.empty_like(q)
    lay = ops.AttnLayout((S * H * hd, hd, H * hd),) * 4 if False else ops.AttnLayout((S * H * hd, hd, H * hd), (S * H * hd, hd, H * hd), (S * H * hd, hd, H * hd), (S * H * hd, hd, H * hd))
    kw = dict(causal=causal, scale=hd ** -0.5, dropout_p=0.25, seed=5, offset=9)
    lse = ops.attn_fwd(q, k, v, o, lay, B, H, S, S, hd, **kw)
    fwd_mask = (o.float().permute(0, 2, 1, 3) > 0)                  # [B,H,q,key]
    do = torch.zeros_like(q)
    do[:] = eye.view(1, S, 1, hd)                                   # dO[q] = e_q
    dq, dk, dv = torch.empty_like(q), torch.empty_like(q), torch.empty_like(q)
    ops.attn_bwd(q, k, v, o, lse, do, dq, dk, dv, lay, B, H, S, S, hd, **kw)
    bwd_mask = (dv.float().permute(0, 2, 1, 3) > 0).transpose(-1, -2)   # dV[key][q] -> [q,key]
    vis = torch.ones(S, S, dtype=torch.bool, device=dev).tril() if causal else torch.ones(S, S, dtype=torch.bool, device=dev)
    diff = (fwd_mask != bwd_mask) & vis
    print("causal", causal, "keep rate fwd %.4f bwd %.4f mismatches %d of %d" % (
        fwd_mask[..., vis].float().mean().item(), bwd_mask[..., vis].float().mean().item(), int(diff.sum()), int(vis.sum()) * B * H))
    if diff.any():
        idx = diff.nonzero()[:12]
        print(idx.tolist())
