"""EVA-ViT-g image encoder on the gfx950 kernels -- drop-in for models/eva_vit.py:245-436 as built by
create_eva_vit_g (:413-427: patch 14, width 1408, 40 blocks, 16 heads of 88, MLP 6144, q/v bias, learned absolute
positions, no relative-position bias, no LayerScale, final LayerNorm) and used by DistributedGPT3_Pretrain_Image
(models/distributed_gpt3.py:256-261).  Parameter names/shapes are the reference's.

Same execution model as vision.TimeSformer: modules hold parameters, forward_features/backward_features launch the
C-ABI kernels over one [B*(1+N), D] token stream and write parameter gradients in place.  Each block is
    x = x + proj(attn(norm1(x)));  x = x + fc2(gelu(fc1(norm2(x))))                      (models/eva_vit.py:172-174)
with q*scale rounded to bf16 before QK^T (:134-135).  (The reference also rounds the scores to bf16 before its
softmax; the fused kernel keeps them in fp32 -- closer to the fp32 oracle than the reference's own bf16 run.)
drop_path is 0 in the shipped visual configs (`visual_cfg.get('drop_path', False)`) and is not implemented.
"""
from __future__ import annotations

import math
from typing import List

import torch
from torch import nn

from . import ops
from .ops import ACT_GELU_ERF
from .vision import Attention, LayerNormWithForceFP32, Mlp, PatchEmbed, _param, _qkv_bias, grad_of


class EvaBlock(nn.Module):
    def __init__(self, dim, heads, hidden, eps, std, device=None):
        super().__init__()
        self.norm1 = LayerNormWithForceFP32(dim, eps, device)
        self.attn = Attention(dim, heads, std, device)
        self.norm2 = LayerNormWithForceFP32(dim, eps, device)
        self.mlp = Mlp(dim, hidden, std, device)


class EvaVisionTransformer(nn.Module):
    def __init__(self, img_size=224, patch_size=14, in_chans=3, embed_dim=1408, depth=40, num_heads=16, mlp_ratio=4.3637,
                 eps=1e-6, drop_path_rate=0.0, device=None, **_):
        super().__init__()
        if drop_path_rate:
            raise NotImplementedError("stochastic depth (drop_path) is 0 in the shipped visual configs")
        D = embed_dim
        hd = D // num_heads
        assert hd % 8 == 0 and hd <= 96, "fused attention kernels take head_dim = multiple of 8, <= 96"
        self.embed_dim = self.num_features = D
        self.num_heads, self.depth, self.image_size = num_heads, depth, img_size
        self.patch_embed = PatchEmbed(img_size, patch_size, in_chans, D, bias=True, std=0.02, device=device)
        N = self.patch_embed.num_patches
        self.cls_token = _param(1, 1, D, std=0.02, device=device)
        self.pos_embed = _param(1, N + 1, D, std=0.02, device=device)
        self.blocks = nn.ModuleList([EvaBlock(D, num_heads, int(D * mlp_ratio), eps, 0.02, device) for _ in range(depth)])
        self.norm = LayerNormWithForceFP32(D, eps, device)
        with torch.no_grad():                                            # fix_init_weight (:307-313)
            for i, blk in enumerate(self.blocks):
                blk.attn.proj.weight.div_(math.sqrt(2.0 * (i + 1)))
                blk.mlp.fc2.weight.div_(math.sqrt(2.0 * (i + 1)))
        self._no_temporal = torch.zeros(1, 1, D, dtype=torch.bfloat16, device=device)     # the assemble kernel's frame term
        self.on_block_grads_ready = None

    def no_weight_decay(self):
        return {"pos_embed", "cls_token"}

    # ------------------------------------------------------------------ forward
    def forward_features(self, image: torch.Tensor, tape: dict):
        """image [B,3,H,W] bf16 -> [B*(1+N), D] (cls first, :338-350), final LayerNorm applied."""
        B, Cc, H, W = image.shape
        D, P, N = self.embed_dim, self.patch_embed.patch_size[0], self.patch_embed.num_patches
        N1, heads, hd = N + 1, self.num_heads, self.embed_dim // self.num_heads
        R, Rt = B * N1, B * N
        Kc = Cc * P * P
        Kp = (Kc + 7) // 8 * 8
        cols = ops.im2col_patches(image.contiguous().view(B, Cc, 1, H, W), B, Cc, 1, H, W, P, Kp)
        wpe = self.patch_embed.proj.weight.detach().view(D, Kc)
        if Kp != Kc:
            wpe = torch.nn.functional.pad(wpe, (0, Kp - Kc))
        patch = ops.gemm(cols, wpe, Rt, D, Kp, bias=self.patch_embed.proj.bias)
        x = ops.vit_embed_assemble_fwd(patch, self.cls_token, self.pos_embed, self._no_temporal, B, 1, N, D)
        tape.update(B=B, N=N, cols=cols, Kp=Kp, Kc=Kc)
        st3 = (N1 * 3 * D, hd, 3 * D)
        lay = ops.AttnLayout(st3, st3, st3, (N1 * D, hd, D))
        blocks: List[dict] = []
        for blk in self.blocks:
            s = {}
            l1, s["m1"], s["r1"] = ops.layernorm_fwd(x, blk.norm1.weight, blk.norm1.bias, blk.norm1.eps, R, D)
            qkv = ops.gemm(l1, blk.attn.qkv.weight, R, 3 * D, D, bias=_qkv_bias(blk.attn))                 # :127-131
            a = torch.empty((R, D), dtype=torch.bfloat16, device=x.device)
            lse = ops.attn_fwd(qkv, qkv[:, D:], qkv[:, 2 * D:], a, lay, B, heads, N1, N1, hd, scale=blk.attn.scale,
                               scale_q_bf16=True)
            y = ops.gemm(a, blk.attn.proj.weight, R, D, D, bias=blk.attn.proj.bias, residual=x)            # :172
            l2, s["m2"], s["r2"] = ops.layernorm_fwd(y, blk.norm2.weight, blk.norm2.bias, blk.norm2.eps, R, D)
            hid = blk.mlp.fc1.out_features
            z = torch.empty((R, hid), dtype=torch.bfloat16, device=x.device)
            h1 = ops.gemm(l2, blk.mlp.fc1.weight, R, hid, D, bias=blk.mlp.fc1.bias, act=ACT_GELU_ERF, preact_out=z, preact_deriv=True)
            out = ops.gemm(h1, blk.mlp.fc2.weight, R, D, hid, bias=blk.mlp.fc2.bias, residual=y)           # :173
            s.update(x=x, l1=l1, qkv=qkv, a=a, lse=lse, y=y, l2=l2, z=z, h1=h1)
            blocks.append(s)
            x = out
        emb, mf, rf = ops.layernorm_fwd(x, self.norm.weight, self.norm.bias, self.norm.eps, R, D)          # :350
        tape.update(blocks=blocks, x_last=x, final_stats=(mf, rf), lay=lay)
        return emb

    # ------------------------------------------------------------------ backward
    def backward_features(self, demb: torch.Tensor, tape: dict):
        B, N = tape["B"], tape["N"]
        D, heads, hd = self.embed_dim, self.num_heads, self.embed_dim // self.num_heads
        N1, R, Rt = N + 1, B * (N + 1), B * N
        lay = tape["lay"]
        mf, rf = tape["final_stats"]
        dx = ops.layernorm_bwd(demb, tape["x_last"], self.norm.weight, mf, rf, R, D, dgamma=grad_of(self.norm.weight),
                               dbeta=grad_of(self.norm.bias))
        for bi in range(len(self.blocks) - 1, -1, -1):
            blk, s = self.blocks[bi], tape["blocks"][bi]
            hid = blk.mlp.fc1.out_features
            ops.gemm(dx, s["h1"], D, hid, R, trans_a=True, trans_b=True, out=grad_of(blk.mlp.fc2.weight), colsum_out=grad_of(blk.mlp.fc2.bias))
            dz = ops.gemm(dx, blk.mlp.fc2.weight, R, hid, D, trans_b=True, act_bwd_z=s["z"], act_bwd=ACT_GELU_ERF, z_is_deriv=True)
            ops.gemm(dz, s["l2"], hid, D, R, trans_a=True, trans_b=True, out=grad_of(blk.mlp.fc1.weight), colsum_out=grad_of(blk.mlp.fc1.bias))
            dl2 = ops.gemm(dz, blk.mlp.fc1.weight, R, D, hid, trans_b=True)
            dy = ops.layernorm_bwd(dl2, s["y"], blk.norm2.weight, s["m2"], s["r2"], R, D, dres=dx,
                                   dgamma=grad_of(blk.norm2.weight), dbeta=grad_of(blk.norm2.bias))
            ops.gemm(dy, s["a"], D, D, R, trans_a=True, trans_b=True, out=grad_of(blk.attn.proj.weight), colsum_out=grad_of(blk.attn.proj.bias))
            da = ops.gemm(dy, blk.attn.proj.weight, R, D, D, trans_b=True)
            qkv = s["qkv"]
            dqkv = torch.empty_like(qkv)
            ops.attn_bwd(qkv, qkv[:, D:], qkv[:, 2 * D:], s["a"], s["lse"], da, dqkv, dqkv[:, D:], dqkv[:, 2 * D:], lay,
                         B, heads, N1, N1, hd, scale=blk.attn.scale, scale_q_bf16=True)
            bsum = torch.empty(3 * D, dtype=torch.bfloat16, device=dqkv.device)
            ops.gemm(dqkv, s["l1"], 3 * D, D, R, trans_a=True, trans_b=True, out=grad_of(blk.attn.qkv.weight), colsum_out=bsum)
            grad_of(blk.attn.q_bias).copy_(bsum[:D])
            grad_of(blk.attn.v_bias).copy_(bsum[2 * D:])
            dl1 = ops.gemm(dqkv, blk.attn.qkv.weight, R, D, 3 * D, trans_b=True)
            dx = ops.layernorm_bwd(dl1, s["x"], blk.norm1.weight, s["m1"], s["r1"], R, D, dres=dy,
                                   dgamma=grad_of(blk.norm1.weight), dbeta=grad_of(blk.norm1.bias))
            tape["blocks"][bi] = None
            if self.on_block_grads_ready is not None:
                self.on_block_grads_ready(bi)
        dpatch = torch.empty((Rt, D), dtype=torch.bfloat16, device=dx.device)
        dtemporal = torch.zeros(1, 1, D, dtype=torch.bfloat16, device=dx.device)
        ops.vit_embed_assemble_bwd(dx, dpatch, grad_of(self.cls_token), grad_of(self.pos_embed), dtemporal, B, 1, N, D)
        Kp, Kc = tape["Kp"], tape["Kc"]
        gw = grad_of(self.patch_embed.proj.weight)
        if Kp == Kc:
            ops.gemm(dpatch, tape["cols"], D, Kp, Rt, trans_a=True, trans_b=True, out=gw)
        else:
            tmp = ops.gemm(dpatch, tape["cols"], D, Kp, Rt, trans_a=True, trans_b=True)
            gw.view(D, Kc).copy_(tmp[:, :Kc])
        ops.colsum(dpatch, Rt, D, out=grad_of(self.patch_embed.proj.bias))
        if self.on_block_grads_ready is not None:
            self.on_block_grads_ready(-1)


def create_eva_vit_g(img_size=224, drop_path_rate=0.0, eps=1e-6, device=None, **overrides):
    """models/eva_vit.py:413-427.  `overrides` (embed_dim, depth, num_heads, mlp_ratio, patch_size) exist for
    shape-reduced tests; the defaults are EVA-ViT-g."""
    kw = dict(img_size=img_size, patch_size=14, embed_dim=1408, depth=40, num_heads=1408 // 88, mlp_ratio=4.3637,
              drop_path_rate=drop_path_rate, eps=eps, device=device)
    kw.update(overrides)
    return EvaVisionTransformer(**kw)
