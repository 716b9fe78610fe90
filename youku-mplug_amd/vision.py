"""TimeSformer video encoder + AttentionPool visual abstractor on the gfx950 kernels.

Mirrors the reference module tree / parameter names (state-dict drop-in):
  TimeSformer   models/vision_transformer.py:440-592   (Block :211-275, Attention :113-207,
                Mlp :93-110, PatchEmbed :377-398, LayerNormWithForceFP32 :43-71)
  AttentionPool models/vision_transformer.py:341-374   (nn.MultiheadAttention add_bias_kv, :353)

Execution model (MI355X-first, not a translation): the modules only HOLD parameters; the
forward/backward are explicit launch sequences over a single token stream laid out as
[B, T, 1+N, D] (frame-major, one cls slot per frame, replicated cls) so that
  * spatial attention / MLP / LayerNorm / their GEMMs run over ALL rows with no gather,
  * the temporal branch addresses token rows through the GEMM/LN row maps (no permute copies;
    the reference performs ~10 rearrange/cat/repeat copies per block, :247-274),
  * gradients of the replicated cls slots simply add up (every consumer is linear in them).
Nothing here uses autograd: backward() consumes the tape saved by forward() and writes
parameter gradients straight into the parameters' .grad buffers (views of the engine's flat
gradient buffer when the DP engine is used).  torch.utils.checkpoint of the reference (:575-577)
is dropped: 288 GB of HBM hold every activation.
"""
from __future__ import annotations

import math
import os
from typing import List

import torch
from torch import nn

from . import ops
from .ops import ACT_GELU_ERF


def _param(*shape, std=0.02, device=None, dtype=torch.bfloat16, const=None):
    if const is not None:
        t = torch.full(shape, const, dtype=dtype, device=device)
    else:
        t = (torch.randn(*shape, device=device, dtype=torch.float32) * std).to(dtype)
    return nn.Parameter(t)


def grad_of(p: nn.Parameter) -> torch.Tensor:
    """The buffer backward() writes into (allocated lazily when no engine pre-assigned one)."""
    if p.grad is None:
        p.grad = torch.zeros_like(p)
    return p.grad


class Linear(nn.Module):
    def __init__(self, inp, out, bias=True, std=0.02, device=None):
        super().__init__()
        self.in_features, self.out_features = inp, out
        self.weight = _param(out, inp, std=std, device=device)
        self.bias = _param(out, const=0.0, device=device) if bias else None


class LayerNormWithForceFP32(nn.Module):
    """Parameter holder; the fp32-statistics LN itself is mpv_layernorm_{fwd,bwd}."""

    def __init__(self, dim, eps=1e-6, device=None):
        super().__init__()
        self.eps = eps
        self.normalized_shape = (dim,)
        self.weight = _param(dim, const=1.0, device=device)
        self.bias = _param(dim, const=0.0, device=device)


class Mlp(nn.Module):
    def __init__(self, dim, hidden, std, device=None):
        super().__init__()
        self.fc1 = Linear(dim, hidden, std=std, device=device)
        self.fc2 = Linear(hidden, dim, std=std, device=device)


class Attention(nn.Module):
    def __init__(self, dim, heads, std, device=None):
        super().__init__()
        self.num_heads = heads
        self.scale = (dim // heads) ** -0.5
        self.qkv = Linear(dim, 3 * dim, bias=False, std=std, device=device)
        self.q_bias = _param(dim, const=0.0, device=device)
        self.v_bias = _param(dim, const=0.0, device=device)
        self.proj = Linear(dim, dim, std=std, device=device)


class Block(nn.Module):
    def __init__(self, dim, heads, mlp_ratio, eps, std, device=None):
        super().__init__()
        self.norm1 = LayerNormWithForceFP32(dim, eps, device)
        self.attn = Attention(dim, heads, std, device)
        self.norm2 = LayerNormWithForceFP32(dim, eps, device)
        self.mlp = Mlp(dim, int(dim * mlp_ratio), std, device)
        self.temporal_attn = Attention(dim, heads, std, device)
        self.temporal_ln = LayerNormWithForceFP32(dim, eps, device)
        self.temporal_fc = Linear(dim, dim, std=std, device=device)


class PatchEmbed(nn.Module):
    def __init__(self, img_size, patch_size, in_chans, dim, bias, std, device=None):
        super().__init__()
        self.img_size, self.patch_size = (img_size, img_size), (patch_size, patch_size)
        self.num_patches = (img_size // patch_size) ** 2
        self.proj = nn.Module()
        self.proj.weight = _param(dim, in_chans, patch_size, patch_size, std=std, device=device)
        self.proj.bias = _param(dim, const=0.0, device=device) if bias else None


def convert_pretrained_vit(weights: dict) -> dict:
    """models/vision_transformer.py:719-729 (_convert_pretrained_vit): a timm / CLIP ViT state dict keeps one fused
    `qkv.bias`; the video ViT holds `q_bias` / `v_bias` (the key bias is identically zero, :173) and no `head`."""
    for key in list(weights.keys()):
        if "qkv.bias" in key:
            q, _, v_ = weights[key].chunk(3)
            weights[key.replace("qkv.bias", "q_bias")] = q
            weights[key.replace("qkv.bias", "v_bias")] = v_
            del weights[key]
        elif "head" in key:
            del weights[key]
    return weights


def resize_pos_embed(posemb: torch.Tensor, posemb_new: torch.Tensor) -> torch.Tensor:
    """models/vision_transformer.py:731-750: bilinear re-grid of the patch position embeddings (cls slot kept) when a
    checkpoint was trained at another resolution.  Host-side checkpoint surgery (runs once, on CPU tensors)."""
    ntok_new = posemb_new.shape[1] - 1
    tok, grid = posemb[:, :1], posemb[0, 1:]
    gs_old, gs_new = int(math.sqrt(len(grid))), int(math.sqrt(ntok_new))
    grid = grid.reshape(1, gs_old, gs_old, -1).permute(0, 3, 1, 2)
    orig = grid.dtype
    grid = torch.nn.functional.interpolate(grid.float(), size=(gs_new, gs_new), mode="bilinear").to(orig)
    grid = grid.permute(0, 2, 3, 1).reshape(1, gs_new * gs_new, -1)
    return torch.cat([tok, grid], dim=1)


def resize_temporal_embed(posemb: torch.Tensor, posemb_new: torch.Tensor, mode: str = "interpolate") -> torch.Tensor:
    """models/vision_transformer.py:753-764: linear interpolation (or zero padding / truncation) of the per-frame
    embeddings when the number of frames changes between pre-training and a downstream run."""
    n_new, n_old = posemb_new.shape[1], posemb.shape[1]
    if mode == "padding":
        if n_old <= n_new:
            out = posemb_new.detach().clone()
            out[:, :n_old] = posemb
            return out
        return posemb[:, :n_new]
    orig = posemb.dtype
    out = torch.nn.functional.interpolate(posemb.float().permute(0, 2, 1), n_new, mode="linear")
    return out.permute(0, 2, 1).to(orig)


def resize_visual_embeds_in_state_dict(state_dict: dict, model: nn.Module, prefix: str = "visual_encoder.") -> dict:
    """What every downstream `--resume` does before load_state_dict (downstream/run_retrieval_distributed_gpt3.py:
    402-420): fit the checkpoint's pos / temporal embeddings to this model's grid and frame count."""
    own = dict(model.named_parameters())
    for key, fn in ((prefix + "pos_embed", resize_pos_embed), (prefix + "temporal_embed", resize_temporal_embed)):
        if key in state_dict and key in own and tuple(state_dict[key].shape) != tuple(own[key].shape):
            state_dict[key] = fn(state_dict[key], own[key].detach().cpu())
    return state_dict


class _WgradLane:
    """Second HIP stream for the weight-gradient GEMMs of the ViT backward (MPV_WGRAD_STREAM=0 turns it off).  They are off
    the critical dX chain and their operands are complete when they are issued, so on a second stream their workgroups
    can fill the CUs a dgrad launch leaves idle in its last, partially filled round of 256x256 tiles (N = 768: 591 tiles =
    2.3 rounds of 256).  `sync()` makes the main stream wait for everything issued here (before an operand is modified
    in place, and at the end of a block before its gradients are handed to the reducer).  Measured at config B:
    87.3 -> 85.5 ms per step, bit-identical losses."""

    def __init__(self, device):
        import os
        self.on = os.environ.get("MPV_WGRAD_STREAM", "1") == "1" and torch.device(device).type == "cuda"
        self.side = torch.cuda.Stream(device=device) if self.on else None

    def __call__(self, fn, *tensors):
        if not self.on:
            fn()
            return
        self.side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.side):
            fn()
        for t in tensors:
            t.record_stream(self.side)

    def sync(self):
        if self.on:
            torch.cuda.current_stream().wait_stream(self.side)

    def check_memory(self, device):
        """Called once, at the start of the first backward (the forward's activations are allocated by then).  A second stream is a
        second pool of the caching allocator: blocks the main stream frees while the side stream still uses them cannot be
        reused at once, and the reserved memory drifts towards twice the activations.  When the activations already take more than
        a third of the device (the ITC retrieval step at its YAML batch of 96 x 16 frames: 130 GiB allocated, 259 GiB reserved of
        288, then multi-second allocator retries) the lane buys nothing anyway (its launches are large) and is switched off."""
        if not self.on or getattr(self, "_mem_checked", False):
            return
        self._mem_checked = True
        total = torch.cuda.get_device_properties(device).total_memory
        if torch.cuda.memory_allocated(device) > total // 3:
            self.on = False


# MPV_VIT_COMPOSE=0: measurement knob -- temporal_attn.proj / temporal_fc as the reference's two products (forward) and two dgrads +
# two wgrads (backward) instead of the composed projection (TimeSformer.forward_features / backward_features)
COMPOSE_TEMPORAL_OUT = os.environ.get("MPV_VIT_COMPOSE", "1") != "0"
_COMPOSE_ON_LANE = os.environ.get("MPV_VIT_COMPOSE_LANE", "1") != "0"    # measurement knob: 0 = the composed weights are built on the main stream
# Round 6: the [D, D] chain-rule products of the composed projection depend on parameters and per-block reduced gradients only, so all
# blocks' worth goes out in ONE launch per kind (mpv_gemm_bf16_batched, mpv_vit_compose_bias_batched, ..._finish_batched): Wc / bc at the
# head of the step, dWf / dWp / d(bp) at the end of the tower's backward -- 5 launches per step instead of 60 (36 of them 768^3 products of
# 36 workgroups at 42-50 TFLOP/s).  Those gradients (and temporal_fc.bias, an operand of the finish) go out with the stem, not with their block (late_grad_params: the engine
# puts them into the stem's bucket).  MPV_VIT_COMPOSE_GROUP=0: measurement knob, the per-block launches of rounds 3-5.
COMPOSE_GROUPED = os.environ.get("MPV_VIT_COMPOSE_GROUP", "1") != "0"      # (only read where COMPOSE_TEMPORAL_OUT is on)
# Round 6: the spatial attention's `q * scale` (a second bf16 rounding, models/vision_transformer.py:179) is applied by the qkv product's
# epilogue (mpv_gemm_epilogue.colscale) instead of by each of the three attention kernels on its q rows / q image per work item.
# MPV_VIT_PRESCALE_Q=0: measurement knob, the kernels scale (rounds 1-5).  Bit-identical either way (tested).
PRESCALE_Q = os.environ.get("MPV_VIT_PRESCALE_Q", "1") != "0"
_SMALL_TILE = int(os.environ.get("MPV_VIT_SMALL_TILE", "128"))    # tile kernel of the [D, D] chain-rule products: 36 tiles of 128x128 beat 9 of 256x256 (same-box 78.5 -> 78.35 ms per step)


def _qkv_bias(att: Attention):
    # models/vision_transformer.py:173: cat(q_bias, zeros, v_bias)
    return torch.cat([att.q_bias.detach(), torch.zeros_like(att.v_bias), att.v_bias.detach()])


class _PackedQkvBias:
    """[len(atts), 3D]: row i = cat(q_bias, zeros, v_bias) of atts[i] (models/vision_transformer.py:173).  The buffer is kept
    across steps (its key third stays zero); refresh() re-copies the q / v halves of every attention of the tower in ONE
    launch (mpv_copy_segments) -- the framework form was a zeros + cat pair per attention, then five launches per tower."""

    def __init__(self):
        self.buf = None

    def refresh(self, atts):
        D = atts[0].q_bias.shape[0]
        dev = atts[0].q_bias.device
        if self.buf is None or self.buf.shape != (len(atts), 3 * D) or self.buf.device != dev:
            self.buf = torch.zeros((len(atts), 3 * D), dtype=torch.bfloat16, device=dev)
        pairs = []
        for i, a in enumerate(atts):
            pairs.append((a.q_bias.detach(), self.buf[i, :D]))
            pairs.append((a.v_bias.detach(), self.buf[i, 2 * D:]))
        ops.copy_segments(pairs)
        return self.buf


_zero_rows = {}


def _zero_row(cols, device):
    """A [1, cols] bf16 row of zeros kept per device: the source of row-mapped zero fills (mpv_copy_rows with a source map of
    stride 0), which touch only the rows that need zeroing instead of clearing a whole activation-sized tensor."""
    key = (cols, str(device))
    z = _zero_rows.get(key)
    if z is None:
        z = _zero_rows[key] = torch.zeros((1, cols), dtype=torch.bfloat16, device=device)
    return z


class TimeSformer(nn.Module):
    def __init__(self, img_size=224, num_frames=4, patch_size=16, in_chans=3, embed_dim=768, depth=12, num_heads=8,
                 mlp_ratio=4.0, eps=1e-6, init_std=0.015, clip_model=True, device=None, **_):
        super().__init__()
        D = embed_dim
        assert (D // num_heads) % 8 == 0 and D // num_heads <= 96, "fused attention kernels take head_dim = multiple of 8, <= 96"
        self.embed_dim = self.num_features = D
        self.num_frames, self.num_heads, self.depth = num_frames, num_heads, depth
        self.patch_embed = PatchEmbed(img_size, patch_size, in_chans, D, bias=not clip_model, std=init_std, device=device)
        N = self.patch_embed.num_patches
        if clip_model:
            self.norm_pre = LayerNormWithForceFP32(D, eps, device)
        self.cls_token = _param(1, 1, D, std=init_std, device=device)
        self.pos_embed = _param(1, N + 1, D, std=init_std, device=device)
        self.temporal_embed = _param(1, num_frames, D, const=0.0, device=device)
        self.blocks = nn.ModuleList([Block(D, num_heads, mlp_ratio, eps, init_std, device) for _ in range(depth)])
        self.norm = LayerNormWithForceFP32(D, eps, device)
        with torch.no_grad():   # fix_init_weight (:513-519) and the temporal_fc zero-init quirk (:491-498)
            for i, blk in enumerate(self.blocks):
                blk.attn.proj.weight.div_(math.sqrt(2.0 * (i + 1)))
                blk.mlp.fc2.weight.div_(math.sqrt(2.0 * (i + 1)))
                blk.temporal_fc.weight.zero_()
                blk.temporal_fc.bias.zero_()
        self.on_block_grads_ready = None   # engine hook: called with the block index after its backward

    def no_weight_decay(self):
        return {"temporal_embed", "pos_embed", "cls_token"}

    def late_grad_params(self):
        """Parameters whose gradients may only be handed to the reducer at the END of the tower's backward (with the stem), not with their block: the
        composed temporal projection's chain-rule products of all blocks run as one batched launch there (COMPOSE_GROUPED).  The
        engine's bucket layout reads this (engine.default_stages)."""
        if not (COMPOSE_TEMPORAL_OUT and COMPOSE_GROUPED):
            return []
        # temporal_fc.bias too: its gradient d(bc) = colsum d(xt) is complete with its block, but it is an OPERAND of the batched finish
        # (dWf += d(bc) bp^T, d(bp) = Wf^T d(bc)) -- handed to the reducer with its block it would be the cross-rank SUM by then (an
        # in-place all-reduce), and the rank-1 terms would count `world` times (found by the two-rank GPU test of round 6)
        return [p for blk in self.blocks for p in (blk.temporal_fc.weight, blk.temporal_fc.bias, blk.temporal_attn.proj.weight, blk.temporal_attn.proj.bias)]

    # ------------------------------------------------------------------ forward
    def forward_features(self, video: torch.Tensor, tape: dict):
        """video [B,3,T,H,W] bf16 -> image_embeds [B*(1+T*N), D] (cls first, then frame-major tokens,
        exactly the reference's `b (t n) c` order, :582-585)."""
        B, Cc, T, H, W = video.shape
        assert T == self.num_frames, (T, self.num_frames)
        D, P = self.embed_dim, self.patch_embed.patch_size[0]
        N = self.patch_embed.num_patches
        N1, heads, hd = N + 1, self.num_heads, self.embed_dim // self.num_heads
        R, Rt = B * T * N1, B * T * N
        tok = (N, N1, 1)
        Kc = Cc * P * P
        Kp = (Kc + 7) // 8 * 8
        video = video.contiguous()
        cols = ops.im2col_patches(video, B, Cc, T, H, W, P, Kp)
        wpe = self.patch_embed.proj.weight.detach().view(D, Kc)
        if Kp != Kc:
            wpe = torch.nn.functional.pad(wpe, (0, Kp - Kc))
        patch = ops.gemm(cols, wpe, Rt, D, Kp, bias=self.patch_embed.proj.bias)
        x0 = ops.vit_embed_assemble_fwd(patch, self.cls_token, self.pos_embed, self.temporal_embed, B, T, N, D)
        tape.update(B=B, T=T, N=N, cols=cols, Kp=Kp, Kc=Kc, x0=x0)
        if hasattr(self, "norm_pre"):
            x, m, r = ops.layernorm_fwd(x0, self.norm_pre.weight, self.norm_pre.bias, self.norm_pre.eps, R, D)
            tape["pre_stats"] = (m, r)
        else:
            x = x0
        blocks: List[dict] = []
        composed = None
        if COMPOSE_TEMPORAL_OUT:
            # Wc = Wf Wp and bc = Wf bp + bf of every block depend on parameters only: all 24 small products go to the second
            # stream now and run beside the patch embedding and the first block's products (36 workgroups each, they fit between
            # the tiles of the big launches); the main stream waits for them once, before block 0 uses its pair.  On the
            # critical path they were 12 x (18 + 7) us per step.
            wl = _WgradLane(video.device) if not hasattr(self, "_wgrad_lane") else self._wgrad_lane
            self._wgrad_lane = wl
            composed = []

            def _compose_all():
                if COMPOSE_GROUPED:
                    nb = len(self.blocks)
                    wcs = torch.empty((nb, D, D), dtype=torch.bfloat16, device=video.device)
                    bcs = torch.empty((nb, D), dtype=torch.bfloat16, device=video.device)
                    wfs = [blk.temporal_fc.weight.detach() for blk in self.blocks]
                    ops.gemm_batched(wfs, [blk.temporal_attn.proj.weight.detach() for blk in self.blocks], list(wcs.unbind(0)),
                                     D, D, D, trans_b=True)                                                  # Wc = Wf Wp, every block
                    ops.vit_compose_bias_batched(wfs, [blk.temporal_attn.proj.bias.detach() for blk in self.blocks],
                                                 [blk.temporal_fc.bias.detach() for blk in self.blocks], list(bcs.unbind(0)), D)   # bc = Wf bp + bf
                    composed.extend(zip(wcs.unbind(0), bcs.unbind(0)))
                    return
                for blk in self.blocks:
                    wf, wp = blk.temporal_fc.weight.detach(), blk.temporal_attn.proj.weight.detach()
                    wc = ops.gemm(wf, wp, D, D, D, trans_b=True, tile_hint=_SMALL_TILE)                      # Wc = Wf Wp
                    bc = ops.gemm(blk.temporal_attn.proj.bias.detach().view(1, D), wf, 1, D, D, bias=blk.temporal_fc.bias)   # bc = Wf bp + bf
                    composed.append((wc, bc))
            if wl.on and _COMPOSE_ON_LANE:
                main = torch.cuda.current_stream()
                wl(_compose_all)
                for wc, bc in composed:       # allocated on the second stream, consumed on the main one
                    wc.record_stream(main)
                    bc.record_stream(main)
            else:
                _compose_all()
        if not hasattr(self, "_qkv_bias_pack"):
            self._qkv_bias_pack = _PackedQkvBias()
        qkv_b = self._qkv_bias_pack.refresh([a for blk in self.blocks for a in (blk.temporal_attn, blk.attn)])
        for bi, blk in enumerate(self.blocks):
            s = {}
            # ---- temporal branch on token rows (:247-251)
            lt, s["mt"], s["rt"] = ops.layernorm_fwd(x, blk.temporal_ln.weight, blk.temporal_ln.bias, blk.temporal_ln.eps,
                                                     Rt, D, xmap=tok, ymap=tok, out_rows=R)
            qkv_t = ops.gemm(lt, blk.temporal_attn.qkv.weight, Rt, 3 * D, D, bias=qkv_b[2 * bi],
                             amap=tok, cmap=tok, out_rows=R)
            at = torch.empty((R, D), dtype=torch.bfloat16, device=x.device)
            ops.temporal_attn_fwd(qkv_t, at, B, T * N1, N, 1, N1, T, heads, hd, blk.temporal_attn.scale)
            xt = torch.empty_like(x)
            if COMPOSE_TEMPORAL_OUT:
                # temporal_attn.proj followed by temporal_fc (:199-200 then :250; proj_drop = 0, only a rearrange between them) is ONE
                # linear map: xt = x + a Wc^T + bc with Wc = Wf Wp, bc = Wf bp + bf.  One product over the 50176 token rows instead
                # of two, at the price of a [D, D] product and a matrix-vector product per block per step (the weights move every
                # step).  The reference rounds proj(a) to bf16 between the two; here Wc is what is rounded -- the same size of
                # perturbation, checked against the reference goldens (forward values and every gradient).
                if bi == 0:
                    self._wgrad_lane.sync()                     # the composed weights of all blocks (second stream, see above)
                wc, bc = composed[bi]
                ops.gemm(at, wc, Rt, D, D, bias=bc, residual=x, amap=tok, cmap=tok, out=xt)
                pt = None
                s["wc"] = wc
            else:
                pt = ops.gemm(at, blk.temporal_attn.proj.weight, Rt, D, D, bias=blk.temporal_attn.proj.bias, amap=tok, cmap=tok,
                              out_rows=R)
                ops.gemm(pt, blk.temporal_fc.weight, Rt, D, D, bias=blk.temporal_fc.bias, residual=x, amap=tok, cmap=tok, out=xt)
            ops.copy_rows(x, xt, B * T, D, smap=(1, N1, 0), dmap=(1, N1, 0))          # cls slots pass through
            # ---- spatial branch on all rows (:254-267)
            l1, s["m1"], s["r1"] = ops.layernorm_fwd(xt, blk.norm1.weight, blk.norm1.bias, blk.norm1.eps, R, D)
            # (PRESCALE_Q: the q third leaves the product as bf16(bf16(q) * scale) -- `q * self.scale`, :179 -- so that none of the
            # three attention kernels re-scales its q rows per work item; dQ stays the gradient of the unscaled q)
            qkv_s = ops.gemm(l1, blk.attn.qkv.weight, R, 3 * D, D, bias=qkv_b[2 * bi + 1],
                             colscale=(D, blk.attn.scale) if PRESCALE_Q else None)
            a_s = torch.empty((R, D), dtype=torch.bfloat16, device=x.device)
            st3 = (N1 * 3 * D, hd, 3 * D)
            lay = ops.AttnLayout(st3, st3, st3, (N1 * D, hd, D))
            lse = ops.attn_fwd(qkv_s, qkv_s[:, D:], qkv_s[:, 2 * D:], a_s, lay, B * T, heads, N1, N1, hd,
                               scale=blk.attn.scale, scale_q_bf16=2 if PRESCALE_Q else True)
            # y = xt + proj(a_s) straight from the GEMM's residual epilogue; the projection's cls rows are tapped out of the
            # same launch and only the B*T cls slots are rewritten as xt_cls + mean_t(proj cls)              (:263-270)
            tap = torch.empty((B * T, D), dtype=torch.bfloat16, device=x.device)
            y = ops.gemm(a_s, blk.attn.proj.weight, R, D, D, bias=blk.attn.proj.bias, residual=xt, row_tap_out=tap, row_tap_group=N1)
            ops.vit_cls_fix_fwd(xt, tap, y, B, T, N1, D)
            # ---- MLP (:271)
            l2, s["m2"], s["r2"] = ops.layernorm_fwd(y, blk.norm2.weight, blk.norm2.bias, blk.norm2.eps, R, D)
            hid = blk.mlp.fc1.out_features
            z = torch.empty((R, hid), dtype=torch.bfloat16, device=x.device)
            h1 = ops.gemm(l2, blk.mlp.fc1.weight, R, hid, D, bias=blk.mlp.fc1.bias, act=ACT_GELU_ERF, preact_out=z, preact_deriv=True)
            out = ops.gemm(h1, blk.mlp.fc2.weight, R, D, hid, bias=blk.mlp.fc2.bias, residual=y)
            s.update(x=x, lt=lt, qkv_t=qkv_t, at=at, pt=pt, xt=xt, l1=l1, qkv_s=qkv_s, a_s=a_s, lse=lse, lay=lay, y=y, l2=l2,
                     z=z, h1=h1)
            blocks.append(s)
            x = out
        tape["blocks"] = blocks
        tape["x_last"] = x
        # ---- final LN + compaction to [B, 1+T*N, D] (:582-585)
        S = 1 + T * N
        emb = torch.empty((B * S, D), dtype=torch.bfloat16, device=x.device)
        _, mt_, rt_ = ops.layernorm_fwd(x, self.norm.weight, self.norm.bias, self.norm.eps, Rt, D, out=emb, xmap=tok,
                                        ymap=(T * N, S, 1))
        _, mc_, rc_ = ops.layernorm_fwd(x, self.norm.weight, self.norm.bias, self.norm.eps, B, D, out=emb,
                                        xmap=(1, T * N1, 0), ymap=(1, S, 0))
        tape["final_stats"] = (mt_, rt_, mc_, rc_)
        return emb

    # ------------------------------------------------------------------ backward
    def backward_features(self, demb: torch.Tensor, tape: dict):
        B, T, N = tape["B"], tape["T"], tape["N"]
        D, heads, hd = self.embed_dim, self.num_heads, self.embed_dim // self.num_heads
        N1, R, Rt, S = N + 1, B * T * (N + 1), B * T * N, 1 + T * N
        tok = (N, N1, 1)
        x_last = tape["x_last"]
        mt_, rt_, mc_, rc_ = tape["final_stats"]
        # cls slots t>0 get no final-LN gradient: zero the B*T cls rows (the t=0 ones are then overwritten below) instead of
        # clearing the whole [R, D] tensor
        dx = torch.empty((R, D), dtype=torch.bfloat16, device=demb.device)
        ops.copy_rows(_zero_row(D, demb.device), dx, B * T, D, smap=(1, 0, 0), dmap=(1, N1, 0))
        gw, gb = grad_of(self.norm.weight), grad_of(self.norm.bias)
        ops.layernorm_bwd(demb, x_last, self.norm.weight, mt_, rt_, Rt, D, dx=dx, dgamma=gw, dbeta=gb, xmap=tok,
                          ymap=(T * N, S, 1))
        ops.layernorm_bwd(demb, x_last, self.norm.weight, mc_, rc_, B, D, dx=dx, dgamma=gw, dbeta=gb, accumulate_dparams=True,
                          xmap=(1, T * N1, 0), ymap=(1, S, 0))
        wl = _WgradLane(demb.device) if not hasattr(self, "_wgrad_lane") else self._wgrad_lane
        self._wgrad_lane = wl
        wl.check_memory(demb.device)
        lnb = self.__dict__.setdefault("_ln_batch", ops.LnDparamBatch())   # a block's three dgamma/dbeta reductions: one launch
        grouped = COMPOSE_TEMPORAL_OUT and COMPOSE_GROUPED
        dwcs = torch.empty((len(self.blocks), D, D), dtype=torch.bfloat16, device=demb.device) if grouped else None   # dWc of every block
        for bi in range(len(self.blocks) - 1, -1, -1):
            blk, s = self.blocks[bi], tape["blocks"][bi]
            hid = blk.mlp.fc1.out_features
            dout = dx
            # ---- MLP
            wl(lambda: ops.gemm(dout, s["h1"], D, hid, R, trans_a=True, trans_b=True, out=grad_of(blk.mlp.fc2.weight),
                                colsum_out=grad_of(blk.mlp.fc2.bias)), dout)
            dz = ops.gemm(dout, blk.mlp.fc2.weight, R, hid, D, trans_b=True, act_bwd_z=s["z"], act_bwd=ACT_GELU_ERF, z_is_deriv=True)
            wl(lambda: ops.gemm(dz, s["l2"], hid, D, R, trans_a=True, trans_b=True, out=grad_of(blk.mlp.fc1.weight),
                                colsum_out=grad_of(blk.mlp.fc1.bias)), dz)
            dl2 = ops.gemm(dz, blk.mlp.fc1.weight, R, D, hid, trans_b=True)
            dy = ops.layernorm_bwd(dl2, s["y"], blk.norm2.weight, s["m2"], s["r2"], R, D, dres=dout,
                                   dgamma=grad_of(blk.norm2.weight), dbeta=grad_of(blk.norm2.bias), defer=lnb)
            # ---- cls merge + spatial attention
            # the projection sees dy with every cls row replaced by the mean over t: done in place on the B*T cls rows,
            # which are put back before dy is used as the residual gradient
            cls_saved = ops.vit_cls_merge_bwd_inplace(dy, B, T, N1, D)
            wl(lambda: ops.gemm(dy, s["a_s"], D, D, R, trans_a=True, trans_b=True, out=grad_of(blk.attn.proj.weight),
                                colsum_out=grad_of(blk.attn.proj.bias)), dy)
            das = ops.gemm(dy, blk.attn.proj.weight, R, D, D, trans_b=True)
            wl.sync()                                   # the wgrad reads the merged cls rows
            ops.copy_rows(cls_saved, dy, B * T, D, dmap=(1, N1, 0))
            qkv_s = s["qkv_s"]
            dqkv = torch.empty_like(qkv_s)
            ops.attn_bwd(qkv_s, qkv_s[:, D:], qkv_s[:, 2 * D:], s["a_s"], s["lse"], das, dqkv, dqkv[:, D:], dqkv[:, 2 * D:],
                         s["lay"], B * T, heads, N1, N1, hd, scale=blk.attn.scale, scale_q_bf16=2 if PRESCALE_Q else True)
            bsum = torch.empty((2, 3 * D), dtype=torch.bfloat16, device=dqkv.device)   # packed bias gradients of the two qkv products
            wl(lambda: ops.gemm(dqkv, s["l1"], 3 * D, D, R, trans_a=True, trans_b=True, out=grad_of(blk.attn.qkv.weight),
                                colsum_out=bsum[0]), dqkv)
            dl1 = ops.gemm(dqkv, blk.attn.qkv.weight, R, D, 3 * D, trans_b=True)
            dxt = ops.layernorm_bwd(dl1, s["xt"], blk.norm1.weight, s["m1"], s["r1"], R, D, dres=dy,
                                    dgamma=grad_of(blk.norm1.weight), dbeta=grad_of(blk.norm1.bias), defer=lnb)
            # ---- temporal branch (token rows)
            if COMPOSE_TEMPORAL_OUT:
                # Backward of temporal_attn.proj followed by temporal_fc (:199-200 then :250; proj_drop = 0, only a rearrange
                # between them) as ONE linear map: xt = x + a Wc^T + bc with Wc = Wf Wp, bc = Wf bp + bf.  One dgrad
                # d(a) = d(xt) Wc and one wgrad dWc = d(xt)^T a over the 50176 token rows instead of two of each; the chain
                # rule through the composition runs on [D, D] operands (768^3 products): dWf = dWc Wp^T + d(bc) bp^T,
                # dWp = Wf^T dWc, d(bp) = Wf^T d(bc), d(bf) = d(bc).  (The forward is composed the same way: forward_features.)
                wf, wp = blk.temporal_fc.weight.detach(), blk.temporal_attn.proj.weight.detach()
                wc = s["wc"]                                                                       # Wc = Wf Wp, from the forward

                def _temporal_out_wgrad(dxt=dxt, wf=wf, wp=wp, bi=bi):
                    dbc = grad_of(blk.temporal_fc.bias)                                   # d(bf) = d(bc) = colsum d(xt)
                    if grouped:         # dWc is parked; the [D, D] chain rule of all blocks runs as one batch behind block 0
                        ops.gemm(dxt, s["at"], D, D, Rt, trans_a=True, trans_b=True, kmap=tok, colsum_out=dbc, out=dwcs[bi])
                        return
                    dwc = ops.gemm(dxt, s["at"], D, D, Rt, trans_a=True, trans_b=True, kmap=tok, colsum_out=dbc)
                    dwf = ops.gemm(dwc, wp, D, D, D, tile_hint=_SMALL_TILE)               # dWc Wp^T
                    ops.gemm(wf, dwc, D, D, D, trans_a=True, trans_b=True, out=grad_of(blk.temporal_attn.proj.weight), tile_hint=_SMALL_TILE)   # Wf^T dWc
                    # dWf = dWc Wp^T + d(bc) bp^T and d(bp) = Wf^T d(bc) in one launch
                    ops.vit_compose_bwd_finish(dwf, dbc, blk.temporal_attn.proj.bias.detach(), wf, grad_of(blk.temporal_fc.weight),
                                               grad_of(blk.temporal_attn.proj.bias), D)
                wl(_temporal_out_wgrad, dxt)
                dat = ops.gemm(dxt, wc, Rt, D, D, trans_b=True, amap=tok, cmap=tok, out_rows=R)
            else:       # the reference's two dgrads and two wgrads (measurement: MPV_VIT_COMPOSE=0)
                wl(lambda: ops.gemm(dxt, s["pt"], D, D, Rt, trans_a=True, trans_b=True, kmap=tok, out=grad_of(blk.temporal_fc.weight),
                                    colsum_out=grad_of(blk.temporal_fc.bias)), dxt)
                dpt = ops.gemm(dxt, blk.temporal_fc.weight, Rt, D, D, trans_b=True, amap=tok, cmap=tok, out_rows=R)
                wl(lambda: ops.gemm(dpt, s["at"], D, D, Rt, trans_a=True, trans_b=True, kmap=tok, out=grad_of(blk.temporal_attn.proj.weight),
                                    colsum_out=grad_of(blk.temporal_attn.proj.bias)), dpt)
                dat = ops.gemm(dpt, blk.temporal_attn.proj.weight, Rt, D, D, trans_b=True, amap=tok, cmap=tok, out_rows=R)
            dqkv_t = torch.empty_like(s["qkv_t"])
            ops.temporal_attn_bwd(s["qkv_t"], dat, dqkv_t, B, T * N1, N, 1, N1, T, heads, hd, blk.temporal_attn.scale)
            def _qkv_t_wgrad(dqkv_t=dqkv_t, bsum=bsum):
                ops.gemm(dqkv_t, s["lt"], 3 * D, D, Rt, trans_a=True, trans_b=True, kmap=tok,
                         out=grad_of(blk.temporal_attn.qkv.weight), colsum_out=bsum[1])
                # q / v thirds of both packed bias gradients -> the four bias parameters' gradients (the key third belongs to no
                # parameter: the key bias is identically zero, models/vision_transformer.py:173), one launch
                ops.copy_segments([(bsum[0, :D], grad_of(blk.attn.q_bias)), (bsum[0, 2 * D:], grad_of(blk.attn.v_bias)),
                                   (bsum[1, :D], grad_of(blk.temporal_attn.q_bias)), (bsum[1, 2 * D:], grad_of(blk.temporal_attn.v_bias))])
            wl(_qkv_t_wgrad, dqkv_t, bsum)
            dlt = ops.gemm(dqkv_t, blk.temporal_attn.qkv.weight, Rt, D, 3 * D, trans_b=True, amap=tok, cmap=tok, out_rows=R)
            wl.sync()                                   # temporal_fc's wgrad reads dxt, which the next launch updates in place
            # dx = dxt (all rows) + LN_t-backward on token rows, accumulated in place
            ops.layernorm_bwd(dlt, s["x"], blk.temporal_ln.weight, s["mt"], s["rt"], Rt, D, dres=dxt, dx=dxt,
                              dgamma=grad_of(blk.temporal_ln.weight), dbeta=grad_of(blk.temporal_ln.bias), xmap=tok, ymap=tok,
                              defer=lnb)
            lnb.finish()
            dx = dxt
            tape["blocks"][bi] = None          # release activations
            if self.on_block_grads_ready is not None:
                self.on_block_grads_ready(bi)
        if grouped:
            # dWf = dWc Wp^T + d(bc) bp^T, dWp = Wf^T dWc, d(bp) = Wf^T d(bc) of EVERY block: three launches, on the second stream beside the
            # stem's backward (their operands -- parameters, the parked dWc, the blocks' d(bc) -- are complete)
            def _compose_chain_rule_all():
                blks = list(self.blocks)
                wfs = [b.temporal_fc.weight.detach() for b in blks]
                wps = [b.temporal_attn.proj.weight.detach() for b in blks]
                dwc_l = list(dwcs.unbind(0))
                pre = torch.empty_like(dwcs)
                ops.gemm_batched(dwc_l, wps, list(pre.unbind(0)), D, D, D)                                     # dWc Wp^T
                ops.gemm_batched(wfs, dwc_l, [grad_of(b.temporal_attn.proj.weight) for b in blks], D, D, D, trans_a=True, trans_b=True)   # Wf^T dWc
                ops.vit_compose_bwd_finish_batched(list(pre.unbind(0)), [grad_of(b.temporal_fc.bias) for b in blks],
                                                   [b.temporal_attn.proj.bias.detach() for b in blks], wfs,
                                                   [grad_of(b.temporal_fc.weight) for b in blks], [grad_of(b.temporal_attn.proj.bias) for b in blks], D)
            wl(_compose_chain_rule_all, dwcs)
        if hasattr(self, "norm_pre"):
            m, r = tape["pre_stats"]
            dx = ops.layernorm_bwd(dx, tape["x0"], self.norm_pre.weight, m, r, R, D, dgamma=grad_of(self.norm_pre.weight),
                                   dbeta=grad_of(self.norm_pre.bias))
        dpatch = torch.empty((Rt, D), dtype=torch.bfloat16, device=dx.device)
        ops.vit_embed_assemble_bwd(dx, dpatch, grad_of(self.cls_token), grad_of(self.pos_embed), grad_of(self.temporal_embed),
                                   B, T, N, D)
        Kp, Kc = tape["Kp"], tape["Kc"]
        gw = grad_of(self.patch_embed.proj.weight)
        if Kp == Kc:
            ops.gemm(dpatch, tape["cols"], D, Kp, Rt, trans_a=True, trans_b=True, out=gw)
        else:
            tmp = ops.gemm(dpatch, tape["cols"], D, Kp, Rt, trans_a=True, trans_b=True)
            gw.view(D, Kc).copy_(tmp[:, :Kc])
        if self.patch_embed.proj.bias is not None:
            ops.colsum(dpatch, Rt, D, out=grad_of(self.patch_embed.proj.bias))
        if grouped:
            wl.sync()                                   # the batched chain-rule products (second stream) belong to the stem's bucket
        if self.on_block_grads_ready is not None:
            self.on_block_grads_ready(-1)


class _MHAParams(nn.Module):
    """Parameter holder with nn.MultiheadAttention's names (in_proj_weight/bias, bias_k/v, out_proj)."""

    def __init__(self, dim, heads, std, device=None):
        super().__init__()
        self.embed_dim, self.num_heads = dim, heads
        self.in_proj_weight = _param(3 * dim, dim, std=std, device=device)
        self.in_proj_bias = _param(3 * dim, const=0.0, device=device)
        self.bias_k = _param(1, 1, dim, std=std, device=device)
        self.bias_v = _param(1, 1, dim, std=std, device=device)
        self.out_proj = Linear(dim, dim, std=std, device=device)


class AttentionPool(nn.Module):
    """models/vision_transformer.py:341-374: x = LN1(q); k = LNk(tokens); x = x + MHA(x,k,k) with one
    learned extra kv token; x = x + Mlp(LN2(x))."""

    def __init__(self, dim, num_heads, mlp_ratio=4.0, eps=1e-6, std=0.02, device=None):
        super().__init__()
        self.dim, self.num_heads = dim, num_heads
        self.norm1 = LayerNormWithForceFP32(dim, eps, device)
        self.normk = LayerNormWithForceFP32(dim, eps, device)
        self.attn = _MHAParams(dim, num_heads, std, device)
        self.norm2 = LayerNormWithForceFP32(dim, eps, device)
        self.mlp = Mlp(dim, int(dim * mlp_ratio), std, device)

    def forward_pool(self, queries: torch.Tensor, emb: torch.Tensor, B: int, S: int, tape: dict):
        """queries [1,Q,D] parameter, emb [B*S, D] -> [B*Q, D]."""
        D, heads = self.dim, self.num_heads
        hd = D // heads
        Q = queries.shape[1]
        a = self.attn
        xin = torch.empty((B * Q, D), dtype=torch.bfloat16, device=emb.device)
        ops.copy_rows(queries.detach().view(Q, D), xin, B * Q, D, smap=(Q, 0, 0))   # queries.repeat(B,1,1) (distributed_gpt3.py:134)
        x, m1, r1 = ops.layernorm_fwd(xin, self.norm1.weight, self.norm1.bias, self.norm1.eps, B * Q, D)
        kn, mk, rk = ops.layernorm_fwd(emb, self.normk.weight, self.normk.bias, self.normk.eps, B * S, D)
        Wi, bi = a.in_proj_weight.detach(), a.in_proj_bias.detach()
        q = ops.gemm(x, Wi[:D], B * Q, D, D, bias=bi[:D])
        kv = ops.gemm(kn, Wi[D:], B * S, 2 * D, D, bias=bi[D:], cmap=(S, S + 1, 0), out_rows=B * (S + 1))
        # add_bias_kv token (last key / value row of every sequence)
        ops.copy_rows(a.bias_k.detach().view(1, D), kv, B, D, smap=(1, 0, 0), dmap=(1, S + 1, S), lds=D, ldd=2 * D)
        ops.copy_rows(a.bias_v.detach().view(1, D), kv[:, D:], B, D, smap=(1, 0, 0), dmap=(1, S + 1, S), lds=D, ldd=2 * D)
        o = torch.empty((B * Q, D), dtype=torch.bfloat16, device=emb.device)
        lay = ops.AttnLayout((Q * D, hd, D), ((S + 1) * 2 * D, hd, 2 * D), ((S + 1) * 2 * D, hd, 2 * D), (Q * D, hd, D))
        lse = ops.attn_fwd(q, kv, kv[:, D:], o, lay, B, heads, Q, S + 1, hd, scale=hd ** -0.5)
        x2 = ops.gemm(o, a.out_proj.weight, B * Q, D, D, bias=a.out_proj.bias, residual=x)      # residual from NORMED x
        l2, m2, r2 = ops.layernorm_fwd(x2, self.norm2.weight, self.norm2.bias, self.norm2.eps, B * Q, D)
        hid = self.mlp.fc1.out_features
        z = torch.empty((B * Q, hid), dtype=torch.bfloat16, device=emb.device)
        h1 = ops.gemm(l2, self.mlp.fc1.weight, B * Q, hid, D, bias=self.mlp.fc1.bias, act=ACT_GELU_ERF, preact_out=z, preact_deriv=True)
        out = ops.gemm(h1, self.mlp.fc2.weight, B * Q, D, hid, bias=self.mlp.fc2.bias, residual=x2)
        tape.update(B=B, S=S, Q=Q, xin=xin, x=x, s1=(m1, r1), emb=emb, kn=kn, sk=(mk, rk), q=q, kv=kv, o=o, lse=lse, lay=lay,
                    x2=x2, l2=l2, s2=(m2, r2), z=z, h1=h1)
        return out

    def backward_pool(self, dout: torch.Tensor, queries: nn.Parameter, tape: dict):
        """-> d(emb) [B*S, D]; writes all AttentionPool grads and d(queries)."""
        B, S, Q = tape["B"], tape["S"], tape["Q"]
        D, heads = self.dim, self.num_heads
        hd, hid = D // heads, self.mlp.fc1.out_features
        a = self.attn
        R = B * Q
        ops.gemm(dout, tape["h1"], D, hid, R, trans_a=True, trans_b=True, out=grad_of(self.mlp.fc2.weight), colsum_out=grad_of(self.mlp.fc2.bias))
        dz = ops.gemm(dout, self.mlp.fc2.weight, R, hid, D, trans_b=True, act_bwd_z=tape["z"], act_bwd=ACT_GELU_ERF, z_is_deriv=True)
        ops.gemm(dz, tape["l2"], hid, D, R, trans_a=True, trans_b=True, out=grad_of(self.mlp.fc1.weight), colsum_out=grad_of(self.mlp.fc1.bias))
        dl2 = ops.gemm(dz, self.mlp.fc1.weight, R, D, hid, trans_b=True)
        lnb = self.__dict__.setdefault("_ln_batch", ops.LnDparamBatch())      # the three dgamma/dbeta reductions: one launch at the end
        dx2 = ops.layernorm_bwd(dl2, tape["x2"], self.norm2.weight, *tape["s2"], R, D, dres=dout,
                                dgamma=grad_of(self.norm2.weight), dbeta=grad_of(self.norm2.bias), defer=lnb)
        ops.gemm(dx2, tape["o"], D, D, R, trans_a=True, trans_b=True, out=grad_of(a.out_proj.weight), colsum_out=grad_of(a.out_proj.bias))
        do = ops.gemm(dx2, a.out_proj.weight, R, D, D, trans_b=True)
        kv = tape["kv"]
        dq = torch.empty_like(tape["q"])
        dkv = torch.empty_like(kv)
        ops.attn_bwd(tape["q"], kv, kv[:, D:], tape["o"], tape["lse"], do, dq, dkv, dkv[:, D:], tape["lay"], B, heads, Q, S + 1, hd,
                     scale=hd ** -0.5)
        gW, gb = grad_of(a.in_proj_weight), grad_of(a.in_proj_bias)
        # q projection
        ops.colsum(dq, R, D, out=gb[:D])
        ops.gemm(dq, tape["x"], D, D, R, trans_a=True, trans_b=True, out=gW[:D])
        dx = ops.gemm(dq, a.in_proj_weight.detach()[:D], R, D, D, trans_b=True, residual=dx2)      # + residual path x -> x2
        # k/v projection over the S real tokens of every batch; the bias-kv row goes to bias_k / bias_v
        kmap = (S, S + 1, 0)
        ops.colsum(dkv, B * S, 2 * D, rmap=kmap, out=gb[D:])
        self._kv_wgrad(dkv, tape["kn"], gW[D:], B, S, D)
        ops.colsum(dkv, B, D, ld=2 * D, rmap=(1, S + 1, S), out=grad_of(a.bias_k).view(D))
        ops.colsum(dkv[:, D:], B, D, ld=2 * D, rmap=(1, S + 1, S), out=grad_of(a.bias_v).view(D))
        dkn = ops.gemm(dkv, a.in_proj_weight.detach()[D:], B * S, D, 2 * D, trans_b=True, amap=kmap)
        demb = ops.layernorm_bwd(dkn, tape["emb"], self.normk.weight, *tape["sk"], B * S, D,
                                 dgamma=grad_of(self.normk.weight), dbeta=grad_of(self.normk.bias), defer=lnb)
        dxin = ops.layernorm_bwd(dx, tape["xin"], self.norm1.weight, *tape["s1"], R, D,
                                 dgamma=grad_of(self.norm1.weight), dbeta=grad_of(self.norm1.bias), defer=lnb)
        lnb.finish()
        ops.colsum(dxin, B, Q * D, out=grad_of(queries).view(Q * D))                                # sum over the batch repeat
        return demb

    @staticmethod
    def _kv_wgrad(dkv, kn, gW_kv, B, S, D):
        # dW_kv[2D, D] = sum over real tokens dkv[b,s,:]^T kn[b,s,:]; dkv rows carry the per-batch extra row
        # (map S -> S+1), kn rows are dense, so the two reduction maps differ: compact dkv first (B*S*2D bf16).
        dkv_c = torch.empty((B * S, 2 * D), dtype=torch.bfloat16, device=dkv.device)
        ops.copy_rows(dkv, dkv_c, B * S, 2 * D, smap=(S, S + 1, 0))
        ops.gemm(dkv_c, kn, 2 * D, D, B * S, trans_a=True, trans_b=True, out=gW_kv)
