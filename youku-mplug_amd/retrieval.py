"""DistributedGPT3_Retrieval (ITC) on the gfx950 kernels -- drop-in for models/distributed_gpt3.py:817-985
(BASELINE.json configs[4]).

video feature = L2-normalised vision_proj(ViT cls);  text feature = L2-normalised text_proj(hidden state at the last
valid token of a text-only GPT forward);  features (+ sample ids) are all-gathered over the data-parallel group
(models/distributed_utils.py:285-311: all-gather forward, reduce-scatter backward -- here one RCCL
all_gather_into_tensor / reduce_scatter_tensor each way);  sim/temp;  soft-target cross-entropy in both directions.
The decoder is frozen and nothing trainable sits below the text hidden state, so the text tower has NO backward.
"""
from __future__ import annotations

import json
from typing import Optional

import torch
import torch.distributed as dist
from torch import nn

from . import ops
from .gpt3 import DistributedGPT3, GPT3Config
from .vision import AttentionPool, Linear, TimeSformer, _param, grad_of


def gather_cat(x: torch.Tensor, group=None) -> torch.Tensor:
    """cat(all_gather(x)) along dim 0 (identity at world size 1)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return x
    out = torch.empty((dist.get_world_size(group) * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    dist.all_gather_into_tensor(out, x.contiguous(), group=group)
    return out


def reduce_scatter_sum(g_all: torch.Tensor, group=None) -> torch.Tensor:
    """Backward of gather_cat: this rank's slice of the sum over ranks (models/distributed_utils.py:299-311)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return g_all
    w, r = dist.get_world_size(group), dist.get_rank(group)
    n = g_all.shape[0] // w
    if dist.get_backend(group) == "nccl":
        out = torch.empty((n,) + tuple(g_all.shape[1:]), dtype=g_all.dtype, device=g_all.device)
        dist.reduce_scatter_tensor(out, g_all.contiguous(), op=dist.ReduceOp.SUM, group=group)
        return out
    g = g_all.clone()            # gloo has no reduce_scatter: all-reduce and slice (the reference falls back similarly)
    dist.all_reduce(g, op=dist.ReduceOp.SUM, group=group)
    return g[r * n:(r + 1) * n].contiguous()


class _RetrievalStep(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, model, video, ids, mask, idx):
        loss, tape = model._forward_pipeline(video, ids, mask, idx)
        ctx.model, ctx.tape = model, tape
        return loss

    @staticmethod
    def backward(ctx, grad_loss):
        ctx.model._backward_pipeline(ctx.tape, grad_loss.contiguous().float())
        ctx.tape = None
        return torch.zeros(1, device=grad_loss.device), None, None, None, None, None


class DistributedGPT3_Retrieval(nn.Module):
    def __init__(self, config: Optional[dict] = None, tokenizer=None, *, visual_cfg: Optional[dict] = None,
                 text_cfg: Optional[GPT3Config] = None, device="cuda"):
        super().__init__()
        config = dict(config or {})
        self.tokenizer = tokenizer
        if visual_cfg is None:
            visual_cfg = json.load(open(config["visual_cfg"], "r"))
        if text_cfg is None:
            text_cfg = GPT3Config.from_json_file(config["text_cfg"])
        self.visual_encoder = TimeSformer(
            img_size=visual_cfg["img_size"], num_frames=config.get("num_frames", visual_cfg.get("num_frames", 4)),
            patch_size=visual_cfg["patch_size"], embed_dim=visual_cfg["embed_dim"], depth=visual_cfg["depth"],
            num_heads=visual_cfg["num_heads"], mlp_ratio=visual_cfg["mlp_ratio"], eps=1e-6, init_std=0.015,
            clip_model=visual_cfg.get("clip_model", False), device=device)                                   # :825-841
        if config.get("text_decoder") and not config.get("_synthetic", False):
            self.text_decoder = DistributedGPT3(model_dir=config["text_decoder"], device=device)            # :864-870
        else:
            self.text_decoder = DistributedGPT3(config=text_cfg, device=device)
        if config.get("freeze_vit", False):
            for name, p in self.visual_encoder.named_parameters():
                if not any(x in name for x in ("time", "temporal")):
                    p.requires_grad = False
        if not config.get("freeze_text_decoder", True):
            raise NotImplementedError("the gfx950 path implements the frozen-decoder recipe "
                                      "(configs/retrieval/retrieval_gpt3_1.3B_youku_v0.yaml: freeze_text_decoder: true)")
        for p in self.text_decoder.parameters():
            p.requires_grad = False
        self.vision_width, self.text_width = visual_cfg["embed_dim"], self.text_decoder.config.hidden_size
        self.num_learnable_token = config.get("num_learnable_token", 256)
        # parameters that exist in the reference state-dict but are unused by the ITC forward (:942-945 commented out)
        self.learnable_queries = _param(1, self.num_learnable_token, self.vision_width, std=0.015, device=device)
        self.attn_pool = AttentionPool(self.vision_width, num_heads=visual_cfg["num_heads"], mlp_ratio=visual_cfg["mlp_ratio"],
                                       eps=1e-6, std=0.02, device=device)
        self.visual_fc = Linear(self.vision_width, self.text_width, std=0.015, device=device)
        self.visual_norm = nn.Identity()
        embed_dim = config.get("contrastive_embed_dim", 256)                                                # :904-907
        assert embed_dim % 8 == 0
        self.vision_proj = Linear(self.vision_width, embed_dim, device=device)
        self.text_proj = Linear(self.text_width, embed_dim, device=device)
        self.temp = nn.Parameter(torch.ones([], dtype=torch.bfloat16, device=device) * config.get("temp", 0.07))
        self._anchor = torch.zeros(1, device=device, requires_grad=True)
        self.on_stage_grads_ready = None
        self.process_group = None

    def no_weight_decay(self):
        return {"visual_encoder.pos_embed", "visual_encoder.cls_token", "visual_encoder.temporal_embed"}

    def unused_parameters(self):
        """No gradient ever reaches these in the ITC forward: the engine must not update (weight-decay) them."""
        return [self.learnable_queries, *self.attn_pool.parameters(), *self.visual_fc.parameters()]

    # ------------------------------------------------------------------ towers
    def _vision_tower(self, video, tape):
        B = video.shape[0]
        D, E = self.vision_width, self.vision_proj.out_features
        emb = self.visual_encoder.forward_features(video.to(torch.bfloat16), tape["vit"])
        S = emb.shape[0] // B
        cls = torch.empty((B, D), dtype=torch.bfloat16, device=emb.device)
        ops.copy_rows(emb, cls, B, D, smap=(1, S, 0))                                                       # image_embeds[:, 0]
        vp = ops.gemm(cls, self.vision_proj.weight, B, E, D, bias=self.vision_proj.bias)
        vf, vn = ops.l2norm_fwd(vp, B, E)
        tape.update(B=B, S=S, cls=cls, vp=vp, vn=vn, vf=vf)
        return vf

    def _text_tower(self, ids, mask, tape):
        B, L = ids.shape
        H, E = self.text_width, self.text_proj.out_features
        out = self.text_decoder.forward_lm(None, ids, None, None, {}, hidden_only=True)
        last = mask.sum(dim=-1) - 1                                                                         # :958-959
        rows = torch.arange(B, device=ids.device) * L + last
        pooled = ops.gather_rows(out["last_hidden_state"].reshape(B * L, H), rows.contiguous(), B, H)
        tp = ops.gemm(pooled, self.text_proj.weight, B, E, H, bias=self.text_proj.bias)
        tf, tn = ops.l2norm_fwd(tp, B, E)
        tape.update(pooled=pooled, tp=tp, tn=tn, tf=tf)
        return tf

    @torch.no_grad()
    def extract_vision_feature(self, image):
        return self._vision_tower(image, {"vit": {}})

    @torch.no_grad()
    def extract_text_feature(self, text):
        return self._text_tower(text.input_ids, text.attention_mask, {})

    # ------------------------------------------------------------------ pipelines
    def _forward_pipeline(self, video, ids, mask, idx):
        tape = {"vit": {}}
        vf = self._vision_tower(video, tape)
        tf = self._text_tower(ids, mask, tape)
        B, E = vf.shape
        g = self.process_group
        v_all, t_all = gather_cat(vf, g), gather_cat(tf, g)                                                 # :962-963
        ids_loc = idx.reshape(-1).contiguous()
        ids_all = gather_cat(ids_loc, g)                                                                    # :964
        WB = v_all.shape[0]
        assert WB % 8 == 0, "global batch must be a multiple of 8 for the similarity GEMMs"
        inv_temp = (1.0 / self.temp.detach().float()).reshape(1)
        sim_i2t = ops.gemm(vf, t_all, B, WB, E, out_f32=True, alpha_dev=inv_temp)                           # :966
        sim_t2i = ops.gemm(tf, v_all, B, WB, E, out_f32=True, alpha_dev=inv_temp)                           # :967
        scale = 0.5 / B                                                                                     # mean over rows, /2 (:976-978)
        l1, ds1, dt1 = ops.soft_target_ce(sim_i2t, ids_loc, ids_all, scale, B, WB)
        l2, ds2, dt2 = ops.soft_target_ce(sim_t2i, ids_loc, ids_all, scale, B, WB)
        loss = (l1.sum() + l2.sum()) * scale
        tape.update(v_all=v_all, t_all=t_all, ds1=ds1, ds2=ds2, dts=dt1.sum() + dt2.sum(), inv_temp=inv_temp, WB=WB, E=E)
        return loss, tape

    def _backward_pipeline(self, tape, grad_loss):
        B, S, E, WB = tape["B"], tape["S"], tape["E"], tape["WB"]
        D, H = self.vision_width, self.text_width
        g = self.process_group
        coef = (tape["inv_temp"] * grad_loss.reshape(1)).contiguous()
        vf, tf, v_all, t_all, ds1, ds2 = tape["vf"], tape["tf"], tape["v_all"], tape["t_all"], tape["ds1"], tape["ds2"]
        dv = ops.gemm(ds1, t_all, B, E, WB, trans_b=True, alpha_dev=coef)                                   # d sim_i2t / d v
        dt_all = ops.gemm(ds1, vf, WB, E, B, trans_a=True, trans_b=True, alpha_dev=coef)                    # d sim_i2t / d t_all
        dt = ops.gemm(ds2, v_all, B, E, WB, trans_b=True, alpha_dev=coef)
        dv_all = ops.gemm(ds2, tf, WB, E, B, trans_a=True, trans_b=True, alpha_dev=coef)
        dv = ops.add(dv, reduce_scatter_sum(dv_all, g))                                                     # all_gather backward
        dt = ops.add(dt, reduce_scatter_sum(dt_all, g))
        # d temp: sim = raw / temp  ->  dL/dtemp = -(1/temp) * sum(dsim * sim)
        grad_of(self.temp).copy_((-(tape["dts"]) * tape["inv_temp"][0] * grad_loss.reshape(())).to(torch.bfloat16))
        # text head (no backward below `pooled`: frozen decoder)
        dtp = ops.l2norm_bwd(dt, tape["tp"], tape["tn"], B, E)
        ops.colsum(dtp, B, E, out=grad_of(self.text_proj.bias))
        ops.gemm(dtp, tape["pooled"], E, H, B, trans_a=True, trans_b=True, out=grad_of(self.text_proj.weight))
        # vision head
        dvp = ops.l2norm_bwd(dv, tape["vp"], tape["vn"], B, E)
        ops.colsum(dvp, B, E, out=grad_of(self.vision_proj.bias))
        ops.gemm(dvp, tape["cls"], E, D, B, trans_a=True, trans_b=True, out=grad_of(self.vision_proj.weight))
        dcls = ops.gemm(dvp, self.vision_proj.weight, B, D, E, trans_b=True)
        if self.on_stage_grads_ready is not None:
            self.on_stage_grads_ready("head")
        demb = torch.zeros((B * S, D), dtype=torch.bfloat16, device=dcls.device)
        ops.copy_rows(dcls, demb, B, D, dmap=(1, S, 0))
        self.visual_encoder.backward_features(demb, tape["vit"])

    @torch.no_grad()
    def forward_backward(self, image, text, idx):
        """forward + backward without autograd (see DistributedGPT3_Pretrain.forward_backward): the capturable form of the step"""
        loss, tape = self._forward_pipeline(image, text.input_ids, text.attention_mask, idx)
        self._backward_pipeline(tape, torch.ones((), dtype=torch.float32, device=loss.device))
        return loss

    def forward(self, image, text, idx):
        ids, mask = text.input_ids, text.attention_mask
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            return _RetrievalStep.apply(self._anchor, self, image, ids, mask, idx)
        return self._forward_pipeline(image, ids, mask, idx)[0]


def synthetic_retrieval_model(shapes, device="cuda", num_frames=None, embed_dim=256) -> DistributedGPT3_Retrieval:
    vis = dict(img_size=shapes.img_size, patch_size=shapes.patch_size, depth=shapes.vit_depth,
               num_frames=num_frames or shapes.num_frames, embed_dim=shapes.vit_dim, num_heads=shapes.vit_heads,
               mlp_ratio=shapes.vit_mlp_ratio, clip_model=True)
    txt = GPT3Config(vocab_size=shapes.vocab, hidden_size=shapes.hidden, ffn_hidden_size=shapes.ffn,
                     num_hidden_layers=shapes.layers, num_attention_heads=shapes.heads, max_position_embeddings=shapes.max_pos,
                     layernorm_epsilon=shapes.gpt_ln_eps)
    return DistributedGPT3_Retrieval({"num_learnable_token": shapes.num_queries, "_synthetic": True, "contrastive_embed_dim": embed_dim,
                                      "num_frames": num_frames or shapes.num_frames}, visual_cfg=vis, text_cfg=txt, device=device)
