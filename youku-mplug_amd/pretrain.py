"""DistributedGPT3_Pretrain on the gfx950 kernels -- drop-in for models/distributed_gpt3.py:31-226.

Same constructor contract (config dict with visual_cfg / text_cfg / text_decoder / megatron_cfg /
freeze_* / num_learnable_token keys, or an explicit PathShapes for synthetic runs), same
forward(video, text) -> (loss_caption, loss_contrastive), same parameter names/shapes.
forward() returns a loss tensor that is attached to autograd through ONE Function whose
backward replays the hand-written backward pipeline (vision.py / gpt3.py) and writes the
parameter gradients in place, so `loss.backward()` / `engine.backward(loss)` work unchanged.
"""
from __future__ import annotations

import json
from types import SimpleNamespace
from typing import Optional

import torch
from torch import nn

from . import ops
from .gpt3 import DistributedGPT3, GPT3Config
from .vision import AttentionPool, Linear, TimeSformer, _param, convert_pretrained_vit, grad_of


class _StepFn(torch.autograd.Function):
    """Bridges the explicit forward/backward pipeline into autograd (one node per step)."""

    @staticmethod
    def forward(ctx, anchor, model, video, ids, mask, prompt_lengths=None):
        loss, tape = model._forward_pipeline(video, ids, mask, prompt_lengths=prompt_lengths)
        ctx.model, ctx.tape = model, tape
        return loss

    @staticmethod
    def backward(ctx, grad_loss):
        ctx.model._backward_pipeline(ctx.tape, grad_loss.contiguous().float())
        ctx.tape = None
        return torch.zeros(1, device=grad_loss.device), None, None, None, None, None


class DistributedGPT3_Pretrain(nn.Module):
    def __init__(self, config: Optional[dict] = None, tokenizer=None, *, visual_cfg: Optional[dict] = None,
                 text_cfg: Optional[GPT3Config] = None, device="cuda"):
        super().__init__()
        config = dict(config or {})
        self.tokenizer = tokenizer
        if visual_cfg is None:
            visual_cfg = json.load(open(config["visual_cfg"], "r"))                       # :36
        if text_cfg is None:
            text_cfg = GPT3Config.from_json_file(config["text_cfg"])                      # :37
        self.visual_encoder = self._build_visual_encoder(config, visual_cfg, device)
        self._init_visual_encoder_from_ckpt(visual_cfg)                                   # :56-72
        if config.get("text_decoder") and not config.get("_synthetic", False):
            self.text_decoder = DistributedGPT3(model_dir=config["text_decoder"], device=device)          # :78-84
        else:
            self.text_decoder = DistributedGPT3(config=text_cfg, device=device)
        if config.get("freeze_vit", False):                                               # :86-89
            for name, p in self.visual_encoder.named_parameters():
                if not any(x in name for x in ("time", "temporal")):
                    p.requires_grad = False
        if config.get("freeze_text_decoder", True):                                       # :91-93 (every shipped recipe freezes it)
            for p in self.text_decoder.parameters():
                p.requires_grad = False
        self.vision_width = visual_cfg["embed_dim"]
        self.text_width = self.text_decoder.config.hidden_size
        self.num_learnable_token = config.get("num_learnable_token", 256)                 # :102
        self.learnable_queries = _param(1, self.num_learnable_token, self.vision_width, std=0.015, device=device)
        self.attn_pool = AttentionPool(self.vision_width, num_heads=visual_cfg["num_heads"], mlp_ratio=visual_cfg["mlp_ratio"],
                                       eps=1e-6, std=0.02, device=device)                 # :106-109
        self.visual_fc = Linear(self.vision_width, self.text_width, std=0.015, device=device)   # :111,116
        self.visual_norm = nn.Identity()
        if visual_cfg.get("connect_ln", False):                                           # :112-115
            from .vision import LayerNormWithForceFP32
            self.visual_norm = LayerNormWithForceFP32(self.text_width, eps=1e-6, device=device)
        self.use_contrastive = config.get("use_contrastive", False)
        if self.use_contrastive:
            raise NotImplementedError("use_contrastive is false in the pre-train recipe (…yaml:26); ITC lives in the retrieval model")
        self.prompt = config.get("prompt", "")
        self._anchor = torch.zeros(1, device=device, requires_grad=True)
        self.on_stage_grads_ready = None      # engine hook(stage_name)

    def _build_visual_encoder(self, config, visual_cfg, device):
        return TimeSformer(
            img_size=visual_cfg["img_size"], num_frames=visual_cfg["num_frames"], patch_size=visual_cfg["patch_size"],
            embed_dim=visual_cfg["embed_dim"], depth=visual_cfg["depth"], num_heads=visual_cfg["num_heads"],
            mlp_ratio=visual_cfg["mlp_ratio"], eps=1e-6, init_std=0.015, clip_model=visual_cfg.get("clip_model", False),
            device=device)                                                                # :39-54

    def _init_visual_encoder_from_ckpt(self, visual_cfg):
        """models/distributed_gpt3.py:56-72: `pretrained_ckpt` of the visual config ("clip/<path>.pth" in every shipped
        config, configs/models/clip-b16.json:2) initialises the vision tower, strict=False; a missing file raises as
        torch.load does in the reference.  `timm/<name>` needs the timm package and network access: refused loudly."""
        pretrained = visual_cfg.get("pretrained_ckpt", None)
        if pretrained is None:
            return None
        if pretrained.startswith("timm"):
            raise NotImplementedError("pretrained_ckpt 'timm/...' needs the timm model zoo (not available here); "
                                      "the shipped configs use 'clip/<file>.pth'")
        if not pretrained.startswith("clip"):
            return None
        path = "/".join(pretrained.split("/")[1:])
        weights = convert_pretrained_vit(torch.load(path, map_location="cpu"))
        own = self.visual_encoder.state_dict()
        weights = {k: (v.to(own[k].dtype) if k in own else v) for k, v in weights.items()}
        msg = self.visual_encoder.load_state_dict(weights, strict=False)
        print("Initialize Vision Encoder from CKPT {}".format(path))
        print(msg)
        return msg

    def no_weight_decay(self):
        return {"visual_encoder.pos_embed", "visual_encoder.cls_token", "visual_encoder.temporal_embed"}       # :224-226

    # ---------------------------------------------------------------- pipeline
    def _forward_pipeline(self, video, ids, mask, want_logits=False, prompt_lengths=None):
        B = video.shape[0]
        Q, Hh = self.num_learnable_token, self.text_width
        tape = {"vit": {}, "pool": {}, "gpt": {}}
        emb = self.visual_encoder.forward_features(video.to(torch.bfloat16), tape["vit"])
        S_img = emb.shape[0] // B
        image_query = self.attn_pool.forward_pool(self.learnable_queries, emb, B, S_img, tape["pool"])      # :134
        qf = ops.gemm(image_query, self.visual_fc.weight, B * Q, Hh, self.vision_width, bias=self.visual_fc.bias)   # :136
        tape["image_query"] = image_query
        if not isinstance(self.visual_norm, nn.Identity):                                 # connect_ln: visual_norm(visual_fc(.))
            vn = self.visual_norm
            tape["qf_raw"] = qf
            qf, tape["vn_mean"], tape["vn_rstd"] = ops.layernorm_fwd(qf, vn.weight, vn.bias, vn.eps, B * Q, Hh)
        if not want_logits and Q > 0 and ids.shape[1] > 0:
            # training / loss-only evaluation: labels and loss weights of the L text positions from one kernel (the Q query slots
            # in front never carry loss, :142-159): LM head + CE run on that window and no [B, S] target tensors are built
            pl = None if prompt_lengths is None else torch.as_tensor(prompt_lengths, device=ids.device, dtype=torch.long).contiguous()
            out = self.text_decoder.forward_lm(qf, ids, None, None, tape["gpt"], loss_window=(Q, ids.shape[1]),
                                               window_targets=ops.caption_targets(ids, mask, pl))
            tape["out"] = out
            return out["loss"], tape
        # targets / loss mask exactly as :142-159 (filler id 100 is always masked)
        targets = torch.cat([ids[:, 1:], ids[:, 1:2]], dim=1)
        targets = torch.cat([torch.full((B, Q), 100, dtype=torch.long, device=ids.device), targets], dim=1)
        tla = mask[:, 1:]
        if prompt_lengths is not None:                                                    # models/distributed_gpt3.py:348-351
            pl = torch.as_tensor(prompt_lengths, device=ids.device).view(-1, 1)
            tla = tla.clone()
            tla[torch.arange(tla.shape[1], device=ids.device)[None] < pl] = 0
        loss_mask = torch.cat([torch.zeros((B, Q), dtype=torch.long, device=ids.device), tla], dim=1)
        # the Q query slots in front never carry loss (zeros above): LM head + CE on the L text positions only
        out = self.text_decoder.forward_lm(qf, ids, targets, loss_mask, tape["gpt"], want_logits=want_logits,
                                           loss_window=(Q, ids.shape[1]))
        tape["out"] = out
        return out["loss"], tape

    def _backward_pipeline(self, tape, grad_loss):
        B = tape["vit"]["B"]
        Q, Hh, D = self.num_learnable_token, self.text_width, self.vision_width
        dqf = self.text_decoder.backward_lm(tape["gpt"], grad_loss)
        iq = tape["image_query"]
        if "qf_raw" in tape:
            vn = self.visual_norm
            dqf = ops.layernorm_bwd(dqf, tape["qf_raw"], vn.weight, tape["vn_mean"], tape["vn_rstd"], B * Q, Hh,
                                    dgamma=grad_of(vn.weight), dbeta=grad_of(vn.bias))
        ops.colsum(dqf, B * Q, Hh, out=grad_of(self.visual_fc.bias))
        ops.gemm(dqf, iq, Hh, D, B * Q, trans_a=True, trans_b=True, out=grad_of(self.visual_fc.weight))
        diq = ops.gemm(dqf, self.visual_fc.weight, B * Q, D, Hh, trans_b=True)
        demb = self.attn_pool.backward_pool(diq, self.learnable_queries, tape["pool"])
        if self.on_stage_grads_ready is not None:
            self.on_stage_grads_ready("head")
        self.visual_encoder.backward_features(demb, tape["vit"])

    # ---------------------------------------------------------------- reference-shaped API
    def forward(self, image, text):
        ids, mask = text.input_ids, text.attention_mask
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            loss = _StepFn.apply(self._anchor, self, image, ids, mask)
        else:
            loss, _ = self._forward_pipeline(image, ids, mask)
        return loss, torch.zeros((), device=loss.device)                                   # :218-221

    @torch.no_grad()
    def forward_backward(self, image, text):
        """forward + backward of one micro-batch WITHOUT autograd: what `loss, _ = model(image, text); loss.backward()` does, as two
        direct calls of the explicit pipeline.  Gradients land in the same .grad views.  This is the form a HIP graph can capture
        (engine.graph_step): under autograd the anchor's AccumulateGrad node runs on the stream the anchor was created on, not
        on the capturing stream."""
        loss, tape = self._forward_pipeline(image, text.input_ids, text.attention_mask, prompt_lengths=getattr(text, "prompt_lengths", None))
        self._backward_pipeline(tape, torch.ones((), dtype=torch.float32, device=loss.device))
        return loss

    @torch.no_grad()
    def forward_outputs(self, image, text):
        """Evaluation helper: full decoder outputs (logits, losses, last_hidden_state, loss)."""
        _, tape = self._forward_pipeline(image, text.input_ids, text.attention_mask, want_logits=True)
        return SimpleNamespace(**tape["out"])


class DistributedGPT3_Pretrain_Image(DistributedGPT3_Pretrain):
    """models/distributed_gpt3.py:229-427 with use_eva_g: the EVA-ViT-g image encoder in front of the same abstractor and
    frozen decoder; forward(image [B,3,H,W], text) -> (loss_caption, loss_contrastive); `text.prompt_lengths`, when
    present, is masked out of the caption loss (:348-351)."""

    def _build_visual_encoder(self, config, visual_cfg, device):
        from .eva_vit import create_eva_vit_g
        if not config.get("use_eva_g", False):
            raise NotImplementedError("the image path is built for use_eva_g: true (models/distributed_gpt3.py:256-261)")
        over = {k: visual_cfg[k] for k in ("embed_dim", "depth", "num_heads", "mlp_ratio", "patch_size") if k in visual_cfg
                and config.get("_synthetic", False)}
        return create_eva_vit_g(img_size=visual_cfg["img_size"], drop_path_rate=visual_cfg.get("drop_path", False) or 0.0,
                                device=device, **over)

    def no_weight_decay(self):
        return {"visual_encoder.pos_embed", "visual_encoder.cls_token"}                      # :425-427

    def forward(self, image, text):
        ids, mask = text.input_ids, text.attention_mask
        pl = getattr(text, "prompt_lengths", None)
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            loss = _StepFn.apply(self._anchor, self, image, ids, mask, pl)
        else:
            loss, _ = self._forward_pipeline(image, ids, mask, prompt_lengths=pl)
        return loss, torch.zeros((), device=loss.device)


def synthetic_image_model(shapes, device="cuda", **eva) -> "DistributedGPT3_Pretrain_Image":
    """Random-init image model; `eva` overrides (embed_dim, depth, num_heads, mlp_ratio, patch_size) for reduced shapes."""
    vis = dict(img_size=shapes.img_size, num_frames=1, embed_dim=eva.get("embed_dim", 1408), num_heads=eva.get("num_heads", 16),
               mlp_ratio=eva.get("mlp_ratio", 4.3637), depth=eva.get("depth", 40), patch_size=eva.get("patch_size", 14))
    txt = GPT3Config(vocab_size=shapes.vocab, hidden_size=shapes.hidden, ffn_hidden_size=shapes.ffn,
                     num_hidden_layers=shapes.layers, num_attention_heads=shapes.heads, max_position_embeddings=shapes.max_pos,
                     layernorm_epsilon=shapes.gpt_ln_eps)
    return DistributedGPT3_Pretrain_Image({"num_learnable_token": shapes.num_queries, "_synthetic": True, "use_eva_g": True},
                                          visual_cfg=vis, text_cfg=txt, device=device)


def synthetic_model(shapes, device="cuda", num_frames=None) -> DistributedGPT3_Pretrain:
    """Random-init model from a shapes object with the fields of oracle.weights.PathConfig (the
    product never imports oracle/: tests pass the dataclass in)."""
    vis = dict(img_size=shapes.img_size, patch_size=shapes.patch_size, depth=shapes.vit_depth,
               num_frames=num_frames or shapes.num_frames, embed_dim=shapes.vit_dim, num_heads=shapes.vit_heads,
               mlp_ratio=shapes.vit_mlp_ratio, clip_model=True, connect_ln=bool(getattr(shapes, "connect_ln", False)))
    txt = GPT3Config(vocab_size=shapes.vocab, hidden_size=shapes.hidden, ffn_hidden_size=shapes.ffn,
                     num_hidden_layers=shapes.layers, num_attention_heads=shapes.heads, max_position_embeddings=shapes.max_pos,
                     layernorm_epsilon=shapes.gpt_ln_eps)
    return DistributedGPT3_Pretrain({"num_learnable_token": shapes.num_queries, "_synthetic": True}, visual_cfg=vis,
                                    text_cfg=txt, device=device)
