"""Generation + classification heads on the gfx950 kernels -- drop-ins for
models/distributed_gpt3.py:431-657 (DistributedGPT3_Cls) and :988-1218 (DistributedGPT3_Retrieval_Cls, the ITM
re-ranker of downstream/run_retrieval_distributed_gpt3_itm.py).

Both run the frozen decoder twice per step on the same visual prefix:
  * generation pass: caption cross-entropy with a per-sample prompt-length loss mask (:1097-1102, 1121-1127);
  * prompt pass (use_cls): the last valid hidden state goes through cls_head = Linear-ReLU-Linear and a
    cross-entropy against `labels` (:1129-1153).
The ITM model first appends query_features[negative_indices] (:1105-1108), so text/prompt/labels carry
B + len(negative_indices) rows.  forward(..., train=False) returns the scores the eval loops rank by
(:1156-1212 / :596-651).  Same constructor contract and parameter names as the reference; forward returns
(loss_caption, loss_cls) attached to autograd through one Function (as pretrain.py does).
"""
from __future__ import annotations

from typing import Optional, Sequence

import torch
from torch import nn

from . import ops
from .ops import ACT_RELU
from .pretrain import DistributedGPT3_Pretrain
from .gpt3 import GPT3Config
from .vision import Linear, grad_of

_PAD = 8   # GEMM / cross-entropy column granule: the class dimension is padded to it


class _GenClsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, model, args):
        (lc, lk), tape = model._forward_pipeline(*args)
        ctx.model, ctx.tape = model, tape
        return lc, lk

    @staticmethod
    def backward(ctx, g_caption, g_cls):
        ctx.model._backward_pipeline(ctx.tape, g_caption.contiguous().float(), g_cls.contiguous().float())
        ctx.tape = None
        return torch.zeros(1, device=g_caption.device), None, None


class _GenCls(DistributedGPT3_Pretrain):
    ITM = False

    def __init__(self, config: Optional[dict] = None, tokenizer=None, *, visual_cfg: Optional[dict] = None,
                 text_cfg: Optional[GPT3Config] = None, device="cuda"):
        config = dict(config or {})
        if visual_cfg is not None and "num_frames" in config:          # TimeSformer(num_frames=config['num_frames']) (:441, :998)
            visual_cfg = dict(visual_cfg, num_frames=config["num_frames"])
        # The pre-train class implements both options below; the generation / classification pipelines of this file do not
        # (their hidden-only prompt pass records no decoder weight-gradient inputs, and _query_features has no visual_norm
        # stage), and no shipped recipe sets either: refuse loudly instead of training silently wrong.
        if not config.get("freeze_text_decoder", True):
            raise NotImplementedError("freeze_text_decoder: false (models/distributed_gpt3.py:492, 722, 1049) is implemented for "
                                      "DistributedGPT3_Pretrain only; the ITM / classification / caption models refuse it")
        super().__init__(config, tokenizer, visual_cfg=visual_cfg, text_cfg=text_cfg, device=device)
        if not isinstance(self.visual_norm, nn.Identity):
            raise NotImplementedError("connect_ln (models/distributed_gpt3.py:513-516, 1070-1073) is implemented for "
                                      "DistributedGPT3_Pretrain only; the ITM / classification / caption models refuse it")
        self.use_cls = config.get("use_cls", False)                                         # :523 / :1079
        if self.use_cls:
            H = self.text_width
            self.num_classes = config["num_classes"]
            self.cls_head = nn.Sequential(Linear(H, H, device=device), nn.ReLU(), Linear(H, self.num_classes, device=device))

    # ------------------------------------------------------------------ pieces
    def _query_features(self, video, tape):
        B = video.shape[0]
        Q, Hh = self.num_learnable_token, self.text_width
        emb = self.visual_encoder.forward_features(video.to(torch.bfloat16), tape["vit"])
        image_query = self.attn_pool.forward_pool(self.learnable_queries, emb, B, emb.shape[0] // B, tape["pool"])
        tape["image_query"] = image_query
        return ops.gemm(image_query, self.visual_fc.weight, B * Q, Hh, self.vision_width, bias=self.visual_fc.bias)

    def _expand(self, qf, src: Sequence[int]):
        """rows of sample src[j] for every output sample j (negatives / per-text repeats)."""
        Q, Hh = self.num_learnable_token, self.text_width
        s = torch.as_tensor(list(src), dtype=torch.long, device=qf.device)
        idx = (s[:, None] * Q + torch.arange(Q, device=qf.device)[None]).reshape(-1).contiguous()
        return ops.gather_rows(qf, idx, idx.numel(), Hh)

    @staticmethod
    def _gen_targets(ids, mask, prompt_lengths, Q):
        n = ids.shape[0]
        targets = torch.cat([ids[:, 1:], ids[:, 1:2]], dim=1)                                 # :1097-1098 (last column unused)
        tla = mask[:, 1:].clone()
        pl = torch.as_tensor(prompt_lengths, device=ids.device).view(-1, 1)
        tla[torch.arange(tla.shape[1], device=ids.device)[None] < pl] = 0                     # :1100-1102
        targets = torch.cat([torch.full((n, Q), 100, dtype=torch.long, device=ids.device), targets], dim=1)
        loss_mask = torch.cat([torch.zeros((n, Q), dtype=torch.long, device=ids.device), tla], dim=1)
        return targets, loss_mask

    def _cls_logits(self, qf_rows, p_ids, p_mask, tape):
        """prompt pass -> pooled last valid hidden state -> cls_head logits (class dim padded to 8 with -1e30 bias)."""
        n, Lp = p_ids.shape
        Q, H, C = self.num_learnable_token, self.text_width, self.num_classes
        S = Q + Lp
        out = self.text_decoder.forward_lm(qf_rows, p_ids, None, None, tape["gpt2"], hidden_only=True, pass_index=1)
        rows = (torch.arange(n, device=p_ids.device) * S + Q + p_mask.sum(dim=-1) - 1).contiguous()      # :1149-1150
        pooled = ops.gather_rows(out["last_hidden_state"].reshape(n * S, H), rows, n, H)
        l0, l2 = self.cls_head[0], self.cls_head[2]
        z = torch.empty((n, H), dtype=torch.bfloat16, device=pooled.device)
        a = ops.gemm(pooled, l0.weight, n, H, H, bias=l0.bias, act=ACT_RELU, preact_out=z)
        Cp = (C + _PAD - 1) // _PAD * _PAD
        w2 = torch.zeros((Cp, H), dtype=torch.bfloat16, device=a.device)
        b2 = torch.full((Cp,), -1e30, dtype=torch.bfloat16, device=a.device)
        w2[:C].copy_(l2.weight.detach())
        b2[:C].copy_(l2.bias.detach())
        logits = ops.gemm(a, w2, n, Cp, H, bias=b2)
        tape.update(rows=rows, pooled=pooled, z=z, a=a, w2=w2, Cp=Cp, n2=n, S2=S)
        return logits

    # ------------------------------------------------------------------ pipelines
    def _forward_pipeline(self, video, ids, mask, prompt_lengths, p_ids, p_mask, negative_indices, labels):
        Bv = video.shape[0]
        Q = self.num_learnable_token
        tape = {"vit": {}, "pool": {}, "gpt": {}, "gpt2": {}}
        qf = self._query_features(video, tape)
        src = list(range(Bv)) + ([int(i) for i in negative_indices] if self.ITM else [])
        qf_rows = self._expand(qf, src) if self.ITM else qf                                   # :1105-1108
        targets, loss_mask = self._gen_targets(ids, mask, prompt_lengths, Q)
        out = self.text_decoder.forward_lm(qf_rows, ids, targets, loss_mask, tape["gpt"])
        loss_caption = out["loss"]
        tape.update(src=src, Bv=Bv)
        if self.use_cls:
            logits = self._cls_logits(qf_rows, p_ids, p_mask, tape)
            n = len(src)
            w = torch.full((n,), 1.0 / n, dtype=torch.float32, device=logits.device)         # F.cross_entropy mean (:1153)
            dlog = torch.empty_like(logits)
            _, loss_cls = ops.cross_entropy(logits, labels.contiguous().view(-1), w, n, tape["Cp"], dlogits=dlog)
            tape["dlog"] = dlog
        else:
            loss_cls = torch.zeros((), dtype=torch.float32, device=loss_caption.device)      # :1155
        return (loss_caption, loss_cls), tape

    def _backward_pipeline(self, tape, g_caption, g_cls):
        Bv, src = tape["Bv"], tape["src"]
        Q, Hh, D = self.num_learnable_token, self.text_width, self.vision_width
        dq_rows = self.text_decoder.backward_lm(tape["gpt"], g_caption)
        if self.use_cls:
            n, S, Cp, C = tape["n2"], tape["S2"], tape["Cp"], self.num_classes
            l0, l2 = self.cls_head[0], self.cls_head[2]
            dlog, a, z, pooled = tape["dlog"], tape["a"], tape["z"], tape["pooled"]
            gscale = g_cls.reshape(1)
            gw2 = ops.gemm(dlog, a, Cp, Hh, n, trans_a=True, trans_b=True, alpha_dev=gscale)
            gb2 = ops.colsum(dlog, n, Cp)
            grad_of(l2.weight).copy_(gw2[:C])
            grad_of(l2.bias).copy_((gb2[:C].float() * g_cls).to(torch.bfloat16))
            dz = ops.gemm(dlog, tape["w2"], n, Hh, Cp, trans_b=True, alpha_dev=gscale, act_bwd_z=z, act_bwd=ACT_RELU)
            ops.gemm(dz, pooled, Hh, Hh, n, trans_a=True, trans_b=True, out=grad_of(l0.weight))
            ops.colsum(dz, n, Hh, out=grad_of(l0.bias))
            dpooled = ops.gemm(dz, l0.weight, n, Hh, Hh, trans_b=True)
            dlh = torch.zeros((n * S, Hh), dtype=torch.bfloat16, device=dz.device)
            ops.scatter_rows(dpooled, tape["rows"], dlh, n, Hh)
            dq2 = self.text_decoder.backward_lm(tape["gpt2"], None, d_last_hidden=dlh)
            dq_rows = ops.add(dq_rows, dq2)
        # fold the expanded rows back onto the videos they came from, in row order (reproducible)
        per = Q * Hh
        dqf = dq_rows[:Bv * Q]
        if len(src) > Bv:
            dqf = dqf.clone()
            flat_all, flat = dq_rows.view(-1), dqf.view(-1)
            for j in range(Bv, len(src)):
                b = src[j]
                ops.add(flat[b * per:(b + 1) * per], flat_all[j * per:(j + 1) * per], out=flat[b * per:(b + 1) * per])
        iq = tape["image_query"]
        ops.colsum(dqf, Bv * Q, Hh, out=grad_of(self.visual_fc.bias))
        ops.gemm(dqf, iq, Hh, D, Bv * Q, trans_a=True, trans_b=True, out=grad_of(self.visual_fc.weight))
        diq = ops.gemm(dqf, self.visual_fc.weight, Bv * Q, D, Hh, trans_b=True)
        demb = self.attn_pool.backward_pool(diq, self.learnable_queries, tape["pool"])
        if self.on_stage_grads_ready is not None:
            self.on_stage_grads_ready("head")
        self.visual_encoder.backward_features(demb, tape["vit"])

    @torch.no_grad()
    def _scores(self, video, text, prompt_text):
        """train=False: generation score -sum(losses * loss_mask) per (video, text) and the cls_head score."""
        Bv = video.shape[0]
        Q = self.num_learnable_token
        ids, mask = text.input_ids, text.attention_mask
        t = ids.shape[0] // Bv
        tape = {"vit": {}, "pool": {}, "gpt": {}, "gpt2": {}}
        qf = self._query_features(video, tape)
        rep = [b for b in range(Bv) for _ in range(t)]                                        # '(v t)' / '(b c)' ordering
        qf_rep = self._expand(qf, rep)
        targets, loss_mask = self._gen_targets(ids, mask, text.prompt_lengths, Q)
        out = self.text_decoder.forward_lm(qf_rep, ids, targets, loss_mask, tape["gpt"])
        gen = (-(out["losses"] * loss_mask).sum(dim=-1)).view(Bv, t)                           # :1180-1181 / :618-619
        cls = None
        if self.use_cls:
            per_video = not self.ITM                                                           # :622-647 vs :1183-1208
            logits = self._cls_logits(qf if per_video else qf_rep, prompt_text.input_ids, prompt_text.attention_mask, tape)
            logits = logits[:, :self.num_classes].float()
            cls = torch.softmax(logits, dim=-1)[:, 1].view(Bv, t) if self.ITM else logits
        return (gen if self.ITM else torch.softmax(gen, dim=-1)), cls

    def _run(self, image, text, prompt_text, negative_indices, labels, train):
        if not train:
            return self._scores(image, text, prompt_text)
        args = (image, text.input_ids, text.attention_mask, text.prompt_lengths,
                None if prompt_text is None else prompt_text.input_ids, None if prompt_text is None else prompt_text.attention_mask,
                negative_indices, labels)
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            return _GenClsFn.apply(self._anchor, self, args)
        return self._forward_pipeline(*args)[0]


class DistributedGPT3_Cls(_GenCls):
    """models/distributed_gpt3.py:431-657."""
    ITM = False

    def forward(self, image, text=None, prompt_text=None, labels=None, train=True):
        return self._run(image, text, prompt_text, None, labels, train)


class DistributedGPT3_Retrieval_Cls(_GenCls):
    """models/distributed_gpt3.py:988-1218."""
    ITM = True

    def forward(self, image, text=None, prompt_text=None, negative_indices=None, labels=None, train=True):
        return self._run(image, text, prompt_text, negative_indices, labels, train)


class DistributedGPT3_Caption(_GenCls):
    """models/distributed_gpt3.py:661-814: caption fine-tuning loss (prompt tokens masked out) and beam-search
    captioning on the KV-cache decode path (generation.py)."""
    ITM = False

    def forward(self, image, text=None):
        return self._run(image, text, None, None, None, True)[0]                              # :768-789 returns loss_caption only

    @torch.no_grad()
    def generate(self, image, text, termination_id=None, **kw):
        """One beam search per sample (:790-809): prompt = text.input_ids[i, :mask.sum()-1], prefix = its query features."""
        Bv = image.shape[0]
        Q, Hh = self.num_learnable_token, self.text_width
        was = self.training
        self.eval()
        try:
            qf = self._query_features(image, {"vit": {}, "pool": {}}).view(Bv, Q, Hh)
            if termination_id is None:
                tok = getattr(self, "tokenizer", None)
                termination_id = tok.tokenizer.eos if tok is not None else None
            res = []
            for i in range(text.input_ids.shape[0]):
                out = self.text_decoder.generate(text.input_ids[i:i + 1], query_embeds=qf[i:i + 1], termination_id=termination_id,
                                                 do_sample=False, prompt_length=int(text.attention_mask.sum(-1)[i]) - 1, **kw)
                res.append(out.sequences.cpu())
            return res
        finally:
            self.train(was)


def synthetic_gencls_model(shapes, kind: str, num_classes: int = 2, device="cuda", num_frames=None):
    """Random-init ITM ("itm") or classification ("cls") model from a PathConfig-like shapes object."""
    vis = dict(img_size=shapes.img_size, patch_size=shapes.patch_size, depth=shapes.vit_depth,
               num_frames=num_frames or shapes.num_frames, embed_dim=shapes.vit_dim, num_heads=shapes.vit_heads,
               mlp_ratio=shapes.vit_mlp_ratio, clip_model=True)
    txt = GPT3Config(vocab_size=shapes.vocab, hidden_size=shapes.hidden, ffn_hidden_size=shapes.ffn,
                     num_hidden_layers=shapes.layers, num_attention_heads=shapes.heads, max_position_embeddings=shapes.max_pos,
                     layernorm_epsilon=shapes.gpt_ln_eps)
    klass = {"itm": DistributedGPT3_Retrieval_Cls, "cls": DistributedGPT3_Cls, "caption": DistributedGPT3_Caption}[kind]
    return klass({"num_learnable_token": shapes.num_queries, "_synthetic": True, "use_cls": kind != "caption", "num_classes": num_classes},
                 visual_cfg=vis, text_cfg=txt, device=device)
