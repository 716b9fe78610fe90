"""Frozen GPT-3 causal LM (Megatron-style decoder) on the gfx950 kernels.

Mirrors models/modeling_distributed_gpt3.py: GPT3Config :459-547, GPT3Embedding :598-666,
GPT3ParallelAttention :820-938 (+ GPT3CoreAttention :689-817), GPT3ParallelMLP :550-595,
GPT3ParallelTransformerLayer :982-1089, GPT3ParallelTransformer :1092-1186, GPT3Model
:1272-1366, DistributedGPT3 :1522-1618 -- same attribute paths and state-dict keys
(text_decoder.dist_model.language_model.{embedding,encoder}...), TP=1/PP=1.

Rows are kept batch-major [B*S, H] (the reference's [s,b,h] transpose :653 is a pure
permutation of rows; every op here is row-wise or addressed by strides).  The decoder is
frozen in the pre-train recipe (models/distributed_gpt3.py:91-93), so backward computes
dgrad only -- no weight gradients through 24 layers (the reference computes and discards
them).  Dropout (hidden 0.1, attention-prob 0.1) is live in train() mode even though the
weights are frozen and is regenerated from (seed, offset) in backward.
"""
from __future__ import annotations

import json
import math
import os
from typing import Optional

import torch
from torch import nn

from . import ops
from .vision import grad_of
from .ops import ACT_GELU_TANH
from .vision import Linear, _param


SEED_PASS_STRIDE = 0x51ED2705      # seed of decoder pass k of a step = step_seed + k * SEED_PASS_STRIDE

class GPT3Config:
    """Field names of configs/models/config_gpt3_*.json; every GPT3Config default (:463-496) overridable."""

    def __init__(self, vocab_size=25600, hidden_size=768, ffn_hidden_size=None, num_hidden_layers=12, num_attention_heads=12,
                 max_position_embeddings=2048, layernorm_epsilon=1e-12, hidden_dropout=0.1, attention_dropout=0.1,
                 init_method_std=0.02, bias_gelu_fusion=True, apply_query_key_layer_scaling=True, **kw):
        self.vocab_size, self.hidden_size = vocab_size, hidden_size
        self.ffn_hidden_size = ffn_hidden_size or 4 * hidden_size
        self.num_hidden_layers, self.num_attention_heads = num_hidden_layers, num_attention_heads
        self.max_position_embeddings = max_position_embeddings
        self.layernorm_epsilon = layernorm_epsilon
        self.hidden_dropout, self.attention_dropout = hidden_dropout, attention_dropout
        self.init_method_std = init_method_std
        self.bias_gelu_fusion = bias_gelu_fusion
        self.apply_query_key_layer_scaling = apply_query_key_layer_scaling
        self.kv_channels = hidden_size // num_attention_heads
        self.extra = kw

    @classmethod
    def from_json_file(cls, path):
        with open(path) as f:
            return cls(**json.load(f))

    @classmethod
    def from_pretrained(cls, model_dir):
        import os
        return cls.from_json_file(os.path.join(model_dir, "config.json"))


class _LN(nn.Module):
    def __init__(self, dim, eps, device=None):
        super().__init__()
        self.eps = eps
        self.weight = _param(dim, const=1.0, device=device)
        self.bias = _param(dim, const=0.0, device=device)


class _WordEmbeddings(nn.Module):
    """mpu.VocabParallelEmbedding at TP=1 (:619).  Callable like the reference's
    (models/distributed_gpt3.py:155) -- the gather runs on the HIP embedding kernel."""

    def __init__(self, vocab, hidden, std, device=None):
        super().__init__()
        self.weight = _param(vocab, hidden, std=std, device=device)
        self._zero_pos = None

    def forward(self, ids: torch.Tensor):
        B, L = ids.shape
        H = self.weight.shape[1]
        if self._zero_pos is None or self._zero_pos.shape[0] < L:
            self._zero_pos = torch.zeros((L, H), dtype=torch.bfloat16, device=self.weight.device)
        return ops.gpt_embed_fwd(None, ids.contiguous(), self.weight, self._zero_pos, B, 0, L, H).view(B, L, H)


class GPT3Embedding(nn.Module):
    def __init__(self, cfg: GPT3Config, device=None):
        super().__init__()
        self.word_embeddings = _WordEmbeddings(cfg.vocab_size, cfg.hidden_size, cfg.init_method_std, device)
        self.position_embeddings = nn.Module()
        self.position_embeddings.weight = _param(cfg.max_position_embeddings, cfg.hidden_size, std=cfg.init_method_std, device=device)


class GPT3ParallelAttention(nn.Module):
    def __init__(self, cfg, layer_number, std, out_std, device=None):
        super().__init__()
        self.layer_number = max(1, layer_number)
        H = cfg.hidden_size
        self.query_key_value = Linear(H, 3 * H, std=std, device=device)     # rows head-major [h0:q,k,v | h1:q,k,v ...] (:895-902)
        self.dense = Linear(H, H, std=out_std, device=device)


class GPT3ParallelMLP(nn.Module):
    def __init__(self, cfg, std, out_std, device=None):
        super().__init__()
        self.dense_h_to_4h = Linear(cfg.hidden_size, cfg.ffn_hidden_size, std=std, device=device)
        self.dense_4h_to_h = Linear(cfg.ffn_hidden_size, cfg.hidden_size, std=out_std, device=device)


class GPT3ParallelTransformerLayer(nn.Module):
    def __init__(self, cfg, layer_number, std, out_std, device=None):
        super().__init__()
        self.layer_number = layer_number
        self.input_layernorm = _LN(cfg.hidden_size, cfg.layernorm_epsilon, device)
        self.self_attention = GPT3ParallelAttention(cfg, layer_number, std, out_std, device)
        self.post_attention_layernorm = _LN(cfg.hidden_size, cfg.layernorm_epsilon, device)
        self.mlp = GPT3ParallelMLP(cfg, std, out_std, device)


class GPT3ParallelTransformer(nn.Module):
    def __init__(self, cfg, device=None):
        super().__init__()
        std = cfg.init_method_std
        out_std = std / math.sqrt(2.0 * cfg.num_hidden_layers)
        self.layers = nn.ModuleList([GPT3ParallelTransformerLayer(cfg, i + 1, std, out_std, device)
                                     for i in range(cfg.num_hidden_layers)])
        self.final_layernorm = _LN(cfg.hidden_size, cfg.layernorm_epsilon, device)


class GPT3TransformerLanguageModel(nn.Module):
    def __init__(self, cfg, device=None):
        super().__init__()
        self.embedding = GPT3Embedding(cfg, device)
        self.encoder = GPT3ParallelTransformer(cfg, device)


class GPT3Model(nn.Module):
    def __init__(self, cfg: GPT3Config, device=None):
        super().__init__()
        self.config = cfg
        self.language_model = GPT3TransformerLanguageModel(cfg, device)

    def word_embeddings_weight(self):
        return self.language_model.embedding.word_embeddings.weight


_SITE_EMBED, _SITE_ATTN, _SITE_DROP1, _SITE_DROP2 = 0, 1, 2, 3

# The residual stream of the decoder is kept in fp32 (ops.ln_stream_fwd: the add of a sublayer's output happens inside the
# LayerNorm that follows it, in fp32, instead of in the GEMM's bf16 residual epilogue).  The reference adds in bf16
# (models/modeling_distributed_gpt3.py:1059-1078); 48 such roundings are what put a 24-layer bf16 run 1.2e-2 from the fp32
# function in the logits (tools/parity_bisect.py: 0.87e-2 with the fp32 stream).  MPV_DECODER_STREAM=bf16 (measurement knob)
# restores the bf16 stream with the residual adds in the GEMM epilogues.
FP32_STREAM = os.environ.get("MPV_DECODER_STREAM", "fp32") != "bf16"
# Round 4: in the fp32-stream form the dropout of a sublayer's output (bias_dropout_add, models/modeling_distributed_gpt3.py:953-979) is
# applied by the LayerNorm that adds it into the stream (mpv_ln_stream_fwd_drop: same element index, threshold and bf16 rounding, so the
# step is bit-identical) instead of by the sublayer GEMM's epilogue.  MPV_DROPOUT_IN_LN=0 (measurement knob) keeps it in the GEMM.
DROPOUT_IN_LN = os.environ.get("MPV_DROPOUT_IN_LN", "1") != "0"


def _offset(layer: int, site: int) -> int:
    return (layer * 4 + site) << 36


class DistributedGPT3(nn.Module):
    """Drop-in for models/modeling_distributed_gpt3.py:1522 at TP=1 (checkpoint loading of
    `<model_dir>/model/mp_rank_00_model_states.pt['module']` :431-441 is kept)."""

    def __init__(self, model_dir=None, rank=0, path_load_tag="model", config: Optional[GPT3Config] = None, device=None,
                 load_state_dict=True, **kwargs):
        super().__init__()
        import os
        self.config = config if config is not None else GPT3Config.from_pretrained(model_dir)
        assert self.config.kv_channels in (64, 80, 96), "fused attention kernels are built for head_dim 64/80/96"
        self.dist_model = GPT3Model(self.config, device=device)
        if config is None and load_state_dict and model_dir is not None:
            path = os.path.join(model_dir, str(path_load_tag), "mp_rank_00_model_states.pt")
            if os.path.isfile(path):
                sd = torch.load(path, map_location="cpu")["module"]
                self.dist_model.load_state_dict({k: v.to(torch.bfloat16) for k, v in sd.items()})
        self.inference_params = None
        self.step_seed = 0        # bumped by the engine every step -> fresh dropout masks
        self.seed_dev = None      # engine.enable_device_step_state(): int64[4] device tensor, seed of decoder pass k of this step

    @property
    def trainable(self) -> bool:
        """freeze_text_decoder: false (models/distributed_gpt3.py:91-93): the decoder's own parameters receive gradients"""
        return any(p.requires_grad for p in self.parameters())

    # -------------------------------------------------------------- explicit forward / backward
    def forward_lm(self, query_features: Optional[torch.Tensor], ids: torch.Tensor, labels: Optional[torch.Tensor],
                   loss_mask: Optional[torch.Tensor], tape: dict, want_logits: bool = False, hidden_only: bool = False,
                   pass_index: int = 0, loss_window: Optional[tuple] = None, window_targets: Optional[tuple] = None):
        """query_features [B*Q, H] (or None), ids [B,L] int64, labels [B,S] int64, loss_mask [B,S-1].
        Returns dict(loss fp32 scalar, losses [B,S-1] fp32, logits?, last_hidden_state [B,S,H]).
        loss_window = (start, length): the caller's promise that loss_mask is zero outside positions
        [start, start + length) of every sequence (pre-training: the Q query slots in front are always masked,
        models/distributed_gpt3.py:142-159).  The tied LM head, the CE and the LM head's dgrad then run on those
        B * length rows only -- the reference evaluates all B * S rows of [S, V] logits and multiplies 80 % of the
        per-token losses by zero; the loss and every gradient are unchanged (those rows' dlogits are exactly 0).
        window_targets = (labels int64 [B * length], weights fp32 [B * length]) of the window rows, already on the device
        (ops.caption_targets): labels / loss_mask are then not consulted and no per-token `losses` tensor is assembled (the
        training step only needs the scalar) -- it takes a dozen small framework launches off every step."""
        cfg = self.config
        lm = self.dist_model.language_model
        B, L = ids.shape
        H, np_, hn, V = cfg.hidden_size, cfg.num_attention_heads, cfg.kv_channels, cfg.vocab_size
        Q = 0 if query_features is None else query_features.shape[0] // B
        S, R = Q + L, B * (Q + L)
        train = self.training
        p_h = cfg.hidden_dropout if train else 0.0
        p_a = cfg.attention_dropout if train else 0.0
        seed = self.step_seed + SEED_PASS_STRIDE * pass_index     # a second decoder pass of one step draws its own dropout masks
        if self.seed_dev is not None:       # the same value, read by the kernels from device memory (include/mpv.h: MPV_SEED_FROM_DEVICE)
            assert pass_index < self.seed_dev.numel()
            seed = (self.seed_dev.data_ptr() + 8 * pass_index) | (1 << 63)
        ids_dev = ids.contiguous() if L > 0 else torch.zeros(1, dtype=torch.long, device=ids.device)   # never dereferenced when L == 0
        h = ops.gpt_embed_fwd(query_features, ids_dev, lm.embedding.word_embeddings.weight,
                              lm.embedding.position_embeddings.weight, B, Q, L, H, dropout_p=p_h, seed=seed,
                              offset=_offset(0, _SITE_EMBED))
        st3 = (S * 3 * H, 3 * hn, 3 * H)
        lay = ops.AttnLayout(st3, st3, st3, (S * H, hn, H))
        scale = 1.0 / math.sqrt(hn)       # alpha=1/(sqrt(hn)*l) then *l inside the softmax (:718-727,757-762): net 1/sqrt(hn)
        window = None
        if loss_window is not None and not want_logits and not hidden_only:
            w0, wl = int(loss_window[0]), int(loss_window[1])
            if 0 <= w0 and w0 + wl <= S and 0 < wl < S:
                window = (w0, wl)
        nl = len(lm.encoder.layers)
        layers = []
        train_dec = self.trainable and not want_logits      # (autograd.Function.forward runs with grad mode off: not a usable signal)
        if train_dec:
            # a trainable decoder runs the reference's own bf16 residual stream (the LayerNorm backward with parameter gradients reads
            # bf16 rows) on all rows of every layer, and keeps what the weight gradients need: the inputs of the four products
            h, xf, mf, rf = self._layers_bf16_stream(h, lay, scale, window, layers, seed, p_h, p_a, B, S, R, H, np_, hn, keep_inputs=True)
        elif FP32_STREAM:
            # stream = the residual stream (bf16 straight out of the embedding, fp32 from the first add on); `pending` = the last
            # sublayer output (bias + dropout applied by its GEMM) that the next LayerNorm adds into the stream in fp32
            stream, pending, pmap = h, None, ops.IDENT
            pend_off = 0                   # dropout site of `pending` (applied by the LayerNorm that adds it: DROPOUT_IN_LN)
            p_g = 0.0 if DROPOUT_IN_LN else p_h      # dropout probability the sublayer GEMMs apply themselves
            p_l = p_h if DROPOUT_IN_LN else 0.0      # ... and the one the adding LayerNorm applies
            smap = ops.IDENT               # rows of `stream` the current layer works on (the loss window in the top layer)
            for li, layer in enumerate(lm.encoder.layers):
                ln = li + 1
                att, mlp = layer.self_attention, layer.mlp
                l1 = layer.input_layernorm
                x1, nxt, m1, r1 = ops.ln_stream_fwd(stream, pending, l1.weight, l1.bias, l1.eps, R, H, amap=pmap,
                                                    add_dropout_p=p_l, seed=seed, offset=pend_off)
                stream = nxt if nxt is not None else stream
                h_in = stream                                                   # x of LayerNorm 1 (for its backward)
                qkv = ops.gemm(x1, att.query_key_value.weight, R, 3 * H, H, bias=att.query_key_value.bias, keep_output=True)
                ctx = torch.empty((R, H), dtype=torch.bfloat16, device=h.device)
                lse = ops.attn_fwd(qkv, qkv[:, hn:], qkv[:, 2 * hn:], ctx, lay, B, np_, S, S, hn, causal=True, scale=scale,
                                   dropout_p=p_a, seed=seed, offset=_offset(ln, _SITE_ATTN))
                top = window is not None and li == nl - 1
                # Top layer under a loss window: nothing downstream reads its hidden states outside the window (they feed only the
                # LM head, which runs on the window), and everything after the attention is row-wise -- the projection, LN2, the MLP
                # and their residual adds run on the B * wl window rows (a row map on the [B*S, H] stream; the sublayer outputs are
                # compact).  K/V of all rows are still needed by the window's queries, so LN1, qkv and the attention stay whole.
                tm = (window[1], S, window[0]) if top else ops.IDENT
                Rl = B * window[1] if top else R
                a1 = ops.gemm(ctx, att.dense.weight, Rl, H, H, bias=att.dense.bias, dropout_p=p_g, seed=seed,
                              offset=_offset(ln, _SITE_DROP1), amap=tm)                            # compact [Rl, H]
                l2 = layer.post_attention_layernorm
                x2, h1, m2, r2 = ops.ln_stream_fwd(stream, a1, l2.weight, l2.bias, l2.eps, Rl, H, hmap=tm, h_rows=R,
                                                   add_dropout_p=p_l, seed=seed, offset=_offset(ln, _SITE_DROP1))
                F4 = mlp.dense_h_to_4h.out_features
                z = torch.empty((Rl, F4), dtype=torch.bfloat16, device=h.device)
                g = ops.gemm(x2, mlp.dense_h_to_4h.weight, Rl, F4, H, bias=mlp.dense_h_to_4h.bias, act=ACT_GELU_TANH, preact_out=z, preact_deriv=True)
                pending = ops.gemm(g, mlp.dense_4h_to_h.weight, Rl, H, F4, bias=mlp.dense_4h_to_h.bias, dropout_p=p_g, seed=seed,
                                   offset=_offset(ln, _SITE_DROP2))
                pend_off = _offset(ln, _SITE_DROP2)                                # compact [Rl, H]
                ent = dict(h=h_in, s1=(m1, r1), qkv=qkv, ctx=ctx, lse=lse, h1=h1, s2=(m2, r2), z=z)
                if top:
                    ent["rows"] = tm
                layers.append(ent)
                stream, smap = h1, tm
            fl = lm.encoder.final_layernorm
            Rf = B * window[1] if window is not None else R
            xf, h, mf, rf = ops.ln_stream_fwd(stream, pending, fl.weight, fl.bias, fl.eps, Rf, H, hmap=smap, h_rows=R,
                                              add_dropout_p=p_l, seed=seed, offset=pend_off)
        else:
            h, xf, mf, rf = self._layers_bf16_stream(h, lay, scale, window, layers, seed, p_h, p_a, B, S, R, H, np_, hn)
        if hidden_only:     # only last_hidden_state is consumed (models/distributed_gpt3.py:958, 583-584, 1149-1150)
            tape.update(B=B, Q=Q, L=L, S=S, layers=layers, h_last=h, sf=(mf, rf), dlogits=None, lm_window=None, lay=lay, scale=scale,
                        seed=seed, p_h=p_h, p_a=p_a)
            return dict(last_hidden_state=xf.view(B, S, H))
        # masked mean of per-token CE over positions 0..S-2 (:1615-1617)
        if window is not None and window_targets is not None:
            Rw = B * window[1]
            logits = ops.gemm(xf, lm.embedding.word_embeddings.weight, Rw, V, H)       # xf is already the window's rows
            _, loss = ops.cross_entropy(logits, window_targets[0], window_targets[1], Rw, V, dlogits=logits)
            tape.update(B=B, Q=Q, L=L, S=S, layers=layers, h_last=h, sf=(mf, rf), dlogits=logits, lm_window=window, lay=lay, scale=scale,
                        seed=seed, p_h=p_h, p_a=p_a, xf=xf if train_dec else None, ids=ids_dev if train_dec else None)
            return dict(loss=loss, losses=None, last_hidden_state=None)
        lmf = loss_mask.to(torch.float32)
        denom = lmf.sum()
        w = torch.zeros((B, S), dtype=torch.float32, device=h.device)
        w[:, :S - 1] = lmf / denom
        if window is not None:
            w0, wl = window
            Rw = B * wl
            logits = ops.gemm(xf, lm.embedding.word_embeddings.weight, Rw, V, H)       # xf is already the window's rows
            losses_w, loss = ops.cross_entropy(logits, labels[:, w0:w0 + wl].contiguous().view(-1),
                                               w[:, w0:w0 + wl].contiguous().view(-1), Rw, V, dlogits=logits)
            losses = torch.zeros((B, S), dtype=torch.float32, device=h.device)
            losses[:, w0:w0 + wl] = losses_w.view(B, wl)
            keep_logits = None
        else:
            logits = ops.gemm(xf, lm.embedding.word_embeddings.weight, R, V, H)          # tied LM head (:1348-1350)
            keep_logits = logits.clone() if want_logits else None
            losses, loss = ops.cross_entropy(logits, labels.contiguous().view(-1), w.view(-1), R, V, dlogits=logits)
        tape.update(B=B, Q=Q, L=L, S=S, layers=layers, h_last=h, sf=(mf, rf), dlogits=logits, lm_window=window, lay=lay, scale=scale,
                    seed=seed, p_h=p_h, p_a=p_a, xf=xf if train_dec else None, ids=ids_dev if train_dec else None)
        # under a loss window the top layer and the final LayerNorm exist on the window rows only: no full last_hidden_state
        out = dict(loss=loss, losses=losses.view(B, S)[:, :S - 1], last_hidden_state=xf.view(B, S, H) if window is None else None)
        if want_logits:
            out["logits"] = keep_logits.view(B, S, V)
        return out

    def _layers_bf16_stream(self, h, lay, scale, window, layers, seed, p_h, p_a, B, S, R, H, np_, hn, keep_inputs=False):
        """The layer stack with the reference's bf16 residual stream (residual adds in the GEMM epilogues): MPV_DECODER_STREAM=bf16,
        and always for a trainable decoder (keep_inputs: x1 / x2 / gelu(z) stay on the tape for the weight gradients, and the top layer
        is not trimmed to the loss window)."""
        lm = self.dist_model.language_model
        nl = len(lm.encoder.layers)
        trim_top = window is not None and not keep_inputs
        for li, layer in enumerate(lm.encoder.layers):
            ln = li + 1
            att, mlp = layer.self_attention, layer.mlp
            x1, m1, r1 = ops.layernorm_fwd(h, layer.input_layernorm.weight, layer.input_layernorm.bias, layer.input_layernorm.eps, R, H)
            qkv = ops.gemm(x1, att.query_key_value.weight, R, 3 * H, H, bias=att.query_key_value.bias, keep_output=True)
            ctx = torch.empty((R, H), dtype=torch.bfloat16, device=h.device)
            lse = ops.attn_fwd(qkv, qkv[:, hn:], qkv[:, 2 * hn:], ctx, lay, B, np_, S, S, hn, causal=True, scale=scale,
                               dropout_p=p_a, seed=seed, offset=_offset(ln, _SITE_ATTN))
            if trim_top and li == nl - 1:
                # Top layer under a loss window: nothing downstream reads its hidden states outside the window (they feed only
                # the LM head, which runs on the window), and everything after the attention is row-wise -- the projection,
                # LN2, the MLP and their residual adds run on the B * wl window rows (a row map on the [B*S, H] stream; the
                # MLP's own tensors are compact).  K/V of all rows are still needed by the window's queries, so LN1, qkv and
                # the attention stay whole.  The reference evaluates all S rows and discards them (models/
                # modeling_distributed_gpt3.py:1340-1366 + distributed_gpt3.py:142-159); loss and gradients are unchanged.
                tm = (window[1], S, window[0])
                Rw = B * window[1]
                h1 = torch.empty((R, H), dtype=torch.bfloat16, device=h.device)
                ops.gemm(ctx, att.dense.weight, Rw, H, H, bias=att.dense.bias, residual=h, dropout_p=p_h, seed=seed,
                         offset=_offset(ln, _SITE_DROP1), amap=tm, cmap=tm, out=h1)
                x2, m2, r2 = ops.layernorm_fwd(h1, layer.post_attention_layernorm.weight, layer.post_attention_layernorm.bias,
                                               layer.post_attention_layernorm.eps, Rw, H, xmap=tm)
                F4 = mlp.dense_h_to_4h.out_features
                z = torch.empty((Rw, F4), dtype=torch.bfloat16, device=h.device)
                g = ops.gemm(x2, mlp.dense_h_to_4h.weight, Rw, F4, H, bias=mlp.dense_h_to_4h.bias, act=ACT_GELU_TANH, preact_out=z, preact_deriv=True)
                h2 = torch.empty((R, H), dtype=torch.bfloat16, device=h.device)
                ops.gemm(g, mlp.dense_4h_to_h.weight, Rw, H, F4, bias=mlp.dense_4h_to_h.bias, residual=h1, dropout_p=p_h,
                         seed=seed, offset=_offset(ln, _SITE_DROP2), cmap=tm, out=h2)
                layers.append(dict(h=h, s1=(m1, r1), qkv=qkv, ctx=ctx, lse=lse, h1=h1, s2=(m2, r2), z=z, rows=tm))
                h = h2
                continue
            h1 = ops.gemm(ctx, att.dense.weight, R, H, H, bias=att.dense.bias, residual=h, dropout_p=p_h, seed=seed,
                          offset=_offset(ln, _SITE_DROP1))
            x2, m2, r2 = ops.layernorm_fwd(h1, layer.post_attention_layernorm.weight, layer.post_attention_layernorm.bias,
                                           layer.post_attention_layernorm.eps, R, H)
            F4 = mlp.dense_h_to_4h.out_features
            z = torch.empty((R, F4), dtype=torch.bfloat16, device=h.device)
            g = ops.gemm(x2, mlp.dense_h_to_4h.weight, R, F4, H, bias=mlp.dense_h_to_4h.bias, act=ACT_GELU_TANH, preact_out=z, preact_deriv=True)
            h2 = ops.gemm(g, mlp.dense_4h_to_h.weight, R, H, F4, bias=mlp.dense_4h_to_h.bias, residual=h1, dropout_p=p_h,
                          seed=seed, offset=_offset(ln, _SITE_DROP2))
            ent = dict(h=h, s1=(m1, r1), qkv=qkv, ctx=ctx, lse=lse, h1=h1, s2=(m2, r2), z=z)
            if keep_inputs:
                ent.update(x1=x1, x2=x2, g=g)
            layers.append(ent)
            h = h2
        fl = lm.encoder.final_layernorm
        if window is not None:       # final LayerNorm on the window rows of the stream -> compact [B * wl, H]
            xf, mf, rf = ops.layernorm_fwd(h, fl.weight, fl.bias, fl.eps, B * window[1], H, xmap=(window[1], S, window[0]))
        else:
            xf, mf, rf = ops.layernorm_fwd(h, fl.weight, fl.bias, fl.eps, R, H)
        return h, xf, mf, rf

    def _window_zero_pair(self, R, H, rows, device):
        key = (R, H, tuple(rows), str(device), torch.cuda.current_stream(device).cuda_stream if torch.device(device).type == "cuda" else 0)
        ent = self.__dict__.setdefault("_wz_cache", {}).get(key)
        if ent is None:
            self._wz_cache.clear()
            ent = self._wz_cache[key] = (torch.zeros((R, H), dtype=torch.bfloat16, device=device),
                                         torch.zeros((R, H), dtype=torch.bfloat16, device=device))
        return ent

    def _dgrad(self, dy, weight, R, n_in, n_out, **kw):
        """dX[R, n_in] = dY[R, n_out] W[n_out, n_in].  A frozen weight (the recipes' freeze_text_decoder: true) is
        kept in a second, transposed copy so the pass runs as the k-contiguous (forward-type, LDS-DMA) GEMM, which is
        measurably faster than the reduction-slow operand path; a trainable weight takes the ordinary dgrad kernel."""
        if weight.requires_grad:
            return ops.gemm(dy, weight, R, n_in, n_out, trans_b=True, **kw)
        cache = self.__dict__.setdefault("_wt_cache", {})
        key = id(weight)
        ent = cache.get(key)
        if ent is None or ent[0] != weight._version or ent[1] != weight.data_ptr():
            ent = (weight._version, weight.data_ptr(), weight.detach().t().contiguous())
            cache[key] = ent
        return ops.gemm(dy, ent[2], R, n_in, n_out, **kw)

    @staticmethod
    def _ln_bwd(dy, x, gamma, mean, rstd, rows, cols, **kw):
        """LayerNorm backward of the frozen decoder: x is the fp32 residual stream (ops.ln_stream_bwd), or bf16 -- the embedding
        output in front of layer 0, and everything under MPV_DECODER_STREAM=bf16."""
        if x.dtype == torch.float32:
            return ops.ln_stream_bwd(dy, x, gamma, mean, rstd, rows, cols, **kw)
        return ops.layernorm_bwd(dy, x, gamma, mean, rstd, rows, cols, **kw)

    def backward_lm(self, tape: dict, grad_loss: Optional[torch.Tensor] = None, d_last_hidden: Optional[torch.Tensor] = None):
        """-> d(query_features) [B*Q, H] (dgrad only: the decoder is frozen).  The seed is the loss (grad_loss scales the
        stored dlogits) and/or d_last_hidden [B*S, H], the gradient of the final-layernorm output."""
        cfg = self.config
        lm = self.dist_model.language_model
        B, Q, L, S = tape["B"], tape["Q"], tape["L"], tape["S"]
        H, np_, hn, V = cfg.hidden_size, cfg.num_attention_heads, cfg.kv_channels, cfg.vocab_size
        R = B * S
        p_h, p_a, seed, lay, scale = tape["p_h"], tape["p_a"], tape["seed"], tape["lay"], tape["scale"]
        nl = len(lm.encoder.layers)
        fl = lm.encoder.final_layernorm
        tm = None
        td = tape.get("xf") is not None       # trainable decoder (forward_lm kept the GEMM inputs): weight gradients below
        wte = lm.embedding.word_embeddings.weight
        if td and tape["dlogits"] is not None:
            # tied LM head (:1348-1350): dWte = grad_loss * dlogits^T xf over the rows the head ran on (the loss window, or all rows);
            # the lookup half of the word-embedding gradient is added at the end (embedding front)
            rows_lm = B * tape["lm_window"][1] if tape.get("lm_window") is not None else R
            ops.gemm(tape["dlogits"], tape["xf"], V, H, rows_lm, trans_a=True, trans_b=True, alpha_dev=grad_loss, out=grad_of(wte))
        if tape["dlogits"] is not None and tape.get("lm_window") is not None:
            # LM head dgrad on the loss window only (few output tiles, K = V: split along K inside mpv_gemm_bf16): dxf is the
            # compact [B * wl, H] gradient of the final LayerNorm's window rows; the final LayerNorm and the top layer's
            # row-wise part run backward on those rows through the row map tm (forward_lm)
            assert d_last_hidden is None, "a loss window leaves no full last_hidden_state to receive a gradient"
            w0, wl = tape["lm_window"]
            tm = (wl, S, w0)
            dxf = self._dgrad(tape["dlogits"], lm.embedding.word_embeddings.weight, B * wl, H, V, alpha_dev=grad_loss)
            tape["dlogits"] = None
        elif tape["dlogits"] is not None:
            dxf = self._dgrad(tape["dlogits"], lm.embedding.word_embeddings.weight, R, H, V, alpha_dev=grad_loss,
                              residual=d_last_hidden)
            tape["dlogits"] = None
        else:
            dxf = d_last_hidden
        drop = p_h > 0.0
        dh_m = torch.empty((R, H), dtype=torch.bfloat16, device=dxf.device) if drop else None
        lnp = lambda ln_mod: dict(dgamma=grad_of(ln_mod.weight), dbeta=grad_of(ln_mod.bias)) if td else {}
        if tm is not None and td:
            # trainable decoder: every layer runs on all rows; the rows outside the window carry an exactly-zero gradient
            dh = torch.zeros((R, H), dtype=torch.bfloat16, device=dxf.device)
            if dh_m is not None:
                dh_m.zero_()
            self._ln_bwd(dxf, tape["h_last"], fl.weight, *tape["sf"], B * tm[0], H, dx=dh, dx_drop=dh_m, dropout_p=p_h, seed=seed,
                              offset=_offset(nl, _SITE_DROP2), xmap=tm, **lnp(fl))
        elif tm is not None:      # window rows of the [B*S, H] stream (the other rows of dh / dh_m are never read)
            dh = torch.empty((R, H), dtype=torch.bfloat16, device=dxf.device)
            self._ln_bwd(dxf, tape["h_last"], fl.weight, *tape["sf"], B * tm[0], H, dx=dh, dx_drop=dh_m, dropout_p=p_h, seed=seed,
                              offset=_offset(nl, _SITE_DROP2), xmap=tm)
        elif td:
            dh = self._ln_bwd(dxf, tape["h_last"], fl.weight, *tape["sf"], R, H, dx_drop=dh_m, dropout_p=p_h, seed=seed,
                                   offset=_offset(nl, _SITE_DROP2), **lnp(fl))
        else:
            dh = self._ln_bwd(dxf, tape["h_last"], fl.weight, *tape["sf"], R, H, dx_drop=dh_m, dropout_p=p_h, seed=seed,
                                   offset=_offset(nl, _SITE_DROP2))
        for li in range(nl - 1, -1, -1):
            layer, s = lm.encoder.layers[li], tape["layers"][li]
            ln = li + 1
            att, mlp = layer.self_attention, layer.mlp
            F4 = mlp.dense_h_to_4h.out_features
            do = dh_m if drop else dh
            if s.get("rows") is not None:
                # top layer under a loss window (forward_lm): MLP, LN2 and the projection backward on the window rows; the
                # gradients of the other rows are exact zeros (dh1 feeds LN1's residual gradient, dctx the attention backward)
                rm = s["rows"]
                Rw = B * rm[0]
                dz = self._dgrad(do, mlp.dense_4h_to_h.weight, Rw, F4, H, act_bwd_z=s["z"], act_bwd=ACT_GELU_TANH, z_is_deriv=True, amap=rm)
                dx2 = self._dgrad(dz, mlp.dense_h_to_4h.weight, Rw, H, F4)
                # the rows outside the window stay exact zeros: two buffers cleared once and kept across steps (only window rows
                # are ever written), instead of two activation-sized clears per step
                dh1, dctx = self._window_zero_pair(R, H, rm, dh.device)
                dh1_m = torch.empty((R, H), dtype=torch.bfloat16, device=dh.device) if drop else None
                self._ln_bwd(dx2, s["h1"], layer.post_attention_layernorm.weight, *s["s2"], Rw, H, dres=dh, dx=dh1, dx_drop=dh1_m,
                                  dropout_p=p_h, seed=seed, offset=_offset(ln, _SITE_DROP1), xmap=rm)
                da = dh1_m if drop else dh1
                self._dgrad(da, att.dense.weight, Rw, H, H, amap=rm, cmap=rm, out=dctx)
            else:
                dz = self._dgrad(do, mlp.dense_4h_to_h.weight, R, F4, H, act_bwd_z=s["z"], act_bwd=ACT_GELU_TANH, z_is_deriv=True)
                if td:      # dense_4h_to_h: dW = do^T gelu(z), db = colsum(do); dense_h_to_4h: dW = dz^T x2, db = colsum(dz)
                    ops.gemm(do, s["g"], H, F4, R, trans_a=True, trans_b=True, out=grad_of(mlp.dense_4h_to_h.weight))
                    ops.colsum(do, R, H, out=grad_of(mlp.dense_4h_to_h.bias))
                    ops.gemm(dz, s["x2"], F4, H, R, trans_a=True, trans_b=True, out=grad_of(mlp.dense_h_to_4h.weight))
                    ops.colsum(dz, R, F4, out=grad_of(mlp.dense_h_to_4h.bias))
                dx2 = self._dgrad(dz, mlp.dense_h_to_4h.weight, R, H, F4)
                dh1_m = torch.empty((R, H), dtype=torch.bfloat16, device=dh.device) if drop else None
                dh1 = self._ln_bwd(dx2, s["h1"], layer.post_attention_layernorm.weight, *s["s2"], R, H, dres=dh, dx_drop=dh1_m,
                                        dropout_p=p_h, seed=seed, offset=_offset(ln, _SITE_DROP1), **lnp(layer.post_attention_layernorm))
                da = dh1_m if drop else dh1
                if td:      # attention output projection: dW = da^T ctx
                    ops.gemm(da, s["ctx"], H, H, R, trans_a=True, trans_b=True, out=grad_of(att.dense.weight))
                    ops.colsum(da, R, H, out=grad_of(att.dense.bias))
                dctx = self._dgrad(da, att.dense.weight, R, H, H)
            qkv = s["qkv"]
            dqkv = torch.empty_like(qkv)
            ops.attn_bwd(qkv, qkv[:, hn:], qkv[:, 2 * hn:], s["ctx"], s["lse"], dctx, dqkv, dqkv[:, hn:], dqkv[:, 2 * hn:], lay,
                         B, np_, S, S, hn, causal=True, scale=scale, dropout_p=p_a, seed=seed, offset=_offset(ln, _SITE_ATTN))
            if td:      # fused q/k/v projection: dW = dqkv^T x1
                ops.gemm(dqkv, s["x1"], 3 * H, H, R, trans_a=True, trans_b=True, out=grad_of(att.query_key_value.weight))
                ops.colsum(dqkv, R, 3 * H, out=grad_of(att.query_key_value.bias))
            dx1 = self._dgrad(dqkv, att.query_key_value.weight, R, H, 3 * H)
            prev_off = _offset(li, _SITE_DROP2) if li > 0 else 0
            want_mask = drop and li > 0
            dh_m = torch.empty((R, H), dtype=torch.bfloat16, device=dh.device) if want_mask else None
            dh = self._ln_bwd(dx1, s["h"], layer.input_layernorm.weight, *s["s1"], R, H, dres=dh1, dx_drop=dh_m,
                                   dropout_p=p_h if want_mask else 0.0, seed=seed, offset=prev_off, **lnp(layer.input_layernorm))
            tape["layers"][li] = None
        if td:
            # embedding front (GPT3Embedding, :640-666): h0 = dropout(x + wpe[s]) with x = the query features (s < Q) or wte[ids].
            # de = mask * dh on every row; dWpe[s] = sum over the batch of de[b, s]; the word-embedding rows get de through the
            # lookup -- a one-hot product so that repeated tokens add up: dWte += onehot(ids)^T de[text rows] (on top of the LM head's half)
            de = ops.gpt_embed_bwd_full(dh, R, H, dropout_p=p_h, seed=seed, offset=_offset(0, _SITE_EMBED))
            wpe = lm.embedding.position_embeddings.weight
            gpe = grad_of(wpe)
            gpe.zero_()
            ops.colsum(de, B, S * H, out=gpe.view(-1)[:S * H])
            if L > 0:
                onehot = torch.zeros((B * L, V), dtype=torch.bfloat16, device=dh.device)
                onehot.scatter_(1, tape["ids"].reshape(-1, 1), 1.0)
                de_text = ops.copy_rows(de, torch.empty((B * L, H), dtype=torch.bfloat16, device=dh.device), B * L, H, smap=(L, S, Q))
                ops.gemm(onehot, de_text, V, H, B * L, trans_a=True, trans_b=True, accumulate=True, out=grad_of(wte))
        if Q == 0:
            return None
        return ops.gpt_embed_bwd(dh, B, Q, L, H, dropout_p=p_h, seed=seed, offset=_offset(0, _SITE_EMBED))

    # -------------------------------------------------------------- reference-shaped API
    def forward(self, tokens=None, input_embeds=None, query_embeds=None, attention_mask=None, position_ids=None, labels=None,
                prompt_length=None, loss_mask=None, is_pair=(False,)):
        """models/modeling_distributed_gpt3.py:1578-1618 signature (no-grad / evaluation use).  The
        training path goes through DistributedGPT3_Pretrain, which drives forward_lm/backward_lm."""
        if tokens is None:
            # models/modeling_distributed_gpt3.py:652-657: without ids the word embeddings ARE input_embeds [B, L, H]
            # (query_embeds, if any, in front): the whole sequence enters as embedding rows, no token lookups
            if input_embeds is None:
                raise ValueError("DistributedGPT3.forward needs `tokens` or `input_embeds`")
            emb = input_embeds if query_embeds is None else torch.cat([query_embeds, input_embeds], dim=1)
            B = emb.shape[0]
            qf = emb.to(torch.bfloat16).reshape(-1, emb.shape[-1]).contiguous()
            tokens = torch.zeros((B, 0), dtype=torch.long, device=emb.device)
            L = 0
        else:
            B, L = tokens.shape
            qf = None
            if query_embeds is not None:
                qf = query_embeds.to(torch.bfloat16).reshape(-1, query_embeds.shape[-1]).contiguous()
        Q = 0 if qf is None else qf.shape[0] // B
        if labels is None:
            labels = torch.zeros((B, Q + L), dtype=torch.long, device=tokens.device)
        if loss_mask is None:
            loss_mask = attention_mask[:, 1:].contiguous() if attention_mask is not None else \
                torch.ones((B, Q + L - 1), dtype=torch.long, device=tokens.device)
        out = self.forward_lm(qf, tokens, labels, loss_mask, {}, want_logits=True)

        class _Out(dict):
            __getattr__ = dict.__getitem__
        return _Out(logits=out["logits"], loss=out["loss"], losses=out["losses"], last_hidden_state=out["last_hidden_state"])


from . import generation as _generation  # noqa: E402  (sample / beam_search / generate, models/modeling_distributed_gpt3.py:1620-1880)

_generation.install(DistributedGPT3)
