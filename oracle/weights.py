"""TEST INFRASTRUCTURE ONLY -- seeded configs, weights and inputs for the hot path.

Key names/shapes follow the reference state-dict (SURVEY.md section 8(b);
models/vision_transformer.py:449-481,125-131,217-234,347-358; models/distributed_gpt3.py:99-116;
models/modeling_distributed_gpt3.py:562-578,619-624,843-857,1002-1024,1131).
Every tensor is drawn from ONE torch CPU generator in sorted-key order so the same
(cfg, seed) reproduces bit-identical weights on any box with this torch build.
Module-default init leaves temporal_fc / temporal_embed at zero and
LN affine at (1,0); here everything is randomised so every path is exercised.
"""
from __future__ import annotations

import dataclasses
import math
from collections import OrderedDict

import torch


@dataclasses.dataclass(frozen=True)
class PathConfig:
    # vision tower (configs/models/clip-b16.json)
    img_size: int = 224
    patch_size: int = 16
    vit_dim: int = 768
    vit_depth: int = 12
    vit_heads: int = 8
    vit_mlp_ratio: int = 4
    num_frames: int = 4
    vit_ln_eps: float = 1e-6          # models/distributed_gpt3.py:45
    # abstractor
    num_queries: int = 128            # num_learnable_token
    # GPT-3 (configs/models/config_gpt3_1.3B.json)
    hidden: int = 2048
    layers: int = 24
    heads: int = 32
    ffn: int = 8192
    vocab: int = 51200
    max_pos: int = 2048
    gpt_ln_eps: float = 1e-5
    connect_ln: bool = False          # visual_cfg['connect_ln']: LayerNorm behind visual_fc (models/distributed_gpt3.py:112-115,136)

    @property
    def n_patches(self):
        return (self.img_size // self.patch_size) ** 2

    @property
    def vit_head_dim(self):
        return self.vit_dim // self.vit_heads

    @property
    def head_dim(self):
        return self.hidden // self.heads


CONFIG_A = PathConfig()                                   # BASELINE.json configs[0] dims (B=2,T=4,L=16)
CONFIG_B = dataclasses.replace(CONFIG_A, num_frames=8)    # configs[1]: B=32, T=8, L=32
CONFIG_D = dataclasses.replace(CONFIG_B, hidden=2560, layers=32, heads=32, ffn=10240)  # 2.7B
# Small true-head-dim config for fast full-tensor parity (vit head_dim 96, gpt head_dim 64).
CONFIG_TINY = PathConfig(img_size=64, patch_size=16, vit_dim=192, vit_depth=2, vit_heads=2,
                         num_frames=4, num_queries=32, hidden=256, layers=2, heads=4, ffn=1024,
                         vocab=1024, max_pos=256)


def state_dict_spec(cfg: PathConfig):
    """[(key, shape, kind)] for DistributedGPT3_Pretrain.state_dict(). kind picks the init scale."""
    D, H = cfg.vit_dim, cfg.hidden
    hid = D * cfg.vit_mlp_ratio
    P = cfg.patch_size
    s = []

    def ln(prefix, n):
        s.append((prefix + ".weight", (n,), "ln_w"))
        s.append((prefix + ".bias", (n,), "ln_b"))

    def lin(prefix, out, inp, bias=True, kind="w_vit"):
        s.append((prefix + ".weight", (out, inp), kind))
        if bias:
            s.append((prefix + ".bias", (out,), "bias"))

    ve = "visual_encoder."
    s.append((ve + "cls_token", (1, 1, D), "embed"))
    s.append((ve + "pos_embed", (1, cfg.n_patches + 1, D), "embed"))
    s.append((ve + "temporal_embed", (1, cfg.num_frames, D), "embed"))
    s.append((ve + "patch_embed.proj.weight", (D, 3, P, P), "w_vit"))
    ln(ve + "norm_pre", D)
    for i in range(cfg.vit_depth):
        b = f"{ve}blocks.{i}."
        for n in ("norm1", "norm2", "temporal_ln"):
            ln(b + n, D)
        for a in ("attn", "temporal_attn"):
            s.append((b + a + ".qkv.weight", (3 * D, D), "w_vit"))
            s.append((b + a + ".q_bias", (D,), "bias"))
            s.append((b + a + ".v_bias", (D,), "bias"))
            lin(b + a + ".proj", D, D)
        lin(b + "temporal_fc", D, D)
        lin(b + "mlp.fc1", hid, D)
        lin(b + "mlp.fc2", D, hid)
    ln(ve + "norm", D)
    s.append(("learnable_queries", (1, cfg.num_queries, D), "embed"))
    for n in ("norm1", "normk", "norm2"):
        ln("attn_pool." + n, D)
    s.append(("attn_pool.attn.in_proj_weight", (3 * D, D), "w_vit"))
    s.append(("attn_pool.attn.in_proj_bias", (3 * D,), "bias"))
    s.append(("attn_pool.attn.bias_k", (1, 1, D), "embed"))
    s.append(("attn_pool.attn.bias_v", (1, 1, D), "embed"))
    lin("attn_pool.attn.out_proj", D, D)
    lin("attn_pool.mlp.fc1", hid, D)
    lin("attn_pool.mlp.fc2", D, hid)
    lin("visual_fc", H, D)
    if cfg.connect_ln:
        ln("visual_norm", H)
    lm = "text_decoder.dist_model.language_model."
    s.append((lm + "embedding.word_embeddings.weight", (cfg.vocab, H), "w_gpt"))
    s.append((lm + "embedding.position_embeddings.weight", (cfg.max_pos, H), "w_gpt"))
    for i in range(cfg.layers):
        b = f"{lm}encoder.layers.{i}."
        ln(b + "input_layernorm", H)
        ln(b + "post_attention_layernorm", H)
        lin(b + "self_attention.query_key_value", 3 * H, H, kind="w_gpt")
        lin(b + "self_attention.dense", H, H, kind="w_gpt_out")
        lin(b + "mlp.dense_h_to_4h", cfg.ffn, H, kind="w_gpt")
        lin(b + "mlp.dense_4h_to_h", H, cfg.ffn, kind="w_gpt_out")
    ln(lm + "encoder.final_layernorm", H)
    return s


def retrieval_spec(cfg: PathConfig, embed_dim: int = 256):
    """DistributedGPT3_Retrieval.state_dict() = pre-train keys + ITC head (models/distributed_gpt3.py:904-907)."""
    s = state_dict_spec(cfg)
    s.append(("vision_proj.weight", (embed_dim, cfg.vit_dim), "w_vit"))
    s.append(("vision_proj.bias", (embed_dim,), "bias"))
    s.append(("text_proj.weight", (embed_dim, cfg.hidden), "w_proj"))
    s.append(("text_proj.bias", (embed_dim,), "bias"))
    s.append(("temp", (), "temp"))
    return s


def cls_spec(cfg: PathConfig, num_classes: int = 2):
    """DistributedGPT3_Cls / DistributedGPT3_Retrieval_Cls .state_dict() = pre-train keys + cls_head
    (models/distributed_gpt3.py:524-530, 1079-1085)."""
    s = state_dict_spec(cfg)
    s.append(("cls_head.0.weight", (cfg.hidden, cfg.hidden), "w_proj"))
    s.append(("cls_head.0.bias", (cfg.hidden,), "bias"))
    s.append(("cls_head.2.weight", (num_classes, cfg.hidden), "w_proj"))
    s.append(("cls_head.2.bias", (num_classes,), "bias"))
    return s


CONFIG_EVA_TINY = dataclasses.replace(CONFIG_TINY, img_size=56, patch_size=14, vit_dim=176, vit_depth=2, vit_heads=2,
                                      vit_mlp_ratio=4.3637, num_frames=1)     # EVA-ViT-g shape-reduced: heads of 88, MLP ratio 4.3637


def eva_spec(cfg: PathConfig):
    """DistributedGPT3_Pretrain_Image(use_eva_g).state_dict(): EVA VisionTransformer keys (models/eva_vit.py:245-305)
    + abstractor / visual_fc / decoder keys of the pre-train model."""
    D = cfg.vit_dim
    hid = int(D * cfg.vit_mlp_ratio)
    P = cfg.patch_size
    s = []

    def ln(prefix, n):
        s.append((prefix + ".weight", (n,), "ln_w"))
        s.append((prefix + ".bias", (n,), "ln_b"))

    def lin(prefix, out, inp):
        s.append((prefix + ".weight", (out, inp), "w_vit"))
        s.append((prefix + ".bias", (out,), "bias"))

    ve = "visual_encoder."
    s.append((ve + "cls_token", (1, 1, D), "embed"))
    s.append((ve + "pos_embed", (1, cfg.n_patches + 1, D), "embed"))
    s.append((ve + "patch_embed.proj.weight", (D, 3, P, P), "w_vit"))
    s.append((ve + "patch_embed.proj.bias", (D,), "bias"))
    for i in range(cfg.vit_depth):
        b = f"{ve}blocks.{i}."
        ln(b + "norm1", D)
        s.append((b + "attn.qkv.weight", (3 * D, D), "w_vit"))
        s.append((b + "attn.q_bias", (D,), "bias"))
        s.append((b + "attn.v_bias", (D,), "bias"))
        lin(b + "attn.proj", D, D)
        ln(b + "norm2", D)
        lin(b + "mlp.fc1", hid, D)
        lin(b + "mlp.fc2", D, hid)
    ln(ve + "norm", D)
    for key, shape, kind in state_dict_spec(dataclasses.replace(cfg, vit_mlp_ratio=1)):
        if key.startswith("visual_encoder."):
            continue
        if key.startswith("attn_pool.mlp.fc1"):
            shape = (hid, D) if key.endswith("weight") else (hid,)
        if key.startswith("attn_pool.mlp.fc2.weight"):
            shape = (D, hid)
        s.append((key, shape, kind))
    return s


def make_state_dict(cfg: PathConfig, seed: int = 0, dtype=torch.float32, spec_fn=state_dict_spec) -> "OrderedDict[str, torch.Tensor]":
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    out = OrderedDict()
    spec = sorted(spec_fn(cfg), key=lambda t: t[0])
    for key, shape, kind in spec:
        r = torch.randn(shape, generator=g, dtype=torch.float32)
        if kind == "ln_w":
            t = 1.0 + 0.1 * r
        elif kind == "ln_b":
            t = 0.05 * r
        elif kind == "bias":
            t = 0.02 * r
        elif kind == "embed":
            t = 0.02 * r
        elif kind == "w_vit":
            t = 0.02 * r
        elif kind == "w_gpt":
            t = 0.02 * r
        elif kind == "w_proj":
            t = 0.05 * r
        elif kind == "temp":
            t = torch.tensor(0.07)
        elif kind == "w_gpt_out":
            t = (0.02 / math.sqrt(2.0 * cfg.layers)) * r
        else:  # pragma: no cover
            raise KeyError(kind)
        out[key] = t.to(dtype)
    # keep original (reference) ordering for load_state_dict friendliness
    ordered = OrderedDict((k, out[k]) for k, _, _ in spec_fn(cfg))
    return ordered


def make_inputs(cfg: PathConfig, batch: int, text_len: int, seed: int = 1234, ragged: bool = False):
    """video ~ N(0,1) [B,3,T,H,W] (dataset contract: dataset/__init__.py:68-69), ids ~ U[0,V)."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    video = torch.randn(batch, 3, cfg.num_frames, cfg.img_size, cfg.img_size, generator=g)
    ids = torch.randint(0, cfg.vocab, (batch, text_len), generator=g)
    mask = torch.ones(batch, text_len, dtype=torch.long)
    if ragged:
        lens = torch.randint(2, text_len + 1, (batch,), generator=g)
        for b in range(batch):
            mask[b, lens[b]:] = 0
    return video, ids, mask
