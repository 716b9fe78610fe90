"""TEST INFRASTRUCTURE ONLY -- run the reference's OWN modules on CPU (this container only).

Imports models/vision_transformer.py, models/modeling_distributed_gpt3.py and
models/distributed_gpt3.py UNMODIFIED from /root/reference, with oracle/shims providing
the un-installable third-party packages, and builds DistributedGPT3_Pretrain
(models/distributed_gpt3.py:31-226) with seeded weights.  /root/reference does not exist
on the GPU box; nothing under tests -m gpu / smoke() / bench.py may call this.
"""
from __future__ import annotations

import contextlib
import json
import os
import sys
import tempfile
import types

import torch

from .weights import PathConfig, make_state_dict

REF_ROOT = os.environ.get("MPLUG_REFERENCE_ROOT", "/root/reference")
_SHIMS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "shims")


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REF_ROOT, "models", "distributed_gpt3.py"))


@contextlib.contextmanager
def _cpu_patches():
    """models/modeling_distributed_gpt3.py:1544 calls model.cuda(torch.cuda.current_device())."""
    old_cuda, old_cur = torch.nn.Module.cuda, torch.cuda.current_device
    torch.nn.Module.cuda = lambda self, device=None: self
    torch.cuda.current_device = lambda: 0
    try:
        yield
    finally:
        torch.nn.Module.cuda, torch.cuda.current_device = old_cuda, old_cur


def import_reference():
    """Returns the three reference modules (cached in sys.modules under their own names)."""
    if not reference_available():
        raise RuntimeError(f"reference tree not found at {REF_ROOT}")
    for p in (REF_ROOT, _SHIMS):           # shims first
        if p in sys.path:
            sys.path.remove(p)
        sys.path.insert(0, p)
    # a stale product/top-level `utils` or `models` module would shadow the reference's
    for name in ("utils", "models"):
        m = sys.modules.get(name)
        if m is not None and not str(getattr(m, "__file__", "") or "").startswith((_SHIMS, REF_ROOT)):
            del sys.modules[name]
    import models.vision_transformer as vt            # noqa: E402
    import models.modeling_distributed_gpt3 as mg     # noqa: E402
    import models.distributed_gpt3 as dg              # noqa: E402
    return vt, mg, dg


def _gpt_config_dict(cfg: PathConfig):
    # mirrors configs/models/config_gpt3_1.3B.json field names (read by GPT3Config,
    # models/modeling_distributed_gpt3.py:459-538)
    return {
        "attention_probs_dropout_prob": 0.1, "hidden_act": "gelu", "hidden_dropout_prob": 0.1,
        "hidden_size": cfg.hidden, "initializer_range": 0.02, "intermediate_size": 768,
        "max_position_embeddings": cfg.max_pos, "num_attention_heads": cfg.heads,
        "num_hidden_layers": cfg.layers, "type_vocab_size": 2, "vocab_size": cfg.vocab,
        "attention_type": "self", "fp16": False, "layernorm_epsilon": cfg.gpt_ln_eps,
        "masked_softmax_fusion": False, "ffn_hidden_size": cfg.ffn, "model_type": "gpt3",
    }


def _visual_config_dict(cfg: PathConfig):
    # mirrors configs/models/clip-b16.json minus pretrained_ckpt (no CLIP weights offline)
    return {
        "img_size": cfg.img_size, "patch_size": cfg.patch_size, "depth": cfg.vit_depth,
        "num_frames": cfg.num_frames, "embed_dim": cfg.vit_dim, "num_heads": cfg.vit_heads,
        "mlp_ratio": cfg.vit_mlp_ratio, "drop_path": 0, "grad_ckpt": False,
        "stop_grad_conv1": False, "use_shared_rel_pos_bias": False, "use_abs_pos_emb": True,
        "clip_model": True, "connect_ln": bool(getattr(cfg, "connect_ln", False)),
    }


def build_reference_model(cfg: PathConfig, seed: int = 0, dtype=torch.float32, state_dict=None):
    """DistributedGPT3_Pretrain on CPU with seeded weights; returns (model, state_dict_fp32)."""
    vt, mg, dg = import_reference()
    sd = state_dict if state_dict is not None else make_state_dict(cfg, seed)
    tmp = tempfile.mkdtemp(prefix="mpv_oracle_")
    with open(os.path.join(tmp, "config.json"), "w") as f:
        json.dump(_gpt_config_dict(cfg), f)
    with open(os.path.join(tmp, "visual.json"), "w") as f:
        json.dump(_visual_config_dict(cfg), f)
    with open(os.path.join(tmp, "text.json"), "w") as f:
        json.dump(_gpt_config_dict(cfg), f)
    config = {
        "visual_cfg": os.path.join(tmp, "visual.json"), "text_cfg": os.path.join(tmp, "text.json"),
        "text_decoder": tmp, "megatron_cfg": {"world_size": 1, "model_parallel_size": 1,
                                              "tensor_model_parallel_size": 1},
        "freeze_vit": False, "freeze_text_decoder": True, "num_learnable_token": cfg.num_queries,
        "use_contrastive": False,
    }
    prefix = "text_decoder.dist_model."
    gpt_sd = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
    old_pre_load = mg.pre_load
    mg.pre_load = lambda *a, **k: gpt_sd        # :437-441 would torch.load a 5 GB checkpoint
    try:
        with _cpu_patches():
            model = dg.DistributedGPT3_Pretrain(config=config, tokenizer=None)
    finally:
        mg.pre_load = old_pre_load
    missing, unexpected = model.load_state_dict(sd, strict=True), None
    model = model.to(dtype)
    return model, sd


def reference_forward(model, video, ids, mask, train: bool = False):
    """One forward through the reference module; returns addict.Dict-like outputs of the
    text decoder plus the caption loss, by replaying models/distributed_gpt3.py:130-166."""
    text = types.SimpleNamespace(input_ids=ids, attention_mask=mask)
    model.train(train)
    captured = {}
    td = model.text_decoder
    orig_forward = td.forward

    def spy(*a, **k):
        out = orig_forward(*a, **k)
        captured["out"] = out
        captured["input_embeds"] = k.get("input_embeds")
        return out

    td.forward = spy
    try:
        loss, loss_ita = model(video, text)
    finally:
        td.forward = orig_forward
    return loss, captured["out"], captured["input_embeds"]


def build_reference_retrieval(cfg: PathConfig, seed: int = 0, dtype=torch.float32, embed_dim: int = 256):
    """DistributedGPT3_Retrieval (models/distributed_gpt3.py:817-985) on CPU with seeded weights."""
    from .weights import retrieval_spec
    vt, mg, dg = import_reference()
    sd = make_state_dict(cfg, seed, spec_fn=lambda c: retrieval_spec(c, embed_dim))
    tmp = tempfile.mkdtemp(prefix="mpv_oracle_")
    for name, d in (("config.json", _gpt_config_dict(cfg)), ("visual.json", _visual_config_dict(cfg)), ("text.json", _gpt_config_dict(cfg))):
        with open(os.path.join(tmp, name), "w") as f:
            json.dump(d, f)
    config = {"visual_cfg": os.path.join(tmp, "visual.json"), "text_cfg": os.path.join(tmp, "text.json"), "text_decoder": tmp,
              "megatron_cfg": {"world_size": 1, "model_parallel_size": 1, "tensor_model_parallel_size": 1}, "freeze_vit": False,
              "freeze_text_decoder": True, "num_learnable_token": cfg.num_queries, "num_frames": cfg.num_frames,
              "contrastive_embed_dim": embed_dim, "temp": 0.07}
    prefix = "text_decoder.dist_model."
    gpt_sd = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
    old = mg.pre_load
    mg.pre_load = lambda *a, **k: gpt_sd
    try:
        with _cpu_patches():
            model = dg.DistributedGPT3_Retrieval(config=config, tokenizer=None)
    finally:
        mg.pre_load = old
    model.load_state_dict(sd, strict=True)
    return model.to(dtype), sd


def build_reference_gencls(cfg: PathConfig, kind: str, seed: int = 0, dtype=torch.float32, num_classes: int = 2):
    """DistributedGPT3_Cls (kind="cls", models/distributed_gpt3.py:431-657) or DistributedGPT3_Retrieval_Cls
    (kind="itm", :988-1218) on CPU with seeded weights and use_cls on."""
    from .weights import cls_spec
    vt, mg, dg = import_reference()
    sd = make_state_dict(cfg, seed, spec_fn=lambda c: cls_spec(c, num_classes))
    tmp = tempfile.mkdtemp(prefix="mpv_oracle_")
    for name, d in (("config.json", _gpt_config_dict(cfg)), ("visual.json", _visual_config_dict(cfg)), ("text.json", _gpt_config_dict(cfg))):
        with open(os.path.join(tmp, name), "w") as f:
            json.dump(d, f)
    config = {"visual_cfg": os.path.join(tmp, "visual.json"), "text_cfg": os.path.join(tmp, "text.json"), "text_decoder": tmp,
              "megatron_cfg": {"world_size": 1, "model_parallel_size": 1, "tensor_model_parallel_size": 1}, "freeze_vit": False,
              "freeze_text_decoder": True, "num_learnable_token": cfg.num_queries, "num_frames": cfg.num_frames,
              "use_cls": True, "num_classes": num_classes}
    prefix = "text_decoder.dist_model."
    gpt_sd = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
    old = mg.pre_load
    mg.pre_load = lambda *a, **k: gpt_sd
    try:
        with _cpu_patches():
            klass = dg.DistributedGPT3_Cls if kind == "cls" else dg.DistributedGPT3_Retrieval_Cls
            model = klass(config=config, tokenizer=None)
    finally:
        mg.pre_load = old
    model.load_state_dict(sd, strict=True)
    return model.to(dtype), sd


def build_reference_image(cfg: PathConfig, seed: int = 0, dtype=torch.float32):
    """DistributedGPT3_Pretrain_Image (models/distributed_gpt3.py:229-427) with use_eva_g on CPU with seeded weights.
    create_eva_vit_g hard-codes the ViT-g sizes (models/eva_vit.py:413-427); for a CPU-sized oracle the SAME
    reference VisionTransformer class is built with cfg's reduced width/depth/heads (everything else as in the factory)."""
    from .weights import eva_spec
    vt, mg, dg = import_reference()
    import models.eva_vit as ev
    sd = make_state_dict(cfg, seed, spec_fn=eva_spec)
    tmp = tempfile.mkdtemp(prefix="mpv_oracle_")
    vis = dict(_visual_config_dict(cfg), drop_path=0)
    for name, d in (("config.json", _gpt_config_dict(cfg)), ("visual.json", vis), ("text.json", _gpt_config_dict(cfg))):
        with open(os.path.join(tmp, name), "w") as f:
            json.dump(d, f)
    config = {"visual_cfg": os.path.join(tmp, "visual.json"), "text_cfg": os.path.join(tmp, "text.json"), "text_decoder": tmp,
              "megatron_cfg": {"world_size": 1, "model_parallel_size": 1, "tensor_model_parallel_size": 1}, "freeze_vit": False,
              "freeze_text_decoder": True, "num_learnable_token": cfg.num_queries, "use_eva_g": True, "use_contrastive": False}

    def reduced_factory(img_size=224, drop_path_rate=0.4, norm_layer=None, use_checkpoint=True, precision="fp16"):
        return ev.VisionTransformer(img_size=img_size, patch_size=cfg.patch_size, use_mean_pooling=False, embed_dim=cfg.vit_dim,
                                    depth=cfg.vit_depth, num_heads=cfg.vit_heads, mlp_ratio=cfg.vit_mlp_ratio, qkv_bias=True,
                                    drop_path_rate=drop_path_rate, norm_layer=norm_layer, use_checkpoint=False)

    prefix = "text_decoder.dist_model."
    gpt_sd = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
    old, old_f = mg.pre_load, dg.create_eva_vit_g
    mg.pre_load = lambda *a, **k: gpt_sd
    dg.create_eva_vit_g = reduced_factory
    try:
        with _cpu_patches():
            model = dg.DistributedGPT3_Pretrain_Image(config=config, tokenizer=None)
    finally:
        mg.pre_load, dg.create_eva_vit_g = old, old_f
    model.load_state_dict(sd, strict=True)
    return model.to(dtype), sd


@contextlib.contextmanager
def cpu_generation_patches():
    """sample()/beam_search() build their bookkeeping tensors on torch.cuda.current_device()
    (models/modeling_distributed_gpt3.py:864, 1664, 1767): point that at the CPU."""
    old_cur = torch.cuda.current_device
    torch.cuda.current_device = lambda: torch.device("cpu")
    try:
        yield
    finally:
        torch.cuda.current_device = old_cur


def build_reference_caption(cfg: PathConfig, seed: int = 0, dtype=torch.float32, tokens_to_generate: int = 12, eod_id: int = 7):
    """DistributedGPT3_Caption (models/distributed_gpt3.py:661-814) on CPU with seeded weights; generate() = per-sample
    beam search over the KV-cache decode path."""
    vt, mg, dg = import_reference()
    sd = make_state_dict(cfg, seed)
    tmp = tempfile.mkdtemp(prefix="mpv_oracle_")
    gcfg = dict(_gpt_config_dict(cfg), tokens_to_generate=tokens_to_generate, eod_id=eod_id)
    for name, d in (("config.json", gcfg), ("visual.json", _visual_config_dict(cfg)), ("text.json", gcfg)):
        with open(os.path.join(tmp, name), "w") as f:
            json.dump(d, f)
    config = {"visual_cfg": os.path.join(tmp, "visual.json"), "text_cfg": os.path.join(tmp, "text.json"), "text_decoder": tmp,
              "megatron_cfg": {"world_size": 1, "model_parallel_size": 1, "tensor_model_parallel_size": 1}, "freeze_vit": False,
              "freeze_text_decoder": True, "num_learnable_token": cfg.num_queries, "num_frames": cfg.num_frames}
    prefix = "text_decoder.dist_model."
    gpt_sd = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
    old = mg.pre_load
    mg.pre_load = lambda *a, **k: gpt_sd
    tok = types.SimpleNamespace(tokenizer=types.SimpleNamespace(eos=eod_id))
    try:
        with _cpu_patches():
            model = dg.DistributedGPT3_Caption(config=config, tokenizer=tok)
    finally:
        mg.pre_load = old
    model.load_state_dict(sd, strict=True)
    model = model.to(dtype)
    for m in model.modules():                 # the KV memory is allocated in params_dtype (:864-871): follow the cast
        if hasattr(m, "params_dtype"):
            m.params_dtype = dtype
    return model, sd


@contextlib.contextmanager
def single_rank_collectives():
    """models/distributed_gpt3.py:962-964 call torch.distributed collectives unconditionally: run them on a
    1-rank gloo group so the reference forward works in a plain process."""
    import torch.distributed as dist
    created = False
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(29400 + os.getpid() % 500))
        dist.init_process_group("gloo", rank=0, world_size=1)
        created = True
    try:
        yield
    finally:
        if created:
            dist.destroy_process_group()
