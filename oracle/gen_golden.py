"""TEST INFRASTRUCTURE ONLY -- generate tests/golden/*.pt from the reference's own modules.

Usage (this container only; needs /root/reference):
    python -m oracle.gen_golden tiny        # full tensors, CONFIG_TINY, B=2, L=8 (ragged mask)
    python -m oracle.gen_golden configA     # BASELINE.json configs[0]: 1.3B dims, B=2, T=4, L=16

For each case the reference DistributedGPT3_Pretrain (models/distributed_gpt3.py:31-226) is
run in eval() mode (dropout off -- SURVEY.md section 7 "Dropout parity") in fp32 ("intended
function") and in bf16 ("reference as run under DeepSpeed bf16"), forward + backward, and a
compact record is written: loss, per-token losses, (sub-sampled) logits / hidden states /
visual features, and for every trainable parameter the gradient's L2 norm plus a strided
sample.  Weights and inputs are NOT stored: they are regenerated bit-identically from
(cfg, seed) by oracle/weights.py.
"""
from __future__ import annotations

import os
import sys
import time

import torch

from .ref_loader import build_reference_model, reference_forward
from .weights import CONFIG_A, CONFIG_TINY, make_inputs

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

CASES = {
    # name: (cfg, batch, text_len, weight_seed, input_seed, ragged, logits_stride(seq_from, vocab_step))
    "tiny": (CONFIG_TINY, 2, 8, 0, 1234, True, (0, 1)),
    "configA": (CONFIG_A, 2, 16, 0, 1234, False, (128, 50)),
}


# Round 5: the true-dims goldens also hold WHOLE-TENSOR gradients of these tensors (fp32 run, stored as bf16: 2.3e-3
# of relative L2 noise) plus the L2 deviation of the reference's own bf16 run on each -- a 64-element sample of a near-zero gradient
# cannot tell a rounding difference from a wrong row, an L2 distance over every row block can.  Tensors above FULL_GRAD_MAX elements
# are stored as a row-strided slab (every k-th row of the [rows, last-dim] view, k recorded) so that a golden stays a few MB.
FULL_GRAD_MAX = 600_000
FULL_GRAD_KEYS = {
    "gencls": ["cls_head.0.weight", "cls_head.0.bias", "cls_head.2.weight", "cls_head.2.bias", "visual_fc.weight", "learnable_queries",
               "visual_encoder.blocks.0.attn.qkv.weight", "visual_encoder.blocks.6.temporal_fc.weight", "visual_encoder.blocks.11.temporal_attn.qkv.weight",
               "visual_encoder.pos_embed", "visual_encoder.temporal_embed", "attn_pool.attn.bias_k", "attn_pool.attn.in_proj_weight",
               "visual_encoder.blocks.3.norm1.weight", "visual_encoder.blocks.9.temporal_attn.q_bias"],
    "eva": ["visual_fc.weight", "learnable_queries", "visual_encoder.blocks.0.attn.qkv.weight", "visual_encoder.blocks.39.mlp.fc2.weight",
            "visual_encoder.blocks.20.attn.proj.weight", "visual_encoder.pos_embed", "visual_encoder.cls_token", "attn_pool.attn.bias_k",
            "visual_encoder.blocks.20.attn.q_bias", "visual_encoder.blocks.39.norm2.weight"],
}


def grad_slab(g: torch.Tensor):
    """(rows ::k of the [rows, last-dim] view, k) with k the smallest stride that brings the tensor under FULL_GRAD_MAX elements."""
    g2 = g.detach().float().reshape(-1, g.shape[-1])
    k = max(1, -(-g2.numel() // FULL_GRAD_MAX))
    return g2[::k].clone(), k


# Round 6: EVERY trainable tensor of a true-dims golden also carries a WIDE element-strided sample of its gradient (about
# WIDE_SAMPLE elements, fp32 run, stored as bf16; odd stride, so the sample walks through every column and across the rows of a tensor
# with an even last dimension) and the relative L2 deviation of the reference's own bf16 run on the same elements: an L2 check on every
# tensor (a permuted row, a sign, a dropped term are O(1) there), which the 64-element samples were too short for.
WIDE_SAMPLE = 4096


def wide_stride(numel: int) -> int:
    step = max(1, numel // WIDE_SAMPLE)
    return step + 1 if (step % 2 == 0 and step > 1) else step


def grad_wide(g: torch.Tensor):
    f = g.detach().float().reshape(-1)
    return f[::wide_stride(f.numel())][:WIDE_SAMPLE].clone()


def record_wide_grads(model, r, keep, tag):
    """every trainable tensor's wide sample: fp32 pass stores it (bf16) and keeps it; bf16 pass stores its relative L2 deviation"""
    params = dict(model.named_parameters())
    if tag == "fp32":
        r["grad_wide"] = {}
        for n, p in params.items():
            if p.grad is not None:
                keep["wide:" + n] = grad_wide(p.grad)
                r["grad_wide"][n] = keep["wide:" + n].to(torch.bfloat16)
    else:
        r["grad_wide_dev"] = {n: float((grad_wide(p.grad) - keep["wide:" + n]).norm() / (keep["wide:" + n].norm() + 1e-30))
                              for n, p in params.items() if p.grad is not None}


def record_full_grads(model, r, keys, keep, tag):
    """fp32 pass: store the slabs (bf16) and keep them; bf16 pass: store ||g_bf16 - g_fp32|| / ||g_fp32|| per slab.  The same for the
    wide sample of every trainable tensor (grad_wide / grad_wide_dev)."""
    params = dict(model.named_parameters())
    if tag == "fp32":
        r["grad_full"], r["grad_full_stride"] = {}, {}
        for n in keys:
            keep[n], r["grad_full_stride"][n] = grad_slab(params[n].grad)
            r["grad_full"][n] = keep[n].to(torch.bfloat16)
    else:
        r["grad_full_dev"] = {n: float((grad_slab(params[n].grad)[0] - keep[n]).norm() / keep[n].norm()) for n in keys}
    record_wide_grads(model, r, keep, tag)


def grad_sample(g: torch.Tensor, n: int = 64):
    f = g.detach().float().reshape(-1)
    step = max(1, f.numel() // n)
    return f[::step][:n].clone()


def run_case(name: str):
    cfg, B, L, wseed, iseed, ragged, (sfrom, vstep) = CASES[name]
    rec = {"meta": dict(case=name, batch=B, text_len=L, weight_seed=wseed, input_seed=iseed,
                        ragged=ragged, logits_seq_from=sfrom, logits_vocab_step=vstep,
                        torch=str(torch.__version__))}
    keep = {}
    for tag, dtype in (("fp32", torch.float32), ("bf16", torch.bfloat16)):
        t0 = time.time()
        model, sd = build_reference_model(cfg, wseed, dtype=dtype)
        video, ids, mask = make_inputs(cfg, B, L, seed=iseed, ragged=ragged)
        loss, out, input_embeds = reference_forward(model, video.to(dtype), ids, mask, train=False)
        loss.backward()
        r = {
            "loss": loss.detach().float().clone(),
            "losses": out.losses.detach().float().clone(),
            "logits": out.logits.detach()[:, sfrom:, ::vstep].float().clone(),
            "last_hidden_state": out.last_hidden_state.detach()[:, :, ::max(1, cfg.hidden // 256)].float().clone(),
            "query_features": input_embeds.detach()[:, :cfg.num_queries, ::max(1, cfg.hidden // 256)].float().clone(),
            "grad_norm": {}, "grad_sample": {},
        }
        for n, p in model.named_parameters():
            if p.grad is not None:
                r["grad_norm"][n] = float(p.grad.float().norm())
                r["grad_sample"][n] = grad_sample(p.grad)
        record_wide_grads(model, r, keep, tag)
        rec[tag] = r
        print(f"[{name}/{tag}] loss={float(loss):.6f}  {time.time() - t0:.1f}s", flush=True)
        del model, sd
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    path = os.path.join(GOLDEN_DIR, f"{name}.pt")
    torch.save(rec, path)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


def run_retrieval(name: str = "retrieval_tiny"):
    """DistributedGPT3_Retrieval (ITC, BASELINE.json configs[4] shape-reduced): B=8, L=10 ragged, duplicate ids."""
    import types
    from .ref_loader import build_reference_retrieval, single_rank_collectives
    cfg = CONFIG_TINY
    rec = {"meta": dict(case=name, batch=8, text_len=10, weight_seed=2, input_seed=5, ragged=True, idx=[3, 1, 4, 1, 5, 9, 2, 6],
                        torch=str(torch.__version__))}
    keep = {}
    for tag, dtype in (("fp32", torch.float32), ("bf16", torch.bfloat16)):
        model, sd = build_reference_retrieval(cfg, 2, dtype=dtype)
        video, ids, mask = make_inputs(cfg, 8, 10, seed=5, ragged=True)
        idx = torch.tensor(rec["meta"]["idx"])
        model.eval()
        with single_rank_collectives():
            loss = model(video.to(dtype), types.SimpleNamespace(input_ids=ids, attention_mask=mask), idx)
            loss.backward()
        r = {"loss": loss.detach().float().clone(), "grad_norm": {}, "grad_sample": {}}
        for n, p in model.named_parameters():
            if p.grad is not None:
                r["grad_norm"][n] = float(p.grad.float().norm())
                r["grad_sample"][n] = grad_sample(p.grad)
        record_wide_grads(model, r, keep, tag)
        rec[tag] = r
        print(f"[{name}/{tag}] loss={float(loss):.6f}", flush=True)
    path = os.path.join(GOLDEN_DIR, f"{name}.pt")
    torch.save(rec, path)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


def gencls_inputs(cfg, kind: str, Bv: int = 3, L: int = 10, Lp: int = 8):
    """Seeded inputs shared by the golden generator and the tests (defaults: the tiny goldens' shapes)."""
    g = torch.Generator().manual_seed(77 if kind == "itm" else 78)
    C = 2 if kind == "itm" else 3
    video = torch.randn(Bv, 3, cfg.num_frames, cfg.img_size, cfg.img_size, generator=g)
    # two derangements (run_retrieval_..._itm.py:110-111): rotations by 1 and by Bv - 1
    neg = ([(i + 1) % Bv for i in range(Bv)] + [(i + Bv - 1) % Bv for i in range(Bv)]) if kind == "itm" else None
    n = Bv + (len(neg) if neg else 0)

    def text(rows, length):
        ids = torch.randint(0, cfg.vocab, (rows, length), generator=g)
        mask = torch.ones(rows, length, dtype=torch.long)
        lens = torch.randint(3, length + 1, (rows,), generator=g)
        for b in range(rows):
            mask[b, lens[b]:] = 0
        return ids, mask
    ids, mask = text(n, L)
    p_ids, p_mask = text(n, Lp)
    plen = torch.randint(1, 3, (n,), generator=g)
    labels = torch.randint(0, C, (n,), generator=g)
    t = 2 if kind == "itm" else C
    e_ids, e_mask = text(Bv * t, L)
    e_pids, e_pmask = text(Bv * t if kind == "itm" else Bv, Lp)
    e_plen = torch.randint(1, 3, (Bv * t,), generator=g)
    return dict(video=video, neg=neg, ids=ids, mask=mask, p_ids=p_ids, p_mask=p_mask, plen=plen, labels=labels, num_classes=C,
                e_ids=e_ids, e_mask=e_mask, e_pids=e_pids, e_pmask=e_pmask, e_plen=e_plen)


FULL_GENCLS = dict(Bv=3, L=12, Lp=8)      # the true-dims ITM / classification goldens: 16 frames, 1.3B, full depth; 3 clips is what 62 GB of host memory hold of the fp32 reference (4 clips + two derangements: OOM-killed)


def full_gencls_cfg():
    import dataclasses
    from .weights import CONFIG_B
    return dataclasses.replace(CONFIG_B, num_frames=16)          # BASELINE.json configs[4]: 16-frame fine-tune at 1.3B dims


def run_gencls(kind: str, full: bool = False):
    """DistributedGPT3_Retrieval_Cls ("itm") / DistributedGPT3_Cls ("cls"), use_cls on: SURVEY.md section 8(f) rank 1.
    full=True: the reference modules at the TRUE dims (ViT-B/16 x 12 blocks x 16 frames, 24-layer 1.3B decoder), Bv = 4."""
    import types
    from .ref_loader import build_reference_gencls
    cfg = full_gencls_cfg() if full else CONFIG_TINY
    shape = FULL_GENCLS if full else {}
    name = f"{kind}_1p3b" if full else f"{kind}_tiny"
    wseed = 23 if full else 3
    rec = {"meta": dict(case=name, weight_seed=wseed, torch=str(torch.__version__), **shape)}
    keep = {}
    for tag, dtype in (("fp32", torch.float32), ("bf16", torch.bfloat16)):
        t0 = time.time()
        inp = gencls_inputs(cfg, kind, **shape)
        model, sd = build_reference_gencls(cfg, kind, wseed, dtype=dtype, num_classes=inp["num_classes"])
        model.eval()                                                       # dropout off; forward(train=True) still takes the loss branch
        text = types.SimpleNamespace(input_ids=inp["ids"], attention_mask=inp["mask"], prompt_lengths=inp["plen"])
        ptext = types.SimpleNamespace(input_ids=inp["p_ids"], attention_mask=inp["p_mask"])
        if kind == "itm":
            lc, lk = model(inp["video"].to(dtype), text, ptext, inp["neg"], inp["labels"])
        else:
            lc, lk = model(inp["video"].to(dtype), text, ptext, inp["labels"])
        (lc + lk).backward()
        r = {"loss_caption": lc.detach().float().clone(), "loss_cls": lk.detach().float().clone(), "grad_norm": {}, "grad_sample": {}}
        for n, p in model.named_parameters():
            if p.grad is not None:
                r["grad_norm"][n] = float(p.grad.float().norm())
                r["grad_sample"][n] = grad_sample(p.grad)
        if full:
            record_full_grads(model, r, FULL_GRAD_KEYS["gencls"], keep, tag)
        else:
            record_wide_grads(model, r, keep, tag)
        etext = types.SimpleNamespace(input_ids=inp["e_ids"], attention_mask=inp["e_mask"], prompt_lengths=inp["e_plen"])
        eptext = types.SimpleNamespace(input_ids=inp["e_pids"], attention_mask=inp["e_pmask"])
        with torch.no_grad():
            if kind == "itm":
                gen, cl = model(inp["video"].to(dtype), etext, eptext, train=False)
            else:
                gen, cl = model(inp["video"].to(dtype), etext, eptext, train=False)
        r["generation_logits"], r["cls_logits"] = gen.float().clone(), cl.float().clone()
        rec[tag] = r
        print(f"[{name}/{tag}] loss_caption={float(lc):.6f} loss_cls={float(lk):.6f}  {time.time() - t0:.0f}s", flush=True)
        del model, sd
    path = os.path.join(GOLDEN_DIR, f"{name}.pt")
    torch.save(rec, path)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


def full_eva_cfg():
    """EVA-ViT-g as models/eva_vit.py:413-427 builds it (patch 14, width 1408, 40 blocks, 16 heads of 88, MLP 6144) in front
    of the 1.3B decoder."""
    import dataclasses
    from .weights import CONFIG_B
    return dataclasses.replace(CONFIG_B, img_size=224, patch_size=14, vit_dim=1408, vit_depth=40, vit_heads=16, vit_mlp_ratio=4.3637,
                               num_frames=1)


def run_eva(name: str = "eva_tiny"):
    """DistributedGPT3_Pretrain_Image with the EVA encoder.  "eva_tiny": shape-reduced
    (heads of 88, MLP ratio 4.3637, patch 14); B=3, L=9 ragged, prompt_lengths masked.  "eva_g_full": the true ViT-g
    (1408 x 40 blocks) + the 24-layer 1.3B decoder, B=2."""
    import types
    from .ref_loader import build_reference_image
    from .weights import CONFIG_EVA_TINY
    full = name != "eva_tiny"
    cfg = full_eva_cfg() if full else CONFIG_EVA_TINY
    B, wseed = (2, 24) if full else (3, 4)
    rec = {"meta": dict(case=name, batch=B, text_len=9, weight_seed=wseed, input_seed=6, prompt_lengths=[1, 2, 1][:B], torch=str(torch.__version__))}
    keep = {}
    for tag, dtype in (("fp32", torch.float32), ("bf16", torch.bfloat16)):
        t0 = time.time()
        model, sd = build_reference_image(cfg, wseed, dtype=dtype)
        video, ids, mask = make_inputs(cfg, B, 9, seed=6, ragged=True)
        image = video[:, :, 0]
        model.eval()
        captured = {}
        td = model.text_decoder
        orig = td.forward

        def spy(*a, **k):
            out = orig(*a, **k)
            captured["out"] = out
            return out
        td.forward = spy
        text = types.SimpleNamespace(input_ids=ids, attention_mask=mask, prompt_lengths=torch.tensor(rec["meta"]["prompt_lengths"]))
        loss, _ = model(image.to(dtype), text)
        td.forward = orig
        loss.backward()
        out = captured["out"]
        r = {"loss": loss.detach().float().clone(), "losses": out.losses.detach().float().clone(),
             "logits": out.logits.detach()[:, :, ::(50 if full else 8)].float().clone(), "grad_norm": {}, "grad_sample": {}}
        for n, p in model.named_parameters():
            if p.grad is not None:
                r["grad_norm"][n] = float(p.grad.float().norm())
                r["grad_sample"][n] = grad_sample(p.grad)
        if full:
            record_full_grads(model, r, FULL_GRAD_KEYS["eva"], keep, tag)
        else:
            record_wide_grads(model, r, keep, tag)
        rec[tag] = r
        print(f"[{name}/{tag}] loss={float(loss):.6f}  {time.time() - t0:.0f}s", flush=True)
        del model, sd, out, captured
    path = os.path.join(GOLDEN_DIR, f"{name}.pt")
    torch.save(rec, path)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


def run_caption(name: str = "caption_tiny"):
    """DistributedGPT3_Caption.generate (beam 5, KV-cache decode) on the tiny config:
    best sequence + score per sample, plus the per-step logits of a teacher-forced decode for the cache-path check."""
    import types
    from .ref_loader import build_reference_caption, cpu_generation_patches
    from .weights import CONFIG_B
    full = name != "caption_tiny"          # "caption_1p3b": BASELINE configs[1] dims (8 frames, 24-layer 1.3B decoder), three clips
    cfg = CONFIG_B if full else CONFIG_TINY
    B, wseed = (3, 26) if full else (2, 6)
    rec = {"meta": dict(case=name, batch=B, text_len=6, weight_seed=wseed, input_seed=8, tokens_to_generate=12, eod_id=7, torch=str(torch.__version__))}
    for tag, dtype in (("fp32", torch.float32), ("bf16", torch.bfloat16)):
        model, sd = build_reference_caption(cfg, wseed, dtype=dtype, tokens_to_generate=12, eod_id=7)
        video, ids, mask = make_inputs(cfg, B, 6, seed=8, ragged=False)
        mask[1, 4:] = 0
        text = types.SimpleNamespace(input_ids=ids, attention_mask=mask, prompt_lengths=torch.tensor([1] * B))
        model.eval()
        captured = []
        td = model.text_decoder
        orig = td.beam_search

        def spy(*a, **k):
            out = orig(*a, **k)
            captured.append(out)
            return out
        td.beam_search = spy
        with torch.no_grad(), cpu_generation_patches():
            res = model.generate(video.to(dtype), text)
        td.beam_search = orig
        rec[tag] = {"sequences": [r.clone() for r in res], "scores": [c.scores.float().clone() for c in captured]}
        print(f"[{name}/{tag}] " + " | ".join(f"{r.tolist()} {float(c.scores[0]):.5f}" for r, c in zip(res, captured)), flush=True)
    path = os.path.join(GOLDEN_DIR, f"{name}.pt")
    torch.save(rec, path)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


VIDEO_CASES = [   # (T, H, W, res, train, python-random seed)
    (4, 90, 120, 64, True, 3), (3, 120, 90, 64, True, 4), (2, 64, 64, 64, True, 5), (3, 100, 160, 56, False, 0), (2, 48, 40, 64, True, 11),
]


def video_clip(T, H, W, seed):
    g = torch.Generator().manual_seed(1000 + seed)
    base = torch.randint(0, 256, (T, H // 4 + 1, W // 4 + 1, 3), generator=g).float()
    clip = torch.nn.functional.interpolate(base.permute(0, 3, 1, 2), size=(H, W), mode="bilinear").permute(0, 2, 3, 1)
    noise = torch.randint(-20, 21, (T, H, W, 3), generator=g)
    return (clip + noise).clamp(0, 255).to(torch.uint8).contiguous()      # smooth + noise: natural-image-like, full 0..255 range


def run_video(name: str = "video_tiny"):
    """the reference's own video transforms (dataset/__init__.py:60-85 minus the cv2
    RandAugment) on seeded clips; records the crop boxes / flips drawn and the float32 outputs."""
    import random
    from .video_ref import import_video_transforms
    vt, vol = import_video_transforms()
    mean, std = [0.48145466, 0.4578275, 0.40821073], [0.26862954, 0.26130258, 0.27577711]
    rec = {"meta": dict(case=name, cases=VIDEO_CASES, torch=str(torch.__version__)), "out": []}
    for (T, H, W, res, train, seed) in VIDEO_CASES:
        clip = video_clip(T, H, W, seed)
        random.seed(seed)
        if train:
            tf = vt.Compose([vt.RandomResizedCrop(res, scale=(0.5, 1.0), interpolation="bicubic"), vt.RandomHorizontalFlip(),
                             vol.ClipToTensor(channel_nb=3), vt.Normalize(mean=mean, std=std)])
        else:
            tf = vt.Compose([vt.Resize((res, res)), vol.ClipToTensor(channel_nb=3), vt.Normalize(mean=mean, std=std)])
        out = tf(clip)
        rec["out"].append(out.float().clone())
        print(f"[{name}] T={T} {H}x{W} -> {res} train={train}: min {out.min():.3f} max {out.max():.3f}", flush=True)
    path = os.path.join(GOLDEN_DIR, f"{name}.pt")
    torch.save(rec, path)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    for c in (sys.argv[1:] or ["tiny"]):
        if c.startswith("retrieval"):
            run_retrieval(c)
        elif c in ("itm", "cls"):
            run_gencls(c)
        elif c in ("itm_1p3b", "cls_1p3b"):
            run_gencls(c.split("_")[0], full=True)
        elif c.startswith("eva"):
            run_eva(c)
        elif c.startswith("caption"):
            run_caption(c)
        elif c.startswith("video"):
            run_video(c)
        else:
            run_case(c)
