"""Parity where the benchmark runs: the HIP path at the TRUE dims and FULL depth of BASELINE.json configs[1] (1.3B, T=8),
configs[3] (2.7B decoder, 32 layers) and configs[4] (ITC retrieval, 16 frames) against the oracle restatement
(oracle/restate.py, pinned to the reference's own modules at 1e-5 by tests/test_host_cpu.py) run live in fp32 on the host,
at batch sizes the host finishes in seconds.  eval() mode.  Every case appends THREE deviations per quantity to a parity report
(gpurun_out/r03_parity.txt, committed under profiles/), all as max-abs error / max-abs reference:
  (1) HIP (bf16) vs the fp32 oracle               -- the distance to the function the reference defines;
  (2) the oracle itself run in bf16 vs its fp32 run -- what the reference's OWN bf16 execution loses (the yardstick);
  (3) HIP (bf16) vs the oracle run in bf16          -- two bf16 executions with different rounding points.
north_star asks for logits within 1e-2.  The gates below are plain numbers (LOGITS_GATE etc.), stated per case.
"""
import dataclasses
import math
import os
import time
import types

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REPORT = os.environ.get("MPV_PARITY_REPORT", os.path.join(ROOT, "gpurun_out", "r03_parity.txt"))


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


def report(line):
    os.makedirs(os.path.dirname(REPORT), exist_ok=True)
    with open(REPORT, "a") as f:
        f.write(line + "\n")
    print(line)


def _pretrain_case(name, cfg, dev, B, L, wseed, grad_keys, logits_gate, hidden_gate, full_mask=False):
    from oracle import restate
    from oracle.weights import make_inputs, make_state_dict
    from youku_mplug_amd.pretrain import synthetic_model
    t0 = time.time()
    model = synthetic_model(cfg, device=dev)
    sd = make_state_dict(cfg, wseed)
    model.load_state_dict({k: v.to(torch.bfloat16) for k, v in sd.items()}, strict=True)
    model.eval()
    video, ids, mask = make_inputs(cfg, B, L, seed=31, ragged=not full_mask)
    text = types.SimpleNamespace(input_ids=ids.to(dev), attention_mask=mask.to(dev))
    vid = video.to(dev).to(torch.bfloat16)
    out = model.forward_outputs(vid, text)
    loss, _ = model(vid, text)
    loss.backward()
    torch.cuda.synchronize()
    # fp32 restatement on the same bf16-rounded weights and inputs
    sdr = {k: v.bfloat16().float() for k, v in sd.items()}
    for k in grad_keys:
        sdr[k].requires_grad_(True)
    ref = restate.pretrain_forward(video.bfloat16().float(), ids, mask, sdr, cfg)
    ref["loss"].backward()
    # the restatement in bf16 (what the reference's own bf16 run loses against its fp32 run)
    with torch.no_grad():
        sdb = {k: v.bfloat16() for k, v in sd.items()}
        refb = restate.pretrain_forward(video.bfloat16(), ids, mask, sdb, cfg)
    e = dict(logits=rel(out.logits, ref["logits"].detach()), hidden=rel(out.last_hidden_state, ref["last_hidden_state"].detach()),
             losses=rel(out.losses, ref["losses"].detach()), loss=abs(out.loss.item() - ref["loss"].item()) / abs(ref["loss"].item()))
    eb = dict(logits=rel(refb["logits"], ref["logits"].detach()), hidden=rel(refb["last_hidden_state"], ref["last_hidden_state"].detach()),
              losses=rel(refb["losses"], ref["losses"].detach()))
    ex = dict(logits=rel(out.logits, refb["logits"]), hidden=rel(out.last_hidden_state, refb["last_hidden_state"]),
              losses=rel(out.losses, refb["losses"]))
    params = dict(model.named_parameters())
    worst, worst_norm = 0.0, 0.0
    for k in grad_keys:
        g, r = params[k].grad.float().cpu(), sdr[k].grad
        worst = max(worst, rel(g, r))
        worst_norm = max(worst_norm, abs(g.norm().item() - r.norm().item()) / r.norm().item())
    report(f"{name}: B={B} L={L} S={cfg.num_queries + L} frames={cfg.num_frames} layers={cfg.layers} mask={'full' if full_mask else 'ragged'}\n"
           f"    (1) HIP vs fp32 oracle        : logits {e['logits']:.3e} hidden {e['hidden']:.3e} losses {e['losses']:.3e} loss {e['loss']:.3e} worst-grad {worst:.3e} worst-grad-norm {worst_norm:.3e}\n"
           f"    (2) oracle bf16 vs fp32 oracle: logits {eb['logits']:.3e} hidden {eb['hidden']:.3e} losses {eb['losses']:.3e}\n"
           f"    (3) HIP vs oracle bf16        : logits {ex['logits']:.3e} hidden {ex['hidden']:.3e} losses {ex['losses']:.3e}\n"
           f"    gates on (1): logits <= {logits_gate:.1e}, hidden <= {hidden_gate:.1e}, losses <= 1.0e-02, loss <= 5.0e-03, worst-grad <= 4.0e-02, worst-grad-norm <= 1.0e-02 | {time.time() - t0:.0f} s")
    assert e["logits"] <= logits_gate, e
    assert e["hidden"] <= hidden_gate, e
    assert e["losses"] <= 1e-2, e
    assert e["loss"] <= 5e-3, e
    assert worst <= 4e-2 and worst_norm <= 1e-2, (worst, worst_norm)


# Gates on deviation (1), as numbers: north_star's 1e-2 for the logits, at every full-depth shape.  With the decoder's residual
# stream in fp32 (gpt3.FP32_STREAM) the measured deviations are 6.6e-3 (config B), 7.7e-3 (S = 208) and 7.9e-3 (config D, 32
# layers); the reference's own bf16 execution -- column (2) -- sits at 1.3-1.6e-2, and so does column (3), which is dominated by it.
LOGITS_GATE_B, HIDDEN_GATE_B = 1.0e-2, 1.0e-2
LOGITS_GATE_D, HIDDEN_GATE_D = 1.0e-2, 1.2e-2

GRAD_KEYS = ["visual_fc.weight", "learnable_queries", "visual_encoder.blocks.0.attn.qkv.weight", "visual_encoder.blocks.11.mlp.fc2.weight",
             "visual_encoder.blocks.5.temporal_fc.weight", "visual_encoder.pos_embed", "visual_encoder.norm.weight"]


def test_configB_full_depth_vs_oracle(dev):
    """configs[1] dims exactly (ViT-B/16 x 12 blocks x 8 frames, 128 queries, 24-layer 1.3B decoder, V = 51200), B = 2."""
    from oracle.weights import CONFIG_B
    _pretrain_case("config B (1.3B, T=8, full depth)", CONFIG_B, dev, B=2, L=32, wseed=11, grad_keys=GRAD_KEYS, logits_gate=LOGITS_GATE_B,
                   hidden_gate=HIDDEN_GATE_B)


def test_yaml_geometry_full_depth_vs_oracle(dev):
    """The geometry the shipped YAML runs (configs/pretrain/gpt3_1.3B/pretrain_gpt3_freezeGPT_youku_v0.yaml: 4 frames, titles of up
    to 80 tokens -> S = 128 + 80 = 208: seven 32-row tiles per decoder attention problem instead of five), 1.3B dims, full depth,
    full (non-ragged) attention masks."""
    from oracle.weights import CONFIG_B
    cfg = dataclasses.replace(CONFIG_B, num_frames=4)
    _pretrain_case("YAML-as-shipped geometry (1.3B, T=4, L=80, full depth)", cfg, dev, B=2, L=80, wseed=14, grad_keys=GRAD_KEYS[:4],
                   logits_gate=LOGITS_GATE_B, hidden_gate=HIDDEN_GATE_B, full_mask=True)


def test_configD_full_depth_vs_oracle(dev):
    """configs[3] dims exactly (2.7B decoder: 32 layers, hidden 2560, head_dim 80, ffn 10240), B = 1."""
    from oracle.weights import CONFIG_D
    _pretrain_case("config D (2.7B, T=8, full depth)", CONFIG_D, dev, B=1, L=32, wseed=12, grad_keys=GRAD_KEYS[:4], logits_gate=LOGITS_GATE_D,
                   hidden_gate=HIDDEN_GATE_D)


def test_retrieval_config5_shape_vs_oracle(dev):
    """configs[4]: ITC retrieval fine-tune at 16 frames, 1.3B dims, full depth; B = 8 (the contrastive loss needs a batch)."""
    from oracle import restate
    from oracle.weights import CONFIG_B, make_inputs, make_state_dict, retrieval_spec
    from youku_mplug_amd.retrieval import synthetic_retrieval_model
    cfg = dataclasses.replace(CONFIG_B, num_frames=16)
    t0 = time.time()
    model = synthetic_retrieval_model(cfg, device=dev)
    sd = make_state_dict(cfg, 13, spec_fn=retrieval_spec)
    model.load_state_dict({k: v.to(torch.bfloat16) for k, v in sd.items()}, strict=True)
    model.eval()
    B, L = 8, 32   # the similarity GEMMs take a global batch that is a multiple of 8
    video, ids, mask = make_inputs(cfg, B, L, seed=41, ragged=True)
    idx = torch.arange(B)
    text = types.SimpleNamespace(input_ids=ids.to(dev), attention_mask=mask.to(dev))
    vid = video.to(dev).to(torch.bfloat16)
    loss = model(vid, text, idx.to(dev))
    loss.backward()
    torch.cuda.synchronize()
    sdr = {k: v.bfloat16().float() for k, v in sd.items()}
    keys = ["vision_proj.weight", "text_proj.weight", "visual_encoder.blocks.0.attn.qkv.weight", "visual_encoder.temporal_embed"]
    for k in keys:
        sdr[k].requires_grad_(True)
    ref = restate.retrieval_forward(video.bfloat16().float(), ids, mask, idx, sdr, cfg)
    ref["loss"].backward()
    vf, tf = model.extract_vision_feature(vid), model.extract_text_feature(text)
    e = dict(vision=rel(vf, ref["vision_feats"].detach()), text=rel(tf, ref["text_feat"].detach()),
             loss=abs(loss.item() - ref["loss"].item()) / abs(ref["loss"].item()))
    params = dict(model.named_parameters())
    worst = max(rel(params[k].grad, sdr[k].grad) for k in keys)
    report(f"retrieval config 5 shape (1.3B, T=16, full depth): B={B} L={L} | HIP vs fp32 oracle: vision feats {e['vision']:.3e} "
           f"text feats {e['text']:.3e} loss {e['loss']:.3e} worst-grad {worst:.3e} | {time.time() - t0:.0f} s")
    assert e["vision"] <= 1.5e-2 and e["text"] <= 1.2e-2 and e["loss"] <= 5e-3 and worst <= 6e-2, (e, worst)
