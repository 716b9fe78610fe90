"""Where does the bf16 deviation of the full-depth config-B forward come from?  CPU-only bisect on the oracle restatement
(oracle/restate.py): the same weights / inputs as tests/test_parity_fullsize_gpu.py, run (1) all fp32 = the yardstick,
(2) all bf16 = what the reference's own bf16 run loses, and mixed arms: (3) vision tower + abstractor in fp32 with the decoder in
bf16, (4) the reverse, (5) everything bf16 except an fp32 RESIDUAL STREAM through the 24 decoder layers (sublayers read a bf16
copy, their outputs are added to an fp32 h), (6) the same for the ViT residual stream too.  Prints max-abs error / max-abs
reference of logits and last hidden state against (1).  Test infrastructure (imports oracle/); python tools/parity_bisect.py [B]"""
import math
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import restate  # noqa: E402
from oracle.weights import CONFIG_B, make_inputs, make_state_dict  # noqa: E402


def rel(a, b):
    a, b = a.float(), b.float()
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


def gpt_forward_fp32_stream(input_embeds, sd, cfg, p="text_decoder.dist_model.language_model."):
    """restate.gpt_forward with bf16 sublayers and an fp32 residual stream (the arm under test)."""
    B, S, H = input_embeds.shape
    bf = torch.bfloat16
    pos = sd[p + "embedding.position_embeddings.weight"][:S]
    h = (input_embeds + pos[None]).transpose(0, 1).contiguous().float()
    causal = torch.tril(torch.ones(1, 1, S, S)) < 0.5
    zero = torch.zeros_like(h, dtype=bf)
    for i in range(cfg.layers):
        lp = f"{p}encoder.layers.{i}."
        # a layer on the bf16 copy of the stream, its two sublayer outputs recovered by differencing against a zero-stream pass
        # would double the cost; instead restate the layer here with the adds in fp32
        s, b, _ = h.shape
        np_, hn = cfg.heads, cfg.head_dim
        x = restate.ln_fp32(h, sd[lp + "input_layernorm.weight"], sd[lp + "input_layernorm.bias"], cfg.gpt_ln_eps).to(bf)
        mixed = F.linear(x, sd[lp + "self_attention.query_key_value.weight"], sd[lp + "self_attention.query_key_value.bias"]).view(s, b, np_, 3 * hn)
        q, k, v = torch.split(mixed, hn, dim=-1)
        q = q.reshape(s, b * np_, hn).transpose(0, 1)
        k = k.reshape(s, b * np_, hn).transpose(0, 1)
        v = v.reshape(s, b * np_, hn).transpose(0, 1)
        scores = (torch.bmm(q, k.transpose(1, 2)).float() / math.sqrt(hn)).view(b, np_, s, s).masked_fill(causal, -10000.0)
        probs = torch.softmax(scores, dim=-1).to(bf)
        ctx = torch.bmm(probs.view(b * np_, s, s), v).view(b, np_, s, hn).permute(2, 0, 1, 3).reshape(s, b, H)
        att = F.linear(ctx, sd[lp + "self_attention.dense.weight"]) + sd[lp + "self_attention.dense.bias"]
        h = h + att.float()
        x2 = restate.ln_fp32(h, sd[lp + "post_attention_layernorm.weight"], sd[lp + "post_attention_layernorm.bias"], cfg.gpt_ln_eps).to(bf)
        inter = restate.gelu_tanh(F.linear(x2, sd[lp + "mlp.dense_h_to_4h.weight"]) + sd[lp + "mlp.dense_h_to_4h.bias"])
        out = F.linear(inter, sd[lp + "mlp.dense_4h_to_h.weight"]) + sd[lp + "mlp.dense_4h_to_h.bias"]
        h = h + out.float()
    hf = restate.ln_fp32(h, sd[p + "encoder.final_layernorm.weight"], sd[p + "encoder.final_layernorm.bias"], cfg.gpt_ln_eps).to(bf)
    logits = F.linear(hf, sd[p + "embedding.word_embeddings.weight"])
    return dict(logits=logits.transpose(0, 1).contiguous(), last_hidden_state=hf.transpose(0, 1).contiguous())


def front(video, ids, sd, cfg):
    B = video.shape[0]
    image_embeds = restate.timesformer(video, sd, cfg)
    queries = sd["learnable_queries"].repeat(B, 1, 1)
    image_query = restate.attention_pool(queries, image_embeds, sd, cfg)
    qf = F.linear(image_query, sd["visual_fc.weight"], sd["visual_fc.bias"])
    emb = F.embedding(ids, sd["text_decoder.dist_model.language_model.embedding.word_embeddings.weight"])
    return torch.cat([qf, emb], dim=1)


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    torch.set_num_threads(os.cpu_count() or 8)
    cfg = CONFIG_B
    sd = make_state_dict(cfg, 11)
    video, ids, mask = make_inputs(cfg, B, 32, seed=31, ragged=True)
    s32 = {k: v.bfloat16().float() for k, v in sd.items()}
    s16 = {k: v.bfloat16() for k, v in sd.items()}
    Q = cfg.num_queries
    labels = torch.zeros(B, Q + 32, dtype=torch.long)
    lmask = torch.ones(B, Q + 31, dtype=torch.long)
    with torch.no_grad():
        t0 = time.time()
        e32 = front(video.bfloat16().float(), ids, s32, cfg)
        ref = restate.gpt_forward(e32, labels, lmask, s32, cfg)
        print(f"fp32 reference done in {time.time() - t0:.0f} s", flush=True)
        e16 = front(video.bfloat16(), ids, s16, cfg)
        arms = {
            "all bf16 (the reference's own bf16 run)": lambda: restate.gpt_forward(e16, labels, lmask, s16, cfg),
            "vision+abstractor fp32, decoder bf16": lambda: restate.gpt_forward(e32.bfloat16(), labels, lmask, s16, cfg),
            "vision+abstractor bf16, decoder fp32": lambda: restate.gpt_forward(e16.float(), labels, lmask, s32, cfg),
            "all bf16, fp32 residual stream in the decoder": lambda: gpt_forward_fp32_stream(e16, s16, cfg),
            "vision fp32, decoder bf16 with fp32 residual stream": lambda: gpt_forward_fp32_stream(e32.bfloat16(), s16, cfg),
        }
        print(f"query_features / input_embeds bf16 vs fp32: {rel(e16, e32):.3e}")
        for name, fn in arms.items():
            t0 = time.time()
            out = fn()
            print(f"{name:55s} logits {rel(out['logits'], ref['logits']):.3e}  hidden {rel(out['last_hidden_state'], ref['last_hidden_state']):.3e}   ({time.time() - t0:.0f} s)", flush=True)


if __name__ == "__main__":
    main()
