"""TEST INFRASTRUCTURE ONLY -- functional PyTorch restatement of the mPLUG-Video pre-train path.

This is the CPU oracle that travels to the GPU box (the reference tree does not).  It
restates, as plain functions over a state-dict, what the reference's nn.Modules compute;
each function cites the reference file:line it follows (paths relative to /root/reference).
It is validated against the reference's own modules (oracle/ref_loader.py) by
tests/test_oracle_vs_reference.py and by the committed goldens in tests/golden/.

Device rule: every function computes on the device of the tensors it is given (tensors it creates follow its inputs).  The CPU run is the
oracle; one test (config B at the benchmarked batch: sixteen fp32 slices) runs most slices of this same code on the GPU in fp32 -- torch's
fp32 ops, nothing of the product -- after checking on its first slice that the two placements agree.

Dtype rule: every function computes in the dtype of the tensors it is given, with the
same explicit fp32 up-casts the reference performs (LayerNormWithForceFP32, fp32 QK^T and
softmax in the ViT, fp32 cross-entropy), so running it in bf16 reproduces the
reference-as-run numerics and running it in fp32 gives the intended function.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from .weights import PathConfig


# --------------------------------------------------------------------------- primitives
def ln_fp32(x, w, b, eps):
    """models/vision_transformer.py:69-71 (LayerNormWithForceFP32) and the Megatron
    MixedFusedLayerNorm used at models/modeling_distributed_gpt3.py:1002,1016,1131:
    statistics and affine in fp32, result cast back to the input dtype."""
    return F.layer_norm(x.float(), (x.shape[-1],), w.float(), b.float(), eps).to(x.dtype)


def gelu_tanh(y):
    """megatron bias_gelu (models/modeling_distributed_gpt3.py:586-588): tanh approximation,
    evaluated in fp32 and rounded once."""
    yf = y.float()
    return (yf * 0.5 * (1.0 + torch.tanh(0.79788456 * yf * (1.0 + 0.044715 * yf * yf)))).to(y.dtype)


def vit_attention(x, sd, p, heads):
    """models/vision_transformer.py:169-207.  x [Bn, N, C]."""
    Bn, N, C = x.shape
    hd = C // heads
    bias = torch.cat([sd[p + "q_bias"], torch.zeros_like(sd[p + "v_bias"]), sd[p + "v_bias"]])   # :173
    qkv = F.linear(x, sd[p + "qkv.weight"], bias).reshape(Bn, N, 3, heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    q = q * (hd ** -0.5)                                                     # :179 (rounds in x.dtype)
    attn = q.float() @ k.float().transpose(-2, -1)                           # :181 fp32 scores
    attn = attn.softmax(dim=-1).to(x.dtype)                                  # :201
    out = (attn @ v).transpose(1, 2).reshape(Bn, N, C)                       # :204
    return F.linear(out, sd[p + "proj.weight"], sd[p + "proj.bias"])         # :205


def vit_mlp(x, sd, p):
    """models/vision_transformer.py:103-110 -- exact (erf) GELU."""
    h = F.gelu(F.linear(x, sd[p + "fc1.weight"], sd[p + "fc1.bias"]))
    return F.linear(h, sd[p + "fc2.weight"], sd[p + "fc2.bias"])


def vit_block(x, cls, sd, p, cfg: PathConfig):
    """models/vision_transformer.py:243-275 (divided space-time Block).
    x [B,T,N,D] frame-major tokens, cls [B,D]."""
    B, T, N, D = x.shape
    eps = cfg.vit_ln_eps
    # temporal attention over t for every (b, n)                                :247-251
    xt = x.permute(0, 2, 1, 3).reshape(B * N, T, D)
    xt = vit_attention(ln_fp32(xt, sd[p + "temporal_ln.weight"], sd[p + "temporal_ln.bias"], eps),
                       sd, p + "temporal_attn.", cfg.vit_heads)
    xt = F.linear(xt.reshape(B, N, T, D), sd[p + "temporal_fc.weight"], sd[p + "temporal_fc.bias"])
    xt = x.permute(0, 2, 1, 3) + xt                                          # [B,N,T,D] patch-major
    # spatial attention over n (+cls copy) for every (b, t)                     :254-267
    xs = xt.permute(0, 2, 1, 3).reshape(B * T, N, D)
    cls_rep = cls[:, None, :].expand(B, T, D).reshape(B * T, 1, D)
    xs = torch.cat([cls_rep, xs], dim=1)
    xs = vit_attention(ln_fp32(xs, sd[p + "norm1.weight"], sd[p + "norm1.bias"], eps),
                       sd, p + "attn.", cfg.vit_heads)
    cls_s = xs[:, 0, :].reshape(B, T, D).mean(dim=1, keepdim=True)           # :263-265
    xs = xs[:, 1:, :].reshape(B, T, N, D).permute(0, 2, 1, 3)                # -> [B,N,T,D]
    # residuals + MLP over [cls | tokens]                                      :270-271
    y = torch.cat([cls[:, None, :], xt.reshape(B, N * T, D)], dim=1) + \
        torch.cat([cls_s, xs.reshape(B, N * T, D)], dim=1)
    y = y + vit_mlp(ln_fp32(y, sd[p + "norm2.weight"], sd[p + "norm2.bias"], eps), sd, p + "mlp.")
    cls_out = y[:, 0, :]
    x_out = y[:, 1:, :].reshape(B, N, T, D).permute(0, 2, 1, 3)              # :274 back to b t n m
    return x_out, cls_out


def patch_embed_fold(video, w, patch):
    """models/vision_transformer.py:546-548,392-398: b c t h w -> (b t) c h w, conv k=s=P (no bias
    for CLIP), flatten -> b (t n) c."""
    B, C, T, H, W = video.shape
    x = video.permute(0, 2, 1, 3, 4).reshape(B * T, C, H, W)
    x = F.conv2d(x, w, None, stride=patch).flatten(2).transpose(1, 2)        # [(b t), n, D]
    return x.reshape(B, T * x.shape[1], x.shape[2])


def timesformer(video, sd, cfg: PathConfig, p="visual_encoder."):
    """models/vision_transformer.py:544-587.  Returns image_embeds [B, 1+T*N, D]."""
    B = video.shape[0]
    T, N, D = cfg.num_frames, cfg.n_patches, cfg.vit_dim
    x = patch_embed_fold(video, sd[p + "patch_embed.proj.weight"], cfg.patch_size)
    x = torch.cat([sd[p + "cls_token"].expand(B, -1, -1), x], dim=1)                     # :555-556
    pos = sd[p + "pos_embed"]
    tile_pos = pos[:, 1:, :].repeat(1, T, 1)                                              # :560
    tile_tmp = sd[p + "temporal_embed"].repeat_interleave(N, 1)                           # :562
    x = x + torch.cat([pos[:, :1, :], tile_pos + tile_tmp], dim=1)                        # :563-565
    x = ln_fp32(x, sd[p + "norm_pre.weight"], sd[p + "norm_pre.bias"], cfg.vit_ln_eps)     # :568-569
    cls, x = x[:, 0, :], x[:, 1:, :].reshape(B, T, N, D)                                  # :571-572
    for i in range(cfg.vit_depth):
        x, cls = vit_block(x, cls, sd, f"{p}blocks.{i}.", cfg)
    x = torch.cat([cls[:, None, :], x.reshape(B, T * N, D)], dim=1)                       # :582-584
    return ln_fp32(x, sd[p + "norm.weight"], sd[p + "norm.bias"], cfg.vit_ln_eps)          # :585


def attention_pool(queries, k, sd, cfg: PathConfig, p="attn_pool."):
    """models/vision_transformer.py:368-374 with nn.MultiheadAttention(add_bias_kv=True) (:353).
    queries [B,Q,D] (already repeated), k [B,S,D]."""
    eps, H = cfg.vit_ln_eps, cfg.vit_heads
    B, Q, D = queries.shape
    hd = D // H
    x = ln_fp32(queries, sd[p + "norm1.weight"], sd[p + "norm1.bias"], eps)
    kn = ln_fp32(k, sd[p + "normk.weight"], sd[p + "normk.bias"], eps)
    Wi, bi = sd[p + "attn.in_proj_weight"], sd[p + "attn.in_proj_bias"]
    q = F.linear(x, Wi[:D], bi[:D])
    kk = F.linear(kn, Wi[D:2 * D], bi[D:2 * D])
    vv = F.linear(kn, Wi[2 * D:], bi[2 * D:])
    kk = torch.cat([kk, sd[p + "attn.bias_k"].expand(B, 1, D)], dim=1)       # extra learned kv token
    vv = torch.cat([vv, sd[p + "attn.bias_v"].expand(B, 1, D)], dim=1)
    S1 = kk.shape[1]
    qh = q.reshape(B, Q, H, hd).transpose(1, 2)
    kh = kk.reshape(B, S1, H, hd).transpose(1, 2)
    vh = vv.reshape(B, S1, H, hd).transpose(1, 2)
    # torch MHA (math path): q scaled by 1/sqrt(hd), softmax in the tensor dtype
    att = torch.softmax((qh * (1.0 / math.sqrt(hd))) @ kh.transpose(-2, -1), dim=-1)
    o = (att @ vh).transpose(1, 2).reshape(B, Q, D)
    o = F.linear(o, sd[p + "attn.out_proj.weight"], sd[p + "attn.out_proj.bias"])
    x = x + o                                                                # residual from NORMED x (:369-371)
    return x + vit_mlp(ln_fp32(x, sd[p + "norm2.weight"], sd[p + "norm2.bias"], eps), sd, p + "mlp.")


def gpt_layer(h, sd, p, layer_number, cfg: PathConfig, causal_mask, drop=None):
    """models/modeling_distributed_gpt3.py:1034-1078 (pre-LN layer) with :868-938 attention and
    :734-817 core attention.  h [s,b,H].  Dropout: off (eval mode) unless `drop` supplies the three multipliers of the
    layer -- keep / (1 - p) or 0 per element, i.e. torch.nn.functional.dropout with the MASK GIVEN instead of drawn:
    "attn" [b,np,s,s] on the softmax output (:779-782), "h1" [s,b,H] on attention-dense output + bias (:1059-1062,
    bias_dropout_add), "h2" [s,b,H] on the MLP output + bias (:1075-1078)."""
    s, b, H = h.shape
    np_, hn = cfg.heads, cfg.head_dim
    x = ln_fp32(h, sd[p + "input_layernorm.weight"], sd[p + "input_layernorm.bias"], cfg.gpt_ln_eps)
    mixed = F.linear(x, sd[p + "self_attention.query_key_value.weight"],
                     sd[p + "self_attention.query_key_value.bias"]).view(s, b, np_, 3 * hn)      # :892-898
    q, k, v = torch.split(mixed, hn, dim=-1)                                                     # :901-902
    q = q.reshape(s, b * np_, hn).transpose(0, 1)
    k = k.reshape(s, b * np_, hn).transpose(0, 1)
    v = v.reshape(s, b * np_, hn).transpose(0, 1)
    norm = math.sqrt(hn) * layer_number                                                          # :719-722
    scores = torch.baddbmm(torch.zeros(b * np_, s, s, dtype=q.dtype, device=q.device), q, k.transpose(1, 2),
                           beta=0.0, alpha=1.0 / norm).view(b, np_, s, s)                        # :757-765
    scores = scores * layer_number                                                               # coeff, :727
    scores = scores.masked_fill(causal_mask, -10000.0)                                           # :684-686
    probs = torch.softmax(scores, dim=-1)
    if drop is not None:
        probs = probs * drop["attn"].to(probs.dtype)
    ctx = torch.bmm(probs.view(b * np_, s, s), v).view(b, np_, s, hn).permute(2, 0, 1, 3).reshape(s, b, H)
    att = F.linear(ctx, sd[p + "self_attention.dense.weight"])                                   # bias returned separately
    a1 = att + sd[p + "self_attention.dense.bias"]
    if drop is not None:
        a1 = a1 * drop["h1"].to(a1.dtype)
    h1 = h + a1                                                                                  # :1059-1062
    x2 = ln_fp32(h1, sd[p + "post_attention_layernorm.weight"], sd[p + "post_attention_layernorm.bias"],
                 cfg.gpt_ln_eps)
    inter = F.linear(x2, sd[p + "mlp.dense_h_to_4h.weight"])
    inter = gelu_tanh(inter + sd[p + "mlp.dense_h_to_4h.bias"])                                  # :586-588
    out = F.linear(inter, sd[p + "mlp.dense_4h_to_h.weight"])
    o2 = out + sd[p + "mlp.dense_4h_to_h.bias"]
    if drop is not None:
        o2 = o2 * drop["h2"].to(o2.dtype)
    return h1 + o2                                                                               # :1075-1078


def gpt_forward(input_embeds, labels, loss_mask, sd, cfg: PathConfig,
                p="text_decoder.dist_model.language_model.", drop=None):
    """models/modeling_distributed_gpt3.py:1309-1366 + :1589-1618.  input_embeds [B,S,H].  drop (train mode with GIVEN masks):
    {"embed": [B,S,H] multiplier of the embedding dropout (:665), "layers": [gpt_layer's dict per layer]}."""
    B, S, H = input_embeds.shape
    pos = sd[p + "embedding.position_embeddings.weight"][:S]                                     # :1294-1296,649
    e = input_embeds + pos[None]
    if drop is not None:
        e = e * drop["embed"].to(e.dtype)
    h = e.transpose(0, 1).contiguous()                                                           # :650-653
    causal = torch.tril(torch.ones(1, 1, S, S, device=input_embeds.device)) < 0.5                # :1288-1292
    for i in range(cfg.layers):
        h = gpt_layer(h, sd, f"{p}encoder.layers.{i}.", i + 1, cfg, causal, None if drop is None else drop["layers"][i])
    h = ln_fp32(h, sd[p + "encoder.final_layernorm.weight"], sd[p + "encoder.final_layernorm.bias"],
                cfg.gpt_ln_eps)                                                                  # :1184
    logits = F.linear(h, sd[p + "embedding.word_embeddings.weight"])                             # :1348-1350 (tied)
    lf = logits.float()                                                                          # :1357
    m = lf.max(dim=-1, keepdim=True)[0]
    z = lf - m
    tgt = z.gather(-1, labels.transpose(0, 1).unsqueeze(-1)).squeeze(-1)
    losses = (torch.log(z.exp().sum(-1)) - tgt).transpose(0, 1).contiguous()                     # [B,S]
    losses = losses[:, :-1].contiguous().float()                                                 # :1615
    lm = loss_mask.reshape(-1).float()
    loss = torch.sum(losses.reshape(-1) * lm) / lm.sum()                                         # :1616-1617
    return dict(loss=loss, losses=losses, logits=logits.transpose(0, 1).contiguous(),
                last_hidden_state=h.transpose(0, 1).contiguous())


def visual_connect(image_query, sd):
    """models/distributed_gpt3.py:136: visual_norm(visual_fc(image_query)); visual_norm is LayerNormWithForceFP32(eps 1e-6) when the
    visual config sets connect_ln (:112-115: its parameters are then in the state dict), the identity otherwise."""
    qf = F.linear(image_query, sd["visual_fc.weight"], sd["visual_fc.bias"])
    if "visual_norm.weight" in sd:
        qf = ln_fp32(qf, sd["visual_norm.weight"], sd["visual_norm.bias"], 1e-6)
    return qf


def pretrain_forward(video, ids, attn_mask, sd, cfg: PathConfig, drop=None):
    """models/distributed_gpt3.py:130-166 (use_contrastive False).  drop: the decoder's dropout multipliers (gpt_forward) -- train
    mode with given masks; the vision tower and the abstractor have no active dropout (drop_rate 0, clip-b16.json:9-10)."""
    B = video.shape[0]
    image_embeds = timesformer(video, sd, cfg)
    queries = sd["learnable_queries"].repeat(B, 1, 1)                                            # :134
    image_query = attention_pool(queries, image_embeds, sd, cfg)
    query_features = visual_connect(image_query, sd)                                             # :136
    Q = query_features.shape[1]
    targets = torch.cat([ids[:, 1:], ids[:, 1:2]], dim=1)                                        # :142-143
    targets = torch.cat([torch.full((B, Q), 100, dtype=torch.long, device=ids.device), targets], dim=1)      # :150-153
    emb = F.embedding(ids, sd["text_decoder.dist_model.language_model.embedding.word_embeddings.weight"])
    input_embeds = torch.cat([query_features, emb], dim=1)                                       # :155-156
    loss_mask = torch.cat([torch.zeros(B, Q, dtype=torch.long, device=attn_mask.device), attn_mask[:, 1:]], dim=1)        # :145,159
    out = gpt_forward(input_embeds, targets, loss_mask, sd, cfg, drop=drop)
    out.update(image_embeds=image_embeds, image_query=image_query, query_features=query_features,
               input_embeds=input_embeds)
    return out


def retrieval_forward(video, ids, attn_mask, idx, sd, cfg: PathConfig):
    """models/distributed_gpt3.py:938-980 at world size 1 (all_gather of one rank is the identity)."""
    B = video.shape[0]
    image_embeds = timesformer(video, sd, cfg)
    image_query = image_embeds[:, 0]                                                              # :939 pooled cls
    vision_feats = F.normalize(F.linear(image_query, sd["vision_proj.weight"], sd["vision_proj.bias"]), dim=-1)   # :947
    emb = F.embedding(ids, sd["text_decoder.dist_model.language_model.embedding.word_embeddings.weight"])
    targets = torch.cat([ids[:, 1:], ids[:, 1:2]], dim=1)                                        # :949-950
    out = gpt_forward(emb, targets, attn_mask[:, 1:], sd, cfg)                                    # :952-956
    hid = out["last_hidden_state"]
    pooled = hid[torch.arange(B), attn_mask.sum(dim=-1) - 1]                                      # :958-959
    text_feat = F.normalize(F.linear(pooled, sd["text_proj.weight"], sd["text_proj.bias"]), dim=-1)        # :960
    sim_i2t = vision_feats @ text_feat.t() / sd["temp"]                                           # :966
    sim_t2i = text_feat @ vision_feats.t() / sd["temp"]                                           # :967
    idx = idx.view(-1, 1)
    pos = torch.eq(idx, idx.t()).float()
    tg = pos / pos.sum(1, keepdim=True)                                                           # :970-972
    loss_i2t = -torch.sum(F.log_softmax(sim_i2t.float(), dim=1) * tg, dim=1).mean()
    loss_t2i = -torch.sum(F.log_softmax(sim_t2i.float(), dim=1) * tg, dim=1).mean()
    return dict(loss=(loss_i2t + loss_t2i) / 2, vision_feats=vision_feats, text_feat=text_feat)


def eva_vit(image, sd, cfg: PathConfig, p="visual_encoder."):
    """models/eva_vit.py:338-350 (forward_features) with Block :169-174 and Attention :123-153, as built by
    create_eva_vit_g (:413-427): conv patch embed with bias, cls + absolute positions, pre-LN blocks with q/v bias,
    no relative-position bias, no LayerScale, final LayerNorm (eps 1e-6, models/distributed_gpt3.py:258)."""
    B = image.shape[0]
    D, heads = cfg.vit_dim, cfg.vit_heads
    x = F.conv2d(image, sd[p + "patch_embed.proj.weight"], sd[p + "patch_embed.proj.bias"], stride=cfg.patch_size)
    x = x.flatten(2).transpose(1, 2)                                                              # :197
    x = torch.cat([sd[p + "cls_token"].expand(B, -1, -1), x], dim=1) + sd[p + "pos_embed"]       # :341-345
    scale = (D // heads) ** -0.5
    for i in range(cfg.vit_depth):
        b = f"{p}blocks.{i}."
        h = ln_fp32(x, sd[b + "norm1.weight"], sd[b + "norm1.bias"], 1e-6)
        bias = torch.cat([sd[b + "attn.q_bias"], torch.zeros_like(sd[b + "attn.v_bias"]), sd[b + "attn.v_bias"]])   # :127
        qkv = F.linear(h, sd[b + "attn.qkv.weight"], bias).reshape(B, -1, 3, heads, D // heads).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0] * scale, qkv[1], qkv[2]                                                  # :134
        a = (q @ k.transpose(-2, -1)).softmax(dim=-1)                                            # :135-148
        a = (a @ v).transpose(1, 2).reshape(B, -1, D)
        x = x + F.linear(a, sd[b + "attn.proj.weight"], sd[b + "attn.proj.bias"])                 # :172
        h = ln_fp32(x, sd[b + "norm2.weight"], sd[b + "norm2.bias"], 1e-6)
        h = F.linear(F.gelu(F.linear(h, sd[b + "mlp.fc1.weight"], sd[b + "mlp.fc1.bias"])), sd[b + "mlp.fc2.weight"], sd[b + "mlp.fc2.bias"])
        x = x + h                                                                                 # :173
    return ln_fp32(x, sd[p + "norm.weight"], sd[p + "norm.bias"], 1e-6)                           # :350


def pretrain_image_forward(image, ids, attn_mask, sd, cfg: PathConfig, prompt_lengths=None):
    """DistributedGPT3_Pretrain_Image.forward with use_eva_g, use_contrastive False (models/distributed_gpt3.py:334-372)."""
    B = image.shape[0]
    image_embeds = eva_vit(image, sd, cfg)
    image_query = attention_pool(sd["learnable_queries"].repeat(B, 1, 1), image_embeds, sd, cfg)
    query_features = visual_connect(image_query, sd)
    Q = query_features.shape[1]
    targets = torch.cat([ids[:, 1:], ids[:, 1:2]], dim=1)
    tla = attn_mask[:, 1:].clone()
    if prompt_lengths is not None:                                                                # :348-351
        for i, pl in enumerate(prompt_lengths):
            tla[i, :pl] = 0
    targets = torch.cat([torch.full((B, Q), 100, dtype=torch.long), targets], dim=1)
    emb = F.embedding(ids, sd["text_decoder.dist_model.language_model.embedding.word_embeddings.weight"])
    out = gpt_forward(torch.cat([query_features, emb], dim=1), targets, torch.cat([torch.zeros(B, Q, dtype=torch.long), tla], dim=1), sd, cfg)
    out.update(image_embeds=image_embeds, query_features=query_features)
    return out


def gencls_forward(video, ids, attn_mask, prompt_lengths, prompt_ids, prompt_mask, labels, sd, cfg: PathConfig,
                   negative_indices=None, train=True, kind="itm"):
    """DistributedGPT3_Retrieval_Cls.forward (kind="itm", models/distributed_gpt3.py:1087-1214) and
    DistributedGPT3_Cls.forward (kind="cls", :532-653) with use_cls on: generation pass with a per-sample
    prompt-length loss mask + a prompt pass whose last valid hidden state feeds cls_head."""
    Bv = video.shape[0]
    image_embeds = timesformer(video, sd, cfg)
    image_query = attention_pool(sd["learnable_queries"].repeat(Bv, 1, 1), image_embeds, sd, cfg)
    qf = F.linear(image_query, sd["visual_fc.weight"], sd["visual_fc.bias"])                      # :1093 / :538
    Q = qf.shape[1]
    wte = sd["text_decoder.dist_model.language_model.embedding.word_embeddings.weight"]
    targets = torch.cat([ids[:, 1:], ids[:, 1:2]], dim=1)                                         # :1097-1098
    tla = attn_mask[:, 1:].clone()
    for i, pl in enumerate(prompt_lengths):                                                       # :1100-1102
        tla[i, :pl] = 0

    def head(x):
        return F.linear(F.relu(F.linear(x, sd["cls_head.0.weight"], sd["cls_head.0.bias"])), sd["cls_head.2.weight"], sd["cls_head.2.bias"])

    def pooled_of(qfeat, p_ids, p_mask):
        n = p_ids.shape[0]
        tt = torch.cat([p_ids[:, 1:], p_ids[:, 1:2]], dim=1)
        tt = torch.cat([torch.full((n, Q), 100, dtype=torch.long), tt], dim=1)
        emb = torch.cat([qfeat, F.embedding(p_ids, wte)], dim=1)
        lm_p = torch.cat([torch.zeros(n, Q, dtype=torch.long), p_mask[:, 1:]], dim=1)             # value unused downstream
        out = gpt_forward(emb, tt, lm_p, sd, cfg)
        am = torch.cat([torch.ones(n, Q, dtype=torch.long), p_mask], dim=1)
        return out["last_hidden_state"][torch.arange(n), am.sum(dim=-1) - 1]                     # :1149-1150

    if train:
        if kind == "itm":
            qf = torch.cat([qf, qf[negative_indices]], dim=0)                                    # :1105-1108
        n = qf.shape[0]
        tg = torch.cat([torch.full((n, Q), 100, dtype=torch.long), targets], dim=1)
        emb = torch.cat([qf, F.embedding(ids, wte)], dim=1)
        loss_mask = torch.cat([torch.zeros(n, Q, dtype=torch.long), tla], dim=1)
        out = gpt_forward(emb, tg, loss_mask, sd, cfg)
        logits = head(pooled_of(qf, prompt_ids, prompt_mask))
        return dict(loss_caption=out["loss"], loss_cls=F.cross_entropy(logits.float(), labels), cls_logits=logits)
    t = ids.shape[0] // Bv
    qf_rep = qf.unsqueeze(1).repeat(1, t, 1, 1).reshape(Bv * t, Q, -1)                           # :1158-1159 / :599-600
    n = qf_rep.shape[0]
    tg = torch.cat([torch.full((n, Q), 100, dtype=torch.long), targets], dim=1)
    emb = torch.cat([qf_rep, F.embedding(ids, wte)], dim=1)
    loss_mask = torch.cat([torch.zeros(n, Q, dtype=torch.long), tla], dim=1)
    out = gpt_forward(emb, tg, loss_mask, sd, cfg)
    gen = (-torch.sum(out["losses"] * loss_mask, dim=-1)).view(Bv, t)                             # :1180-1181
    if kind == "cls":
        gen = torch.softmax(gen, dim=-1)                                                          # :620
        cls_logits = head(pooled_of(qf, prompt_ids, prompt_mask))                                 # per video (:622-647)
    else:
        cls_logits = torch.softmax(head(pooled_of(qf_rep, prompt_ids, prompt_mask)), dim=-1)[:, 1].view(Bv, t)   # :1207-1208
    return dict(generation_logits=gen, cls_logits=cls_logits)


# --------------------------------------------------------------------------- optimizer
def adamw_step(p, g, m, v, step, lr, beta1, beta2, eps, wd):
    """optim/adamw.py:66-115 (decoupled decay first, then bias-corrected Adam); in place, fp32."""
    p.mul_(1.0 - lr * wd)
    m.mul_(beta1).add_(g, alpha=1.0 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1.0 - beta2)
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-(lr / bc1))


def param_group_of(name: str, shape, skip=("visual_encoder.pos_embed", "visual_encoder.cls_token",
                                           "visual_encoder.temporal_embed")):
    """optim/optim_factory.py:219-265 with visual_backbone_scale=True (clip_model):
    returns (group_name, lr_scale, decays)."""
    no_decay = len(shape) == 1 or name.endswith(".bias") or name in skip or \
        "bias" in name or "LayerNorm.weight" in name
    vis = name.startswith("visual_encoder.") and "temporal" not in name
    g = ("visual_encoder_" if vis else "") + ("no_decay" if no_decay else "decay")
    return g, (0.1 if vis else 1.0), (not no_decay)


def cosine_schedule(base, final, total_steps, warmup_steps, start_warmup=0.0):
    """utils.py:350-372: linear warm-up then cosine to `final`."""
    import numpy as np
    warm = np.linspace(start_warmup, base, warmup_steps) if warmup_steps > 0 else np.array([])
    it = np.arange(total_steps - warmup_steps)
    sched = np.array([final + 0.5 * (base - final) * (1 + math.cos(math.pi * i / len(it))) for i in it])
    return np.concatenate((warm, sched))
