"""Torch stand-ins for the device entry points the pre-train pipelines call (youku-mplug_amd/ops.py), TEST INFRASTRUCTURE
ONLY: they let the real host pipelines (vision.py / gpt3.py / pretrain.py: row maps, strides, tapes, the loss window, the
composed temporal-projection backward) run on CPU inside `-m "not gpu"` tests and be checked against the reference goldens.
Each function restates the contract of its C entry point in include/mpv.h (same arguments as the ops.py wrapper, bf16
storage, fp32 arithmetic, bf16 rounding at the points the kernels round) with plain indexing -- no attempt at speed, tiny
shapes only.  Dropout is not modelled (the counter-hash masks live in the kernels): every stand-in asserts dropout_p == 0.
Nothing in the product imports this file; tests install it with `install(monkeypatch)`."""
import math

import torch
import torch.nn.functional as F

BF = torch.bfloat16


def _flat(t):
    """1-D view of t's storage from t's first element on (what a raw device pointer sees)."""
    n = t.untyped_storage().nbytes() // t.element_size() - t.storage_offset()
    return t.as_strided((n,), (1,), t.storage_offset())


def _map(m, r):
    g, s, o = m
    return r if g == 0 else (r // g) * s + (r % g) + o


def _rows(m, n):
    return _map(m, torch.arange(n, dtype=torch.int64))


def _rd(t, rows, ld, cols):
    """fp32 [len(rows), cols] gathered from the row-major buffer behind t (row pitch ld)."""
    idx = rows[:, None] * ld + torch.arange(cols, dtype=torch.int64)[None, :]
    return _flat(t)[idx].float()


def _wr(t, rows, ld, val):
    idx = rows[:, None] * ld + torch.arange(val.shape[1], dtype=torch.int64)[None, :]
    _flat(t)[idx] = val.to(t.dtype)


def _r16(x):
    return x.to(BF).float()


ACT_NONE, ACT_GELU_ERF, ACT_GELU_TANH, ACT_RELU = 0, 1, 2, 3
IDENT = (0, 0, 0)


def _act(v, kind):
    if kind == ACT_GELU_ERF:
        return F.gelu(v)
    if kind == ACT_GELU_TANH:
        return F.gelu(v, approximate="tanh")
    return F.relu(v)


def _act_grad(z, kind):
    with torch.enable_grad():      # called from inside an autograd.Function's backward, where grad mode is off
        z = z.detach().clone().requires_grad_(True)
        (g,) = torch.autograd.grad(_act(z, kind).sum(), z)
    return g


gemm_calls = 0


def gemm(a, b, M, N, K, *, out=None, trans_a=False, trans_b=False, lda=None, ldb=None, ldc=None, bias=None, act=0, preact_out=None,
         residual=None, ldr=0, act_bwd_z=None, act_bwd=0, ldz=0, dropout_p=0.0, seed=0, offset=0, alpha_dev=None, alpha=0.0,
         amap=IDENT, cmap=IDENT, kmap=IDENT, out_rows=None, accumulate=False, out_f32=False, colsum_out=None, tile_hint=0,
         row_tap_out=None, row_tap_group=0, split_hint=0, gm_hint=0, preact_deriv=False, z_is_deriv=False, keep_output=False, colscale=None):
    """include/mpv.h mpv_gemm_bf16: C[M,N] = epilogue(sum_k A(m,k) B(n,k)).  preact_deriv / z_is_deriv: the product's own pair of
    switches (ops.gemm): preact_out receives bf16(act'(zb)) and the dgrad multiplies by that tensor (MPV_ACT_DERIV)."""
    from youku_mplug_amd import ops as _ops
    deriv_on = _ops.GELU_DERIV_FWD
    assert dropout_p == 0.0, "stand-ins do not model the hash dropout"
    assert not (trans_a and not trans_b)
    lda = lda if lda is not None else (M if trans_a else K)
    ldb = ldb if ldb is not None else (N if trans_b else K)
    ldc = ldc if ldc is not None else N
    if out is None:
        out = torch.zeros((out_rows if out_rows is not None else M, ldc), dtype=torch.float32 if out_f32 else BF, device=a.device)
    kr = _rows(kmap, K)
    A = _rd(a, kr, lda, M).t() if trans_a else _rd(a, _rows(amap, M), lda, K)          # [M, K]
    Bm = _rd(b, kr, ldb, N) if trans_b else _rd(b, torch.arange(N), ldb, K).t()        # [K, N]
    acc = A @ Bm
    al = (alpha if alpha != 0.0 else 1.0) * (float(alpha_dev.float().item()) if alpha_dev is not None else 1.0)
    crow = _rows(cmap, M)
    if colsum_out is not None:
        assert trans_a and trans_b and not out_f32
        _flat(colsum_out)[:M] = A.sum(1).to(BF)
    if out_f32:
        v = acc * al
        if accumulate:
            v = v + _rd(out, crow, ldc, N)
        _wr(out, crow, ldc, v)
        return out
    z = acc * al
    if bias is not None:
        z = z + _flat(bias)[:N].float()[None, :]
    zb = _r16(z)
    if colscale is not None:          # mpv.h colscale: columns below ncols are rounded, scaled, rounded again
        assert not (act or act_bwd or residual is not None or preact_out is not None or accumulate or row_tap_out is not None)
        zb = zb.clone()
        zb[:, :colscale[0]] = _r16(zb[:, :colscale[0]] * colscale[1])
    v = zb
    if row_tap_out is not None:
        sel = torch.arange(0, M, row_tap_group)
        _wr(row_tap_out, torch.arange(len(sel)), N, zb[sel])
    if act:
        if preact_out is not None:
            _wr(preact_out, crow, ldc, _act_grad(zb, act) if (preact_deriv and deriv_on) else zb)
        v = _act(v, act)
    if act_bwd_z is not None and act_bwd:
        zz = _rd(act_bwd_z, torch.arange(M), ldz if ldz else N, N)
        v = v * (zz if (z_is_deriv and deriv_on) else _act_grad(zz, act_bwd))
    if residual is not None:
        v = v + _rd(residual, crow, ldr if ldr else ldc, N)
    if accumulate:
        v = v + _rd(out, crow, ldc, N)
    _wr(out, crow, ldc, v)
    return out


def layernorm_fwd(x, gamma, beta, eps, rows, cols, *, out=None, xmap=IDENT, ymap=IDENT, out_rows=None, ldx=None, ldy=None,
                  want_stats=True):
    ldx, ldy = ldx or cols, ldy or cols
    if out is None:
        out = torch.zeros((out_rows if out_rows is not None else rows, ldy), dtype=BF, device=x.device)
    xv = _rd(x, _rows(xmap, rows), ldx, cols)
    mu = xv.mean(1)
    rs = torch.rsqrt(((xv - mu[:, None]) ** 2).mean(1) + eps)
    y = (xv - mu[:, None]) * rs[:, None] * gamma.float()[None, :] + beta.float()[None, :]
    _wr(out, _rows(ymap, rows), ldy, y)
    return out, (mu if want_stats else None), (rs if want_stats else None)


class LnDparamBatch:
    """Deferred dgamma / dbeta reductions (mpv_layernorm_dparam_finish): the partial is held as one fp32 row pair."""

    def __init__(self):
        self._pending = []

    def finish(self):
        for dg, db, dgamma, dbeta, acc in self._pending:
            if acc:
                dg, db = dg + dgamma.float().view(-1), db + dbeta.float().view(-1)
            dgamma.view(-1).copy_(dg.to(BF))
            dbeta.view(-1).copy_(db.to(BF))
        self._pending = []


def layernorm_bwd(dy, x, gamma, mean, rstd, rows, cols, *, dres=None, dx=None, dx_drop=None, dropout_p=0.0, seed=0, offset=0,
                  dgamma=None, dbeta=None, accumulate_dparams=False, xmap=IDENT, ymap=IDENT, ldx=None, ldy=None, dx_rows=None,
                  defer=None):
    assert dropout_p == 0.0
    ldx, ldy = ldx or cols, ldy or cols
    if dx is None:
        dx = torch.zeros((dx_rows if dx_rows is not None else rows, ldx), dtype=BF, device=x.device)
    xr = _rows(xmap, rows)
    xv, dv = _rd(x, xr, ldx, cols), _rd(dy, _rows(ymap, rows), ldy, cols)
    xh = (xv - mean[:rows, None]) * rstd[:rows, None]
    g = dv * gamma.float()[None, :]
    c1, c2 = g.mean(1, keepdim=True), (g * xh).mean(1, keepdim=True)
    o = rstd[:rows, None] * (g - c1 - xh * c2)
    if dres is not None:
        o = o + _rd(dres, xr, ldx, cols)
    _wr(dx, xr, ldx, o)
    if dx_drop is not None:
        _wr(dx_drop, xr, ldx, _r16(o))
    if dgamma is not None and defer is not None:       # MPV_LN_DPARAM_DEFER: nothing is written until the batch's finish()
        defer._pending.append(((dv * xh).sum(0), dv.sum(0), dgamma, dbeta, bool(accumulate_dparams)))
    elif dgamma is not None:
        dg, db = (dv * xh).sum(0), dv.sum(0)
        if accumulate_dparams:
            dg, db = dg + dgamma.float().view(-1), db + dbeta.float().view(-1)
        dgamma.view(-1).copy_(dg.to(BF))
        dbeta.view(-1).copy_(db.to(BF))
    return dx


def ln_stream_fwd(h_in, add, gamma, beta, eps, rows, cols, *, add_dropout_p=0.0, seed=0, offset=0, h_out=None, out=None, hmap=IDENT, amap=IDENT, ymap=IDENT, out_rows=None,
                  h_rows=None):
    assert add_dropout_p == 0.0, "stand-ins do not model the hash dropout"
    if out is None:
        out = torch.zeros((out_rows if out_rows is not None else rows, cols), dtype=BF)
    hr = _rows(hmap, rows)
    v = _rd(h_in, hr, cols, cols)                      # fp32 (or bf16 for the first LayerNorm) -> fp32
    if add is not None:
        v = v + _rd(add, _rows(amap, rows), cols, cols)
        if h_out is None:
            h_out = torch.zeros((h_rows if h_rows is not None else h_in.shape[0], cols), dtype=torch.float32)
        _wr(h_out, hr, cols, v)
    mu = v.mean(1)
    rs = torch.rsqrt(((v - mu[:, None]) ** 2).mean(1) + eps)
    _wr(out, _rows(ymap, rows), cols, (v - mu[:, None]) * rs[:, None] * gamma.float()[None, :] + beta.float()[None, :])
    return out, (h_out if add is not None else None), mu, rs


def ln_stream_bwd(dy, x, gamma, mean, rstd, rows, cols, *, dres=None, dx=None, dx_drop=None, dropout_p=0.0, seed=0, offset=0, xmap=IDENT,
                  ymap=IDENT, dx_rows=None):
    assert x.dtype == torch.float32
    return layernorm_bwd(dy, x, gamma, mean, rstd, rows, cols, dres=dres, dx=dx, dx_drop=dx_drop, dropout_p=dropout_p, seed=seed, offset=offset,
                         xmap=xmap, ymap=ymap, dx_rows=dx_rows)


class AttnLayout:
    def __init__(self, q, k, v, o):
        self.q, self.k, self.v, self.o = q, k, v, o


def _bhrd(t, st, batch, heads, rows, hd):
    return t.as_strided((batch, heads, rows, hd), (st[0], st[1], st[2], 1), t.storage_offset())


def _scores(q, k, lay, batch, heads, sq, sk, hd, causal, scale, scale_q_bf16):
    qf = _bhrd(q, lay.q, batch, heads, sq, hd).float()
    kf = _bhrd(k, lay.k, batch, heads, sk, hd).float()
    if scale_q_bf16 == 2:              # q arrives as bf16(q * scale) already (the qkv product's colscale epilogue)
        sc = 1.0
    elif scale_q_bf16:
        qf, sc = _r16(qf * scale), 1.0
    else:
        sc = scale
    s = (qf @ kf.transpose(-1, -2)) * sc
    if causal:
        i, j = torch.arange(sq)[:, None], torch.arange(sk)[None, :]
        s = s.masked_fill(j > i + (sk - sq), float("-inf"))
    return qf, kf, s, sc


def attn_fwd(q, k, v, o, lay, batch, heads, sq, sk, hd, *, causal=False, scale=1.0, scale_q_bf16=False, dropout_p=0.0, seed=0,
             offset=0):
    assert dropout_p == 0.0
    _, _, s, _ = _scores(q, k, lay, batch, heads, sq, sk, hd, causal, scale, scale_q_bf16)
    m = s.max(-1, keepdim=True).values
    p = torch.exp(s - m)
    l = p.sum(-1, keepdim=True)
    vf = _bhrd(v, lay.v, batch, heads, sk, hd).float()
    _bhrd(o, lay.o, batch, heads, sq, hd).copy_(((_r16(p) @ vf) / l).to(BF))
    return (m + torch.log(l)).squeeze(-1).contiguous()


def attn_bwd(q, k, v, o, lse, do, dq, dk, dv, lay, batch, heads, sq, sk, hd, *, causal=False, scale=1.0, scale_q_bf16=False,
             dropout_p=0.0, seed=0, offset=0):
    assert dropout_p == 0.0
    qf, kf, s, sc = _scores(q, k, lay, batch, heads, sq, sk, hd, causal, scale, scale_q_bf16)
    vf = _bhrd(v, lay.v, batch, heads, sk, hd).float()
    of = _bhrd(o, lay.o, batch, heads, sq, hd).float()
    dof = _bhrd(do, lay.o, batch, heads, sq, hd).float()
    p = torch.exp(s - lse.view(batch, heads, sq, 1))
    dp = dof @ vf.transpose(-1, -2)
    delta = (dof * of).sum(-1, keepdim=True)
    ds = p * (dp - delta)
    pb, dsb = _r16(p), _r16(ds)
    _bhrd(dv, lay.v, batch, heads, sk, hd).copy_((pb.transpose(-1, -2) @ dof).to(BF))
    _bhrd(dk, lay.k, batch, heads, sk, hd).copy_(((dsb.transpose(-1, -2) @ qf) * sc).to(BF))
    _bhrd(dq, lay.q, batch, heads, sq, hd).copy_(((dsb @ kf) * scale).to(BF))


def _temporal_views(qkv, n_outer, outer_stride, n_inner, inner_offset, t_stride, T, heads, hd):
    D = heads * hd
    o, i, t = torch.meshgrid(torch.arange(n_outer), torch.arange(n_inner), torch.arange(T), indexing="ij")
    rows = (o * outer_stride + inner_offset + i + t * t_stride).reshape(-1)              # (o, i, t)
    x = _rd(qkv, rows, 3 * D, 3 * D).view(n_outer * n_inner, T, 3, heads, hd)
    return rows, D, x[:, :, 0].transpose(1, 2), x[:, :, 1].transpose(1, 2), x[:, :, 2].transpose(1, 2)   # [S, heads, T, hd]


def temporal_attn_fwd(qkv, out, n_outer, outer_stride, n_inner, inner_offset, t_stride, T, heads, hd, scale):
    rows, D, q, k, v = _temporal_views(qkv, n_outer, outer_stride, n_inner, inner_offset, t_stride, T, heads, hd)
    q = _r16(q * scale)
    p = _r16(torch.softmax(q @ k.transpose(-1, -2), dim=-1))
    o = (p @ v).transpose(1, 2).reshape(-1, D)                                            # rows (s, t), cols (head, hd)
    _wr(out, rows, D, o)


def temporal_attn_bwd(qkv, dout, dqkv, n_outer, outer_stride, n_inner, inner_offset, t_stride, T, heads, hd, scale):
    rows, D, q, k, v = _temporal_views(qkv, n_outer, outer_stride, n_inner, inner_offset, t_stride, T, heads, hd)
    do = _rd(dout, rows, D, D).view(-1, T, heads, hd).transpose(1, 2)
    q = _r16(q * scale)
    p = torch.softmax(q @ k.transpose(-1, -2), dim=-1)
    dp = do @ v.transpose(-1, -2)
    ds = p * (dp - (p * dp).sum(-1, keepdim=True))
    dq, dk, dv = (ds @ k) * scale, ds.transpose(-1, -2) @ q, p.transpose(-1, -2) @ do
    pack = torch.stack([dq, dk, dv], dim=1)                                               # [S, 3, heads, T, hd]
    _wr(dqkv, rows, 3 * D, pack.permute(0, 3, 1, 2, 4).reshape(-1, 3 * D))


def im2col_patches(video, B, Cc, T, H, W, P, kpad):
    v = video.float().view(B, Cc, T, H // P, P, W // P, P).permute(0, 2, 3, 5, 1, 4, 6).reshape(B * T * (H // P) * (W // P), Cc * P * P)
    cols = torch.zeros((v.shape[0], kpad), dtype=BF, device=video.device)
    cols[:, :v.shape[1]] = v.to(BF)
    return cols


def vit_embed_assemble_fwd(patch, cls_token, pos_embed, temporal_embed, B, T, N, D):
    pos, tem = pos_embed.float().view(N + 1, D), temporal_embed.float().view(-1, D)[:T]
    x = torch.empty((B, T, N + 1, D), dtype=torch.float32)
    x[:, :, 0] = cls_token.float().view(1, 1, D) + pos[0]
    x[:, :, 1:] = patch.float().view(B, T, N, D) + pos[1:][None, None] + tem[None, :, None]
    return x.view(-1, D).to(BF)


def vit_embed_assemble_bwd(dx, dpatch, dcls, dpos, dtemporal, B, T, N, D):
    d = dx.float().view(B, T, N + 1, D)
    dpatch.view(B, T, N, D).copy_(d[:, :, 1:].to(BF))
    dcls.view(-1).copy_(d[:, :, 0].sum((0, 1)).to(BF))
    dpos.view(N + 1, D).copy_(d.sum((0, 1)).to(BF))
    dtemporal.view(-1, D)[:T].copy_(d[:, :, 1:].sum((0, 2)).to(BF))


def vit_cls_fix_fwd(xt, tap, y, B, T, N1, D):
    m = _r16(tap.float().view(B, T, D).mean(1, keepdim=True))
    yv, xv = y.view(B, T, N1, D), xt.view(B, T, N1, D)
    yv[:, :, 0] = (xv[:, :, 0].float() + m).to(BF)
    return y


def vit_cls_merge_bwd_inplace(dy, B, T, N1, D):
    v = dy.view(B, T, N1, D)
    saved = v[:, :, 0].clone().view(B * T, D)
    v[:, :, 0] = v[:, :, 0].float().mean(1, keepdim=True).to(BF).expand(B, T, D)
    return saved


def copy_rows(src, dst, rows, cols, smap=IDENT, dmap=IDENT, lds=None, ldd=None):
    _wr(dst, _rows(dmap, rows), ldd or cols, _rd(src, _rows(smap, rows), lds or cols, cols))
    return dst


def colsum(x, rows, cols, *, out=None, ld=None, rmap=IDENT, accumulate=False):
    if out is None:
        out = torch.zeros(cols, dtype=BF, device=x.device)
    s = _rd(x, _rows(rmap, rows), ld or cols, cols).sum(0)
    if accumulate:
        s = s + out.float().view(-1)
    out.view(-1).copy_(s.to(BF))
    return out


def gpt_embed_fwd(query, ids, wte, wpe, B, Q, L, H, dropout_p=0.0, seed=0, offset=0):
    assert dropout_p == 0.0
    assert Q + L <= wpe.shape[0]
    h = torch.empty((B, Q + L, H), dtype=torch.float32)
    if Q:
        h[:, :Q] = query.float().view(B, Q, H)
    if L:
        h[:, Q:] = wte.float()[ids.view(B, L)]
    h = h + wpe.float()[:Q + L][None]
    return h.view(-1, H).to(BF)


def gpt_embed_bwd(dh, B, Q, L, H, dropout_p=0.0, seed=0, offset=0):
    assert dropout_p == 0.0
    return dh.view(B, Q + L, H)[:, :Q].reshape(B * Q, H).clone()


def gpt_embed_bwd_full(dh, rows, H, dropout_p=0.0, seed=0, offset=0):
    assert dropout_p == 0.0
    return dh.view(rows, H).clone()


def cross_entropy(logits, labels, weight, rows, vocab, *, ld=None, dlogits=None, want_losses=True):
    ld = ld or vocab
    lg = _rd(logits, torch.arange(rows), ld, vocab)
    lab = labels.view(-1)[:rows]
    ok = (lab >= 0) & (lab < vocab)
    lse = torch.logsumexp(lg, dim=1)
    tgt = lg.gather(1, lab.clamp(0, vocab - 1)[:, None]).squeeze(1)
    losses = torch.where(ok, lse - tgt, torch.zeros_like(lse))
    w = weight.float().view(-1)[:rows] if weight is not None else torch.ones(rows)
    if dlogits is not None:
        g = torch.softmax(lg, dim=1)
        g[torch.arange(rows)[ok], lab[ok]] -= 1.0
        g = g * (w * ok.float())[:, None]
        _wr(dlogits, torch.arange(rows), ld, g)
    return (losses if want_losses else None), (losses * w).sum()


def add(a, b, out=None):
    out = out if out is not None else torch.empty_like(a)
    out.copy_((a.float() + b.float()).to(BF))
    return out


def accum_f32(acc, g, first):
    if first:
        acc.copy_(g.float())
    else:
        acc.add_(g.float())
    return acc


def f32_to_bf16(src, dst):
    dst.copy_(src.to(BF))
    return dst


def copy_segments(pairs):
    for src, dst in pairs:
        assert src.numel() == dst.numel() and src.is_contiguous() and dst.is_contiguous()
        dst.view(-1).copy_(src.reshape(-1))


def vit_compose_bwd_finish(dwc_wpT, dbc, bp, wf, dwf, dbp, D):
    dwf.copy_((dwc_wpT.float().view(D, D) + dbc.float().view(D, 1) * bp.float().view(1, D)).to(BF))
    dbp.view(-1).copy_((wf.float().view(D, D) * dbc.float().view(D, 1)).sum(0).to(BF))


def gemm_batched(a_list, b_list, c_list, M, N, K, trans_a=False, trans_b=False):
    for a, b, c in zip(a_list, b_list, c_list):
        gemm(a, b, M, N, K, out=c, trans_a=trans_a, trans_b=trans_b)
    return c_list


def vit_compose_bias_batched(wf_list, bp_list, bf_list, bc_list, D):
    for wf, bp, bf, bc in zip(wf_list, bp_list, bf_list, bc_list):
        bc.view(-1).copy_((wf.float().view(D, D) @ bp.float().view(D) + bf.float().view(D)).to(BF))


def vit_compose_bwd_finish_batched(dwc_wpT, dbc, bp, wf, dwf, dbp, D):
    for args in zip(dwc_wpT, dbc, bp, wf, dwf, dbp):
        vit_compose_bwd_finish(*args, D)


def caption_targets(ids, attention_mask, prompt_len=None):
    B, L = ids.shape
    m = attention_mask[:, 1:].clone().float()
    if prompt_len is not None:
        m[torch.arange(L - 1)[None] < prompt_len.view(-1, 1)] = 0
    w = torch.zeros((B, L), dtype=torch.float32)
    w[:, :L - 1] = m / m.sum()
    labels = torch.cat([ids[:, 1:], ids[:, 1:2]], dim=1) if L > 1 else torch.zeros_like(ids)
    return labels.reshape(-1).contiguous(), w.view(-1)


def gather_rows(src, idx, rows, cols, ld=None):
    return _rd(src, idx.view(-1)[:rows], ld or cols, cols).to(BF)


def scatter_rows(src, idx, dst, rows, cols, ld=None):
    _wr(dst, idx.view(-1)[:rows], ld or cols, _rd(src, torch.arange(rows), cols, cols))
    return dst


def l2norm_fwd(x, rows, cols, eps=1e-12):
    xv = _rd(x, torch.arange(rows), cols, cols)
    n = xv.norm(dim=1).clamp_min(eps)
    return (xv / n[:, None]).to(BF), n


def l2norm_bwd(dy, x, nrm, rows, cols):
    y = _rd(x, torch.arange(rows), cols, cols) / nrm[:rows, None]
    dv = _rd(dy, torch.arange(rows), cols, cols)
    return ((dv - y * (y * dv).sum(1, keepdim=True)) / nrm[:rows, None]).to(BF)


def soft_target_ce(sim, row_ids, col_ids, scale, rows, cols, want_grad=True):
    s = sim.float().view(rows, cols)
    t = (row_ids.view(-1)[:rows, None] == col_ids.view(-1)[None, :cols]).float()
    t = t / t.sum(1, keepdim=True)
    losses = -(torch.log_softmax(s, dim=1) * t).sum(1)
    if not want_grad:
        return losses, None, None
    dsim = ((torch.softmax(s, dim=1) - t) * scale).to(BF)
    return losses, dsim, (dsim.float() * s).sum(1)


def gather_rows_ld(src, idx, dst, rows, cols, lds, ldd):
    _wr(dst, torch.arange(rows), ldd, _rd(src, idx.view(-1)[:rows], lds, cols))
    return dst


def logprob_topk(logits, k, add=None, rows=None, vocab=None, ld=None):
    rows = rows if rows is not None else logits.shape[0]
    vocab = vocab if vocab is not None else logits.shape[-1]
    lp = torch.log_softmax(_rd(logits, torch.arange(rows), ld or vocab, vocab), dim=1)
    if add is not None:
        lp = lp + add.float().view(-1)[:rows, None]
    val, idx = torch.sort(lp, dim=1, descending=True, stable=True)      # ties: the lower index first
    return val[:, :k].contiguous(), idx[:, :k].contiguous()


def decode_step(self, tokens, query_embeds=None):
    """include/mpv.h mpv_gpt_decode_step behind generation.DecodeState.step: one incremental decoder forward over the KV
    caches, composed from the stand-ins above (no dropout: generation runs in eval mode)."""
    gpt = self.gpt
    lm, cfg = gpt.dist_model.language_model, gpt.config
    B, H, np_, hn, V, ML = self.batch, self.H, self.np_, self.hn, self.V, self.max_len
    qf = None if query_embeds is None else query_embeds.reshape(-1, H).to(BF).contiguous()
    Q = 0 if qf is None else qf.shape[0] // B
    L = 0 if tokens is None else tokens.shape[1]
    n, pos0 = Q + L, self.pos
    assert pos0 + n <= ML and pos0 + n <= lm.embedding.position_embeddings.weight.shape[0]
    h = torch.empty((B, n, H), dtype=torch.float32)
    if Q:
        h[:, :Q] = qf.float().view(B, Q, H)
    if L:
        h[:, Q:] = lm.embedding.word_embeddings.weight.float()[tokens]
    h = (h + lm.embedding.position_embeddings.weight.float()[pos0:pos0 + n][None]).to(BF).view(B * n, H)
    cs = (ML * 3 * H, 3 * hn, 3 * H)
    lay = AttnLayout(cs, cs, cs, (n * H, hn, H))
    for li, layer in enumerate(lm.encoder.layers):
        att, mlp, cache = layer.self_attention, layer.mlp, self.cache[li]
        x1, _, _ = layernorm_fwd(h, layer.input_layernorm.weight, layer.input_layernorm.bias, layer.input_layernorm.eps, B * n, H)
        gemm(x1, att.query_key_value.weight, B * n, 3 * H, H, bias=att.query_key_value.bias, cmap=(n, ML, pos0), out=cache)
        ctx = torch.zeros((B * n, H), dtype=BF)
        attn_fwd(cache[pos0:], cache[:, hn:], cache[:, 2 * hn:], ctx, lay, B, np_, n, pos0 + n, hn, causal=True, scale=1.0 / math.sqrt(hn))
        h1 = gemm(ctx, att.dense.weight, B * n, H, H, bias=att.dense.bias, residual=h)
        x2, _, _ = layernorm_fwd(h1, layer.post_attention_layernorm.weight, layer.post_attention_layernorm.bias,
                                 layer.post_attention_layernorm.eps, B * n, H)
        F4 = mlp.dense_h_to_4h.out_features
        g = gemm(x2, mlp.dense_h_to_4h.weight, B * n, F4, H, bias=mlp.dense_h_to_4h.bias, act=ACT_GELU_TANH)
        h = gemm(g, mlp.dense_4h_to_h.weight, B * n, H, F4, bias=mlp.dense_4h_to_h.bias, residual=h1)
    fl = lm.encoder.final_layernorm
    xf, _, _ = layernorm_fwd(h, fl.weight, fl.bias, fl.eps, B, H, xmap=(1, n, n - 1))
    self.pos += n
    return gemm(xf, lm.embedding.word_embeddings.weight, B, V, H)


NAMES = ["ln_stream_fwd", "ln_stream_bwd", "LnDparamBatch", "accum_f32", "f32_to_bf16", "copy_segments", "vit_compose_bwd_finish", "gemm_batched", "vit_compose_bias_batched", "vit_compose_bwd_finish_batched", "caption_targets", "gather_rows_ld", "logprob_topk", "add", "gather_rows", "scatter_rows", "l2norm_fwd", "l2norm_bwd", "soft_target_ce", "gemm", "layernorm_fwd", "layernorm_bwd", "AttnLayout", "attn_fwd", "attn_bwd", "temporal_attn_fwd", "temporal_attn_bwd",
         "im2col_patches", "vit_embed_assemble_fwd", "vit_embed_assemble_bwd", "vit_cls_fix_fwd", "vit_cls_merge_bwd_inplace",
         "copy_rows", "colsum", "gpt_embed_fwd", "gpt_embed_bwd", "gpt_embed_bwd_full", "cross_entropy"]


def install(monkeypatch):
    """Replace the device wrappers of youku_mplug_amd.ops by the stand-ins above for the duration of a test."""
    import youku_mplug_amd  # noqa: F401
    from youku_mplug_amd import ops
    g = globals()
    for n in NAMES:
        monkeypatch.setattr(ops, n, g[n])
    from youku_mplug_amd import generation
    monkeypatch.setattr(generation.DecodeState, "step", decode_step)
