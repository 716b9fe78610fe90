"""Tensor-level wrappers over the C ABI (include/mpv.h).  PyTorch is used only for device
memory and streams; every function here launches hand-written gfx950 kernels on the current
stream.  No CPU / eager fallback exists: non-CUDA tensors raise.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import os

import torch

from . import _lib
from ._lib import AttnDesc, GemmEpilogue, check

ACT_NONE, ACT_GELU_ERF, ACT_GELU_TANH, ACT_RELU, ACT_DERIV = 0, 1, 2, 3, 4
# MPV_GELU_DERIV=0 (measurement knob): the MLPs park the pre-activation z and the dgrad epilogue evaluates GELU'(z), as until round 3.
# Default: the forward epilogue parks GELU'(z) itself and the dgrad epilogue is one multiply (include/mpv.h: preact_deriv, MPV_ACT_DERIV).
GELU_DERIV_FWD = os.environ.get("MPV_GELU_DERIV", "1") != "0"
RowMap = Tuple[int, int, int]
IDENT: RowMap = (0, 0, 0)

_ws = {}


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.MpvError("mpv ops run on the GPU only (no CPU fallback): got a CPU tensor")


def workspace(nbytes: int, device) -> torch.Tensor:
    """Stream-ordered scratch shared by all ops launched on one (device, stream) pair, grown on demand: launches on
    different streams may run concurrently and must not share split-K partials or arrival counters."""
    key = (torch.device(device).index or 0, torch.cuda.current_stream(device).cuda_stream)
    w = _ws.get(key)
    if w is None or w.numel() < nbytes:
        w = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _ws[key] = w
    return w


gemm_calls = 0      # mpv_gemm_bf16 calls of this process (profiles: per-call traffic = counter totals / calls; a call may be several row-band launches)


def gemm(a: torch.Tensor, b: torch.Tensor, M: int, N: int, K: int, *, out: Optional[torch.Tensor] = None,
         trans_a: bool = False, trans_b: bool = False, lda: Optional[int] = None, ldb: Optional[int] = None,
         ldc: Optional[int] = None, bias=None, act: int = 0, preact_out=None, residual=None, ldr: int = 0,
         act_bwd_z=None, act_bwd: int = 0, ldz: int = 0, dropout_p: float = 0.0, seed: int = 0, offset: int = 0,
         alpha_dev=None, alpha: float = 0.0, amap: RowMap = IDENT, cmap: RowMap = IDENT, kmap: RowMap = IDENT,
         out_rows: Optional[int] = None, accumulate: bool = False, out_f32: bool = False, colsum_out=None,
         tile_hint: int = 0, row_tap_out=None, row_tap_group: int = 0, split_hint: int = 0, gm_hint: int = 0,
         preact_deriv: bool = False, z_is_deriv: bool = False, keep_output: bool = False, colscale=None) -> torch.Tensor:
    """C[M,N] = epilogue(sum_k A(m,k) B(n,k)).  See include/mpv.h:mpv_gemm_bf16.
    colscale = (ncols, s): output columns below ncols leave as bf16(bf16(acc + bias) * s) (the q third of a packed qkv product).
    preact_deriv (forward of an MLP's first product): preact_out receives act'(z) instead of z; z_is_deriv (the matching dgrad):
    act_bwd_z is that tensor, multiply by it.  A call-site PAIR: both follow the one knob GELU_DERIV_FWD.
    keep_output: the output is small and read right behind this launch by a latency-bound kernel: plain instead of non-temporal stores."""
    global gemm_calls
    gemm_calls += 1
    _need_cuda(a, b)
    lda = lda if lda is not None else (M if trans_a else K)
    ldb = ldb if ldb is not None else (N if trans_b else K)
    ldc = ldc if ldc is not None else N
    if out is None:
        rows = out_rows if out_rows is not None else M
        out = torch.empty((rows, ldc), dtype=torch.float32 if out_f32 else torch.bfloat16, device=a.device)
    ep = GemmEpilogue()
    ep.a_group, ep.a_stride, ep.a_offset = amap
    ep.c_group, ep.c_stride, ep.c_offset = cmap
    ep.k_group, ep.k_stride, ep.k_offset = kmap
    ep.bias = _p(bias)
    ep.act = act
    ep.preact_out = _p(preact_out)
    ep.residual = _p(residual)
    ep.ldr = ldr
    ep.act_bwd_z = _p(act_bwd_z)
    ep.ldz = ldz
    ep.act_bwd = act_bwd
    ep.dropout_p = dropout_p
    ep.seed = seed
    ep.offset = offset
    ep.alpha_dev = _p(alpha_dev)
    ep.alpha = alpha
    ep.out_f32 = int(out_f32)
    ep.accumulate = int(accumulate)
    ep.colsum_out = _p(colsum_out)
    ep.tile_hint = tile_hint
    ep.row_tap_out = _p(row_tap_out)
    ep.row_tap_group = row_tap_group
    ep.split_hint = split_hint
    ep.gm_hint = gm_hint
    ep.preact_deriv = int(bool(preact_deriv and GELU_DERIV_FWD and preact_out is not None))
    ep.keep_output = int(bool(keep_output))
    if colscale is not None:
        ep.colscale_cols, ep.colscale = int(colscale[0]), float(colscale[1])
    if z_is_deriv and GELU_DERIV_FWD and act_bwd_z is not None:
        ep.act_bwd = ACT_DERIV
    ws, wsn = None, 0
    if not out_f32:
        wsn = _lib.lib().mpv_gemm_workspace_size(M, N, K, int(trans_a), int(trans_b))
        ws = workspace(wsn, a.device)
        wsn = ws.numel()
    check(_lib.lib().mpv_gemm_bf16(a.data_ptr(), b.data_ptr(), out.data_ptr(), M, N, K, lda, ldb, ldc, int(trans_a),
                                   int(trans_b), C.byref(ep), _p(ws), wsn, _stream()), "mpv_gemm_bf16")
    return out


def layernorm_fwd(x, gamma, beta, eps: float, rows: int, cols: int, *, out=None, xmap: RowMap = IDENT,
                  ymap: RowMap = IDENT, out_rows: Optional[int] = None, ldx: Optional[int] = None,
                  ldy: Optional[int] = None, want_stats: bool = True):
    _need_cuda(x, gamma, beta)
    ldx = ldx or cols
    ldy = ldy or cols
    if out is None:
        out = torch.empty((out_rows if out_rows is not None else rows, ldy), dtype=torch.bfloat16, device=x.device)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device) if want_stats else None
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device) if want_stats else None
    check(_lib.lib().mpv_layernorm_fwd(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), out.data_ptr(), _p(mean), _p(rstd),
                                       rows, cols, ldx, ldy, eps, *xmap, *ymap, _stream()), "mpv_layernorm_fwd")
    return out, mean, rstd


LN_DPARAM_DEFER = 2      # include/mpv.h: MPV_LN_DPARAM_DEFER


class LnDparamBatch:
    """dgamma / dbeta reductions of several LayerNorm backward calls batched into one two-launch finish (mpv_layernorm_dparam_finish).
    Pass it as `defer=` to layernorm_bwd: the call then leaves its per-workgroup partials in a buffer of this batch (kept
    and reused across steps) instead of launching its own two reduce kernels; finish() folds every pending entry on the
    current stream.  Entries of one batch must target distinct parameters."""

    def __init__(self):
        self._bufs = []
        self._pending = []
        self.cols = None

    def slot(self, nbytes: int, device) -> torch.Tensor:
        i = len(self._pending)
        while len(self._bufs) <= i:
            self._bufs.append(None)
        b = self._bufs[i]
        if b is None or b.numel() < nbytes or b.device != torch.device(device):
            b = torch.empty(nbytes, dtype=torch.uint8, device=device)
            self._bufs[i] = b
        return b

    def add(self, buf, nrows, dgamma, dbeta, accumulate, cols):
        assert self.cols in (None, cols), "one LnDparamBatch serves LayerNorms of one width"
        self.cols = cols
        self._pending.append((buf, nrows, dgamma, dbeta, int(accumulate)))

    def finish(self):
        n = len(self._pending)
        if n == 0:
            return
        vp, ip = C.c_void_p * n, C.c_int * n
        check(_lib.lib().mpv_layernorm_dparam_finish(vp(*[e[0].data_ptr() for e in self._pending]), ip(*[e[1] for e in self._pending]),
                                                     vp(*[e[2].data_ptr() for e in self._pending]), vp(*[e[3].data_ptr() for e in self._pending]),
                                                     ip(*[e[4] for e in self._pending]), n, self.cols, _stream()),
              "mpv_layernorm_dparam_finish")
        self._pending = []


def layernorm_bwd(dy, x, gamma, mean, rstd, rows: int, cols: int, *, dres=None, dx=None, dx_drop=None,
                  dropout_p: float = 0.0, seed: int = 0, offset: int = 0, dgamma=None, dbeta=None,
                  accumulate_dparams: bool = False, xmap: RowMap = IDENT, ymap: RowMap = IDENT,
                  ldx: Optional[int] = None, ldy: Optional[int] = None, dx_rows: Optional[int] = None,
                  defer: Optional[LnDparamBatch] = None):
    _need_cuda(dy, x, gamma)
    ldx = ldx or cols
    ldy = ldy or cols
    if dx is None:
        dx = torch.empty((dx_rows if dx_rows is not None else rows, ldx), dtype=torch.bfloat16, device=x.device)
    ws, wsn = None, 0
    mode = int(accumulate_dparams)
    if dgamma is not None:
        wsn = _lib.lib().mpv_layernorm_bwd_workspace_size(cols)
        if defer is not None:
            ws = defer.slot(wsn, x.device)
            defer.add(ws, _lib.lib().mpv_layernorm_bwd_partial_rows(rows), dgamma, dbeta, accumulate_dparams, cols)
            mode = LN_DPARAM_DEFER
        else:
            ws = workspace(wsn, x.device)
        wsn = ws.numel()
    check(_lib.lib().mpv_layernorm_bwd(dy.data_ptr(), x.data_ptr(), gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                       _p(dres), dx.data_ptr(), _p(dx_drop), dropout_p, seed, offset, _p(dgamma), _p(dbeta),
                                       mode, rows, cols, ldx, ldy, *xmap, *ymap, _p(ws), wsn, _stream()),
          "mpv_layernorm_bwd")
    return dx


def ln_stream_fwd(h_in, add, gamma, beta, eps: float, rows: int, cols: int, *, h_out=None, out=None, hmap: RowMap = IDENT,
                  amap: RowMap = IDENT, ymap: RowMap = IDENT, out_rows: Optional[int] = None, h_rows: Optional[int] = None,
                  add_dropout_p: float = 0.0, seed: int = 0, offset: int = 0):
    """The decoder's residual stream in fp32 (include/mpv.h: mpv_ln_stream_fwd): h' = h_in + add in fp32, y = LN(h') in bf16.
    h_in is the fp32 stream, or the bf16 embedding output for the first LayerNorm; add None: plain LN of h_in.
    add_dropout_p > 0: `add` is dropped here (mpv_ln_stream_fwd_drop: the index / threshold / rounding of the GEMM's dropout epilogue).
    -> (y, h' (fp32, None without add), mean, rstd)."""
    _need_cuda(h_in, gamma, beta)
    if out is None:
        out = torch.empty((out_rows if out_rows is not None else rows, cols), dtype=torch.bfloat16, device=h_in.device)
    if add is not None and h_out is None:
        h_out = torch.empty((h_rows if h_rows is not None else h_in.shape[0], cols), dtype=torch.float32, device=h_in.device)
    mean = torch.empty(rows, dtype=torch.float32, device=h_in.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=h_in.device)
    check(_lib.lib().mpv_ln_stream_fwd_drop(h_in.data_ptr(), int(h_in.dtype == torch.bfloat16), _p(add), _p(h_out) if add is not None else None,
                                       gamma.data_ptr(), beta.data_ptr(), out.data_ptr(), mean.data_ptr(), rstd.data_ptr(), rows, cols,
                                       cols, cols, cols, eps, *hmap, *amap, *ymap, float(add_dropout_p) if add is not None else 0.0, seed, offset, _stream()), "mpv_ln_stream_fwd_drop")
    return out, (h_out if add is not None else None), mean, rstd


def ln_stream_bwd(dy, x, gamma, mean, rstd, rows: int, cols: int, *, dres=None, dx=None, dx_drop=None, dropout_p: float = 0.0,
                  seed: int = 0, offset: int = 0, xmap: RowMap = IDENT, ymap: RowMap = IDENT, dx_rows: Optional[int] = None):
    """LayerNorm backward of the frozen decoder with x read from the fp32 stream (no parameter gradients)."""
    _need_cuda(dy, x, gamma)
    assert x.dtype == torch.float32
    if dx is None:
        dx = torch.empty((dx_rows if dx_rows is not None else rows, cols), dtype=torch.bfloat16, device=x.device)
    check(_lib.lib().mpv_ln_stream_bwd(dy.data_ptr(), x.data_ptr(), gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(), _p(dres), dx.data_ptr(),
                                       _p(dx_drop), dropout_p, seed, offset, rows, cols, cols, cols, *xmap, *ymap, _stream()),
          "mpv_ln_stream_bwd")
    return dx


class AttnLayout:
    """Strides (in elements) of q/k/v/o for mpv_attn_*: (batch, head, row)."""

    def __init__(self, q, k, v, o):
        self.q, self.k, self.v, self.o = q, k, v, o


def _attn_desc(q, k, v, o, lse, lay: AttnLayout, batch, heads, sq, sk, hd, causal, scale, scale_q_bf16, dropout_p, seed,
               offset):
    d = AttnDesc()
    d.q, d.k, d.v, d.o, d.lse = q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), _p(lse)
    d.q_bs, d.q_hs, d.q_rs = lay.q
    d.k_bs, d.k_hs, d.k_rs = lay.k
    d.v_bs, d.v_hs, d.v_rs = lay.v
    d.o_bs, d.o_hs, d.o_rs = lay.o
    d.batch, d.heads, d.sq, d.sk, d.head_dim = batch, heads, sq, sk, hd
    d.causal, d.scale, d.scale_q_bf16 = int(causal), scale, int(scale_q_bf16)
    d.dropout_p, d.seed, d.offset = dropout_p, seed, offset
    return d


def attn_fwd(q, k, v, o, lay: AttnLayout, batch, heads, sq, sk, hd, *, causal=False, scale=1.0, scale_q_bf16=False,
             dropout_p=0.0, seed=0, offset=0):
    """q/k/v/o are tensors whose data_ptr() is element (0,0,0,0) under `lay`; returns lse [batch, heads, sq]."""
    _need_cuda(q, k, v, o)
    lse = torch.empty((batch, heads, sq), dtype=torch.float32, device=q.device)
    d = _attn_desc(q, k, v, o, lse, lay, batch, heads, sq, sk, hd, causal, scale, scale_q_bf16, dropout_p, seed, offset)
    check(_lib.lib().mpv_attn_fwd(C.byref(d), _stream()), "mpv_attn_fwd")
    return lse


def attn_bwd(q, k, v, o, lse, do, dq, dk, dv, lay: AttnLayout, batch, heads, sq, sk, hd, *, causal=False, scale=1.0,
             scale_q_bf16=False, dropout_p=0.0, seed=0, offset=0):
    _need_cuda(q, k, v, o, do, dq, dk, dv)
    delta = torch.empty((batch, heads, sq), dtype=torch.float32, device=q.device)
    d = _attn_desc(q, k, v, o, lse, lay, batch, heads, sq, sk, hd, causal, scale, scale_q_bf16, dropout_p, seed, offset)
    check(_lib.lib().mpv_attn_bwd(C.byref(d), do.data_ptr(), dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), delta.data_ptr(),
                                  _stream()), "mpv_attn_bwd")


def temporal_attn_fwd(qkv, out, n_outer, outer_stride, n_inner, inner_offset, t_stride, T, heads, hd, scale):
    _need_cuda(qkv, out)
    check(_lib.lib().mpv_temporal_attn_fwd(qkv.data_ptr(), out.data_ptr(), n_outer, outer_stride, n_inner, inner_offset,
                                           t_stride, T, heads, hd, scale, _stream()), "mpv_temporal_attn_fwd")


def temporal_attn_bwd(qkv, dout, dqkv, n_outer, outer_stride, n_inner, inner_offset, t_stride, T, heads, hd, scale):
    _need_cuda(qkv, dout, dqkv)
    check(_lib.lib().mpv_temporal_attn_bwd(qkv.data_ptr(), dout.data_ptr(), dqkv.data_ptr(), n_outer, outer_stride, n_inner,
                                           inner_offset, t_stride, T, heads, hd, scale, _stream()), "mpv_temporal_attn_bwd")


def im2col_patches(video, B, Cc, T, H, W, P, kpad):
    _need_cuda(video)
    rows = B * T * (H // P) * (W // P)
    cols = torch.empty((rows, kpad), dtype=torch.bfloat16, device=video.device)
    check(_lib.lib().mpv_im2col_patches(video.data_ptr(), cols.data_ptr(), B, Cc, T, H, W, P, kpad, _stream()),
          "mpv_im2col_patches")
    return cols


def vit_embed_assemble_fwd(patch, cls_token, pos_embed, temporal_embed, B, T, N, D):
    x = torch.empty((B * T * (N + 1), D), dtype=torch.bfloat16, device=patch.device)
    check(_lib.lib().mpv_vit_embed_assemble_fwd(patch.data_ptr(), cls_token.data_ptr(), pos_embed.data_ptr(),
                                                temporal_embed.data_ptr(), x.data_ptr(), B, T, N, D, _stream()),
          "mpv_vit_embed_assemble_fwd")
    return x


def vit_embed_assemble_bwd(dx, dpatch, dcls, dpos, dtemporal, B, T, N, D):
    check(_lib.lib().mpv_vit_embed_assemble_bwd(dx.data_ptr(), dpatch.data_ptr(), dcls.data_ptr(), dpos.data_ptr(),
                                                dtemporal.data_ptr(), B, T, N, D, _stream()), "mpv_vit_embed_assemble_bwd")


def vit_cls_merge_fwd(xt, a, B, T, N1, D, out=None):
    y = out if out is not None else torch.empty_like(xt)
    check(_lib.lib().mpv_vit_cls_merge_fwd(xt.data_ptr(), a.data_ptr(), y.data_ptr(), B, T, N1, D, _stream()),
          "mpv_vit_cls_merge_fwd")
    return y


def vit_cls_fix_fwd(xt, tap, y, B, T, N1, D):
    """y (= xt + a on all rows) gets its cls slots rewritten: xt_cls + bf16(mean_t tap)."""
    check(_lib.lib().mpv_vit_cls_fix_fwd(xt.data_ptr(), tap.data_ptr(), y.data_ptr(), B, T, N1, D, _stream()), "mpv_vit_cls_fix_fwd")
    return y


def vit_cls_merge_bwd_inplace(dy, B, T, N1, D):
    """dy's cls rows <- their mean over t, in place; returns the saved originals [B*T, D] (restore with copy_rows)."""
    saved = torch.empty((B * T, D), dtype=torch.bfloat16, device=dy.device)
    check(_lib.lib().mpv_vit_cls_merge_bwd_inplace(dy.data_ptr(), saved.data_ptr(), B, T, N1, D, _stream()),
          "mpv_vit_cls_merge_bwd_inplace")
    return saved


def vit_cls_merge_bwd(dy, B, T, N1, D, out=None):
    da = out if out is not None else torch.empty_like(dy)
    check(_lib.lib().mpv_vit_cls_merge_bwd(dy.data_ptr(), da.data_ptr(), B, T, N1, D, _stream()), "mpv_vit_cls_merge_bwd")
    return da


def copy_rows(src, dst, rows, cols, smap: RowMap = IDENT, dmap: RowMap = IDENT, lds=None, ldd=None):
    check(_lib.lib().mpv_copy_rows(src.data_ptr(), dst.data_ptr(), rows, cols, lds or cols, ldd or cols, *smap, *dmap,
                                   _stream()), "mpv_copy_rows")
    return dst


def colsum(x, rows, cols, *, out=None, ld=None, rmap: RowMap = IDENT, accumulate=False):
    if out is None:
        out = torch.empty(cols, dtype=torch.bfloat16, device=x.device)
    wsn = _lib.lib().mpv_colsum_workspace_size(cols)
    ws = workspace(wsn, x.device)
    check(_lib.lib().mpv_colsum(x.data_ptr(), out.data_ptr(), rows, cols, ld or cols, *rmap, int(accumulate), ws.data_ptr(),
                                ws.numel(), _stream()), "mpv_colsum")
    return out


def add(a, b, out=None):
    out = out if out is not None else torch.empty_like(a)
    check(_lib.lib().mpv_add(a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), _stream()), "mpv_add")
    return out


def accum_f32(acc, g, first: bool):
    """acc (fp32) = g if first else acc + g, g bf16: the gradient-accumulation window sum."""
    check(_lib.lib().mpv_accum_f32(acc.data_ptr(), g.data_ptr(), g.numel(), int(first), _stream()), "mpv_accum_f32")
    return acc


def f32_to_bf16(src, dst):
    check(_lib.lib().mpv_f32_to_bf16(src.data_ptr(), dst.data_ptr(), src.numel(), _stream()), "mpv_f32_to_bf16")
    return dst


def copy_segments(pairs):
    """[(src, dst), ...] of equally sized contiguous bf16 tensors: dst[i] <- src[i], one launch."""
    n = len(pairs)
    if n == 0:
        return
    vp, lp = C.c_void_p * n, C.c_int64 * n
    check(_lib.lib().mpv_copy_segments(vp(*[s.data_ptr() for s, _ in pairs]), vp(*[d.data_ptr() for _, d in pairs]),
                                       lp(*[d.numel() for _, d in pairs]), n, _stream()), "mpv_copy_segments")


def vit_compose_bwd_finish(dwc_wpT, dbc, bp, wf, dwf, dbp, D):
    """dwf <- bf16(float(dwc_wpT) + dbc (x) bp);  dbp <- Wf^T dbc  (include/mpv.h: mpv_vit_compose_bwd_finish)."""
    check(_lib.lib().mpv_vit_compose_bwd_finish(dwc_wpT.data_ptr(), dbc.data_ptr(), bp.data_ptr(), wf.data_ptr(), dwf.data_ptr(),
                                                dbp.data_ptr(), D, _stream()), "mpv_vit_compose_bwd_finish")


def _ptr_table(tensors):
    return (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


def gemm_batched(a_list, b_list, c_list, M, N, K, trans_a=False, trans_b=False):
    """c_list[i] <- a_list[i] op b_list[i] for every i in ONE launch (include/mpv.h: mpv_gemm_bf16_batched): mpv_gemm_bf16's operand
    forms, one shape, contiguous bf16 operands, plain epilogue."""
    global gemm_calls
    n = len(a_list)
    assert n == len(b_list) == len(c_list) and n > 0
    _need_cuda(*a_list, *b_list, *c_list)
    lda = M if trans_a else K
    ldb = N if trans_b else K
    gemm_calls += 1
    check(_lib.lib().mpv_gemm_bf16_batched(_ptr_table(a_list), _ptr_table(b_list), _ptr_table(c_list), n, M, N, K, lda, ldb, N,
                                           int(trans_a), int(trans_b), _stream()), "mpv_gemm_bf16_batched")
    return c_list


def vit_compose_bias_batched(wf_list, bp_list, bf_list, bc_list, D):
    """bc[i] <- bf16(Wf[i] bp[i] + bf[i]) for every block in one launch (include/mpv.h: mpv_vit_compose_bias_batched)."""
    _need_cuda(*wf_list, *bp_list, *bf_list, *bc_list)
    check(_lib.lib().mpv_vit_compose_bias_batched(_ptr_table(wf_list), _ptr_table(bp_list), _ptr_table(bf_list), _ptr_table(bc_list),
                                                  len(wf_list), D, _stream()), "mpv_vit_compose_bias_batched")


def vit_compose_bwd_finish_batched(dwc_wpT, dbc, bp, wf, dwf, dbp, D):
    """mpv_vit_compose_bwd_finish for every block in one launch (lists of tensors)."""
    _need_cuda(*dwc_wpT, *dbc, *bp, *wf, *dwf, *dbp)
    check(_lib.lib().mpv_vit_compose_bwd_finish_batched(_ptr_table(dwc_wpT), _ptr_table(dbc), _ptr_table(bp), _ptr_table(wf), _ptr_table(dwf),
                                                        _ptr_table(dbp), len(wf), D, _stream()), "mpv_vit_compose_bwd_finish_batched")


def caption_targets(ids, attention_mask, prompt_len=None):
    """-> (labels int64 [B*L], weights fp32 [B*L]) of the L text positions (include/mpv.h: mpv_caption_targets)."""
    _need_cuda(ids, attention_mask)
    B, L = ids.shape
    labels = torch.empty(B * L, dtype=torch.long, device=ids.device)
    weights = torch.empty(B * L, dtype=torch.float32, device=ids.device)
    check(_lib.lib().mpv_caption_targets(ids.contiguous().data_ptr(), attention_mask.contiguous().data_ptr(), _p(prompt_len), B, L,
                                         labels.data_ptr(), weights.data_ptr(), _stream()), "mpv_caption_targets")
    return labels, weights


def gpt_embed_fwd(query, ids, wte, wpe, B, Q, L, H, dropout_p=0.0, seed=0, offset=0):
    if Q + L > wpe.shape[0]:      # nn.Embedding would raise on the position ids (models/modeling_distributed_gpt3.py:640-666)
        raise _lib.MpvError(f"sequence of {Q + L} positions exceeds max_position_embeddings = {wpe.shape[0]}")
    h = torch.empty((B * (Q + L), H), dtype=torch.bfloat16, device=wte.device)
    check(_lib.lib().mpv_gpt_embed_fwd(_p(query), ids.data_ptr(), wte.data_ptr(), wpe.data_ptr(), h.data_ptr(), B, Q, L, H,
                                       dropout_p, seed, offset, _stream()), "mpv_gpt_embed_fwd")
    return h


def gpt_embed_bwd_full(dh, rows, H, dropout_p=0.0, seed=0, offset=0):
    """dropout-masked gradient of every row of the embedding output (trainable decoder): include/mpv.h"""
    out = torch.empty((rows, H), dtype=torch.bfloat16, device=dh.device)
    check(_lib.lib().mpv_gpt_embed_bwd_full(dh.data_ptr(), out.data_ptr(), rows, H, dropout_p, seed, offset, _stream()), "mpv_gpt_embed_bwd_full")
    return out


def gpt_embed_bwd(dh, B, Q, L, H, dropout_p=0.0, seed=0, offset=0):
    dq = torch.empty((B * Q, H), dtype=torch.bfloat16, device=dh.device)
    check(_lib.lib().mpv_gpt_embed_bwd(dh.data_ptr(), dq.data_ptr(), B, Q, L, H, dropout_p, seed, offset, _stream()),
          "mpv_gpt_embed_bwd")
    return dq


def cross_entropy(logits, labels, weight, rows, vocab, *, ld=None, dlogits=None, want_losses=True):
    """Returns (losses[rows] fp32, loss_sum fp32 scalar tensor)."""
    losses = torch.empty(rows, dtype=torch.float32, device=logits.device) if want_losses else None
    loss_sum = torch.zeros((), dtype=torch.float32, device=logits.device)
    check(_lib.lib().mpv_cross_entropy(logits.data_ptr(), labels.data_ptr(), _p(weight), _p(losses), loss_sum.data_ptr(),
                                       _p(dlogits), rows, vocab, ld or vocab, _stream()), "mpv_cross_entropy")
    return losses, loss_sum


def grad_sumsq(grad_flat, sumsq):
    """sumsq += sum g^2, bit-reproducible (include/mpv.h: per-workgroup partials in the workspace, added in a fixed order)"""
    ws = workspace(_lib.lib().mpv_grad_sumsq_workspace_size(), grad_flat.device)
    check(_lib.lib().mpv_grad_sumsq(grad_flat.data_ptr(), grad_flat.numel(), sumsq.data_ptr(), ws.data_ptr(), ws.numel(), _stream()), "mpv_grad_sumsq")


def adamw_step(p16, master, m, v, g16, lr, beta1, beta2, eps, wd, step, grad_scale=1.0, sumsq=None, max_norm=0.0):
    check(_lib.lib().mpv_adamw_step(p16.data_ptr(), master.data_ptr(), m.data_ptr(), v.data_ptr(), g16.data_ptr(), p16.numel(),
                                    lr, beta1, beta2, eps, wd, step, grad_scale, _p(sumsq), max_norm, _stream()),
          "mpv_adamw_step")


def adamw_step_grouped(p16, master, m, v, g16, tile_group, lrs, wds, beta1, beta2, eps, step, grad_scale=1.0, sumsq=None,
                       max_norm=0.0):
    n = len(lrs)
    arr = (C.c_float * n)
    check(_lib.lib().mpv_adamw_step_grouped(p16.data_ptr(), master.data_ptr(), m.data_ptr(), v.data_ptr(), g16.data_ptr(),
                                            p16.numel(), tile_group.data_ptr(), arr(*lrs), arr(*wds), n, beta1, beta2, eps, step,
                                            grad_scale, _p(sumsq), max_norm, _stream()), "mpv_adamw_step_grouped")


def store_words(dst: torch.Tensor, words):
    """dst (device, 4-byte aligned) <- up to 32 32-bit words, by value in the launch (include/mpv.h: mpv_store_words)"""
    n = len(words)
    check(_lib.lib().mpv_store_words(dst.data_ptr(), (C.c_uint32 * n)(*words), n, _stream()), "mpv_store_words")


def store_u64(dst: torch.Tensor, values):
    words = []
    for v in values:
        v &= 0xFFFFFFFFFFFFFFFF
        words += [v & 0xFFFFFFFF, v >> 32]
    store_words(dst, words)


def adamw_hyper_upload(lrs, wds, beta1, beta2, step, hyper_dev: torch.Tensor):
    """hyper_dev (device float32[18]) <- lr[8], wd[8], 1/(1-beta1^step), 1/sqrt(1-beta2^step), exactly as the by-value entry computes them"""
    n = len(lrs)
    arr = (C.c_float * n)
    out = (C.c_float * 18)()
    check(_lib.lib().mpv_adamw_hyper_pack(arr(*lrs), arr(*wds), n, beta1, beta2, step, out), "mpv_adamw_hyper_pack")
    store_words(hyper_dev, list((C.c_uint32 * 18).from_buffer(out)))


def adamw_step_grouped_dev(p16, master, m, v, g16, tile_group, hyper_dev, beta1, beta2, eps, grad_scale=1.0, sumsq=None, max_norm=0.0):
    """adamw_step_grouped with lr / weight decay / bias corrections read from device memory (hyper_dev float32[18]): include/mpv.h"""
    check(_lib.lib().mpv_adamw_step_grouped_dev(p16.data_ptr(), master.data_ptr(), m.data_ptr(), v.data_ptr(), g16.data_ptr(),
                                                p16.numel(), tile_group.data_ptr(), hyper_dev.data_ptr(), beta1, beta2, eps, grad_scale,
                                                _p(sumsq), max_norm, _stream()), "mpv_adamw_step_grouped_dev")


def l2norm_fwd(x, rows, cols, eps=1e-12):
    y = torch.empty((rows, cols), dtype=torch.bfloat16, device=x.device)
    nrm = torch.empty(rows, dtype=torch.float32, device=x.device)
    check(_lib.lib().mpv_l2norm_fwd(x.data_ptr(), y.data_ptr(), nrm.data_ptr(), rows, cols, eps, _stream()), "mpv_l2norm_fwd")
    return y, nrm


def l2norm_bwd(dy, x, nrm, rows, cols):
    dx = torch.empty((rows, cols), dtype=torch.bfloat16, device=x.device)
    check(_lib.lib().mpv_l2norm_bwd(dy.data_ptr(), x.data_ptr(), nrm.data_ptr(), dx.data_ptr(), rows, cols, _stream()), "mpv_l2norm_bwd")
    return dx


def gather_rows(src, idx, rows, cols, ld=None):
    dst = torch.empty((rows, cols), dtype=torch.bfloat16, device=src.device)
    check(_lib.lib().mpv_gather_rows(src.data_ptr(), idx.data_ptr(), dst.data_ptr(), rows, cols, ld or cols, _stream()), "mpv_gather_rows")
    return dst


def scatter_rows(src, idx, dst, rows, cols, ld=None):
    check(_lib.lib().mpv_scatter_rows(src.data_ptr(), idx.data_ptr(), dst.data_ptr(), rows, cols, ld or cols, _stream()), "mpv_scatter_rows")
    return dst


def gather_rows_ld(src, idx, dst, rows, cols, lds, ldd):
    check(_lib.lib().mpv_gather_rows_ld(src.data_ptr(), idx.data_ptr(), dst.data_ptr(), rows, cols, lds, ldd, _stream()), "mpv_gather_rows_ld")
    return dst


def logprob_topk(logits, k, add=None, rows=None, vocab=None, ld=None):
    """-> (values fp32 [rows,k], indices int64 [rows,k]) of log_softmax(logits) + add[:,None], descending."""
    rows = rows if rows is not None else logits.shape[0]
    vocab = vocab if vocab is not None else logits.shape[-1]
    val = torch.empty((rows, k), dtype=torch.float32, device=logits.device)
    idx = torch.empty((rows, k), dtype=torch.int64, device=logits.device)
    ws = workspace(_lib.lib().mpv_logprob_topk_workspace_size(rows, k), logits.device)
    check(_lib.lib().mpv_logprob_topk(logits.data_ptr(), _p(add), rows, vocab, ld or vocab, k, val.data_ptr(), idx.data_ptr(),
                                      ws.data_ptr(), ws.numel(), _stream()), "mpv_logprob_topk")
    return val, idx


def soft_target_ce(sim, row_ids, col_ids, scale, rows, cols, want_grad=True):
    losses = torch.empty(rows, dtype=torch.float32, device=sim.device)
    dsim = torch.empty((rows, cols), dtype=torch.bfloat16, device=sim.device) if want_grad else None
    dts = torch.empty(rows, dtype=torch.float32, device=sim.device) if want_grad else None
    check(_lib.lib().mpv_soft_target_ce(sim.data_ptr(), row_ids.data_ptr(), col_ids.data_ptr(), scale, losses.data_ptr(), _p(dsim),
                                        _p(dts), rows, cols, _stream()), "mpv_soft_target_ce")
    return losses, dsim, dts
