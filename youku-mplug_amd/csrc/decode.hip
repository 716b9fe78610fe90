// decode.hip -- host-side driver of one incremental decoder forward (prefill or a single-token decode step) over the
// per-layer KV caches: every launch of the step is issued from here, so generation costs one FFI call per step instead
// of ~170 (the step is launch-bound otherwise: ~5 rows of work per kernel).  Pure orchestration of the C-ABI kernels.
// Reference: models/modeling_distributed_gpt3.py:640-666 (embedding), :868-938 + :1034-1078 (layer with
// InferenceParams KV memory), :1184 (final LN), :1348-1350 (tied LM head).
#include <math.h>

#include "../../include/mpv.h"
#include "mpv_common.h"

static inline char* carve(char*& p, size_t bytes) {
  char* r = p;
  p += (bytes + 255) / 256 * 256;
  return r;
}

extern "C" size_t mpv_gpt_decode_workspace_size(const mpv_gpt_weights* w, int batch, int n_new) {
  const size_t R = (size_t)batch * n_new;
  return 4 * (R * w->hidden * 2 + 256) + (R * w->ffn * 2 + 256) + 2 * ((size_t)batch * w->hidden * 2 + 256) +
         mpv_gemm_workspace_size(R, w->ffn, w->hidden, 0, 0);
}

extern "C" int mpv_gpt_decode_step(const mpv_gpt_weights* w, void* const* kv_cache, int batch, int max_len, int pos0,
                                   const void* query, int Q, const int64_t* tokens, int L, void* workspace,
                                   size_t workspace_bytes, void* logits, hipStream_t stream) {
  MPV_REQUIRE(w && kv_cache && workspace && logits, MPV_E_ARG, "mpv_gpt_decode_step: null pointer");
  const int n = Q + L, H = w->hidden, np = w->heads, hn = H / np, F4 = w->ffn;
  MPV_REQUIRE(batch > 0 && n > 0 && pos0 >= 0 && pos0 + n <= max_len, MPV_E_SHAPE,
              "mpv_gpt_decode_step: positions [%d, %d) do not fit max_len %d", pos0, pos0 + n, max_len);
  MPV_REQUIRE(w->max_positions <= 0 || pos0 + n <= w->max_positions, MPV_E_SHAPE,
              "mpv_gpt_decode_step: positions [%d, %d) run past the %d rows of the position embedding", pos0, pos0 + n, w->max_positions);
  MPV_REQUIRE(workspace_bytes >= mpv_gpt_decode_workspace_size(w, batch, n), MPV_E_ARG, "mpv_gpt_decode_step: workspace too small");
  const int64_t R = (int64_t)batch * n;
  char* p = (char*)workspace;
  void* h = carve(p, R * H * 2);
  void* x = carve(p, R * H * 2);
  void* ctx = carve(p, R * H * 2);
  void* h1 = carve(p, R * H * 2);
  void* g = carve(p, R * F4 * 2);
  void* last = carve(p, (size_t)batch * H * 2);
  void* xf = carve(p, (size_t)batch * H * 2);
  void* gws = p;
  const size_t gws_bytes = workspace_bytes - (size_t)(p - (char*)workspace);
  int rc = mpv_gpt_embed_fwd(query, tokens, w->wte, (const char*)w->wpe + (size_t)pos0 * H * 2, h, batch, Q, L, H, 0.f, 0, 0, stream);
  if (rc) return rc;
  for (int li = 0; li < w->layers; ++li) {
    const mpv_gpt_layer_weights* lw = &w->layer[li];
    char* c = (char*)kv_cache[li];
    rc = mpv_layernorm_fwd(h, lw->ln1_w, lw->ln1_b, x, nullptr, nullptr, R, H, H, H, w->ln_eps, 0, 0, 0, 0, 0, 0, stream);
    if (rc) return rc;
    mpv_gemm_epilogue ep = {};
    ep.bias = lw->qkv_b;
    ep.c_group = n; ep.c_stride = max_len; ep.c_offset = pos0;                 // the new rows land in the cache
    rc = mpv_gemm_bf16(x, lw->qkv_w, c, R, 3 * H, H, H, H, 3 * H, 0, 0, &ep, gws, gws_bytes, stream);
    if (rc) return rc;
    mpv_attn_desc d = {};
    d.q = c + (size_t)pos0 * 3 * H * 2;
    d.k = c + (size_t)hn * 2;
    d.v = c + (size_t)2 * hn * 2;
    d.o = ctx;
    d.q_bs = d.k_bs = d.v_bs = (int64_t)max_len * 3 * H;
    d.q_hs = d.k_hs = d.v_hs = 3 * hn;
    d.q_rs = d.k_rs = d.v_rs = 3 * H;
    d.o_bs = (int64_t)n * H; d.o_hs = hn; d.o_rs = H;
    d.batch = batch; d.heads = np; d.sq = n; d.sk = pos0 + n; d.head_dim = hn;
    d.causal = 1;
    d.scale = 1.0f / sqrtf((float)hn);
    rc = mpv_attn_fwd(&d, stream);
    if (rc) return rc;
    mpv_gemm_epilogue e1 = {};
    e1.bias = lw->dense_b; e1.residual = h; e1.ldr = H;
    rc = mpv_gemm_bf16(ctx, lw->dense_w, h1, R, H, H, H, H, H, 0, 0, &e1, gws, gws_bytes, stream);
    if (rc) return rc;
    rc = mpv_layernorm_fwd(h1, lw->ln2_w, lw->ln2_b, x, nullptr, nullptr, R, H, H, H, w->ln_eps, 0, 0, 0, 0, 0, 0, stream);
    if (rc) return rc;
    mpv_gemm_epilogue e2 = {};
    e2.bias = lw->fc1_b; e2.act = MPV_ACT_GELU_TANH;
    rc = mpv_gemm_bf16(x, lw->fc1_w, g, R, F4, H, H, H, F4, 0, 0, &e2, gws, gws_bytes, stream);
    if (rc) return rc;
    mpv_gemm_epilogue e3 = {};
    e3.bias = lw->fc2_b; e3.residual = h1; e3.ldr = H;
    // the layer input `h` was last read by the dense GEMM (as its residual): its buffer takes the layer output
    rc = mpv_gemm_bf16(g, lw->fc2_w, h, R, H, F4, F4, F4, H, 0, 0, &e3, gws, gws_bytes, stream);
    if (rc) return rc;
  }
  const void* lastrow = h;
  if (n > 1) {   // LM head on the last position of every sequence only
    rc = mpv_copy_rows(h, last, batch, H, H, H, 1, n, n - 1, 0, 0, 0, stream);
    if (rc) return rc;
    lastrow = last;
  }
  rc = mpv_layernorm_fwd(lastrow, w->lnf_w, w->lnf_b, xf, nullptr, nullptr, batch, H, H, H, w->ln_eps, 0, 0, 0, 0, 0, 0, stream);
  if (rc) return rc;
  return mpv_gemm_bf16(xf, w->wte, logits, batch, w->vocab, H, H, H, w->vocab, 0, 0, nullptr, gws, gws_bytes, stream);
}

// InferenceParams.swap_key_value_dict (:1459-1473) for every layer: dst[l][j] = src[l][idx[j]], first `rows` positions
extern "C" int mpv_kv_reorder(void* const* src, void* const* dst, int layers, const int64_t* idx, int batch, int max_len, int rows,
                              int hidden, hipStream_t stream) {
  MPV_REQUIRE(src && dst && idx, MPV_E_ARG, "mpv_kv_reorder: null pointer");
  const int64_t pitch = (int64_t)max_len * 3 * hidden;
  for (int l = 0; l < layers; ++l) {
    const int rc = mpv_gather_rows_ld(src[l], idx, dst[l], batch, (int64_t)rows * 3 * hidden, pitch, pitch, stream);
    if (rc) return rc;
  }
  return MPV_OK;
}
