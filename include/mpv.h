/* mpv.h -- C ABI of libmpv_hip.so: hand-written gfx950 (MI355X / CDNA4) kernels for the
 * mPLUG-Video pre-train hot path.
 *
 * The reference (X-PLUG/Youku-mPLUG) has no FFI/plugin layer: its hot path reaches native code
 * only through PyTorch/cuBLAS/cuDNN, megatron_util's fused CUDA kernels and DeepSpeed's
 * FusedAdam.  Each entry point below names the reference call site(s) whose native work it
 * replaces (paths relative to the reference repo root).  The Python host side
 * (youku-mplug_amd/) binds these with ctypes; see INTEGRATION.md for the binding a reference
 * maintainer would add.
 *
 * Conventions
 *  - All tensor pointers are device pointers owned by the caller (PyTorch); bf16 unless noted.
 *    The library never allocates user-visible memory: scratch is passed in (workspace + size,
 *    with a *_workspace_size query).
 *  - Every call takes the hipStream_t to launch on and is asynchronous.
 *  - Return 0 on success, negative MPV_E_* otherwise; mpv_last_error() returns a thread-local
 *    message.  Nothing aborts.
 *  - Re-entrant; no global mutable state besides the thread-local error string.
 *  - Row maps: logical row r lives at physical row (r / group) * stride + (r % group) + offset;
 *    group == 0 means identity.  They address the token rows of the [B,T,1+N,D] ViT stream
 *    (one cls slot per frame) and the abstractor's K/V buffer with its extra bias-kv token.
 */
#ifndef MPV_H_
#define MPV_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* mpv_stream_t; /* == hipStream_t */

#define MPV_OK 0
#define MPV_E_SHAPE (-1)
#define MPV_E_ALIGN (-2)
#define MPV_E_ARCH (-3)
#define MPV_E_HIP (-4)
#define MPV_E_ARG (-5)

#define MPV_ACT_NONE 0
#define MPV_ACT_GELU_ERF 1  /* nn.GELU, models/vision_transformer.py:94,99 */
#define MPV_ACT_GELU_TANH 2 /* megatron bias_gelu_impl, models/modeling_distributed_gpt3.py:586-588 */
#define MPV_ACT_RELU 3      /* nn.ReLU of cls_head, models/distributed_gpt3.py:526-530, 1081-1085 */
#define MPV_ACT_DERIV 4     /* act_bwd only: act_bwd_z already HOLDS act'(z) (a forward launch with preact_deriv wrote it): the
                               backward epilogue is one multiply.  The GELU' polynomial then runs in the forward epilogue, which
                               is bound by its two stores and has the VALU slots free, instead of in the dgrad epilogue, which it
                               bound (fc2-dgrad x GELU' at K = 768: 33 vs 26 us per round of tiles, r04) */

/* Dropout seeds.  Every `seed` argument / field of this header is a 64-bit value.  A seed with bit 63 CLEAR is the seed itself.
 * A seed with bit 63 SET (MPV_SEED_FROM_DEVICE(ptr)) carries in its low 63 bits the device address of a uint64_t that holds the
 * seed; the kernel reads it when it RUNS.  Kernel arguments are frozen when a training step is captured into a HIP graph: the
 * indirect form is what lets each replay draw fresh masks (the host rewrites the 8 bytes before launching the graph).  Both forms
 * of the same value produce the same masks. */
#define MPV_SEED_FROM_DEVICE(ptr) ((uint64_t)(uintptr_t)(ptr) | (1ull << 63))

int mpv_version(void);
const char* mpv_last_error(void);
/* 0 if the current device is gfx950, MPV_E_ARCH otherwise */
int mpv_check_device(void);
/* the same test on a gcnArchName string (hipDeviceProp_t::gcnArchName, e.g. "gfx950:sramecc+:xnack-"): 0 / MPV_E_ARCH / MPV_E_ARG */
int mpv_check_arch_name(const char* gcn_arch_name);

/* ------------------------------------------------------------------------------------------
 * GEMM: C[M,N] = epilogue(sum_k A(m,k) B(n,k)), bf16 in, fp32 accumulate (v_mfma_f32_16x16x32_bf16 in the 256x256 kernel, 32x32x16 in the 128x128 fallback).
 * transA=0: A is [M][K] (lda); transA=1: A is [K][M].  transB=0: B is [N][K] (nn.Linear
 * weight); transB=1: B is [K][N].  (0,0)=forward, (0,1)=dgrad, (1,1)=wgrad (split-K inside).
 * Replaces: F.linear / nn.Linear (models/vision_transformer.py:104,108,175,205,250),
 * nn.MultiheadAttention projections (:353), visual_fc (models/distributed_gpt3.py:111,136),
 * mpu.Column/RowParallelLinear + LM head (models/modeling_distributed_gpt3.py:562,573,843,852,
 * 1348), bias_gelu (:586-588), bias_dropout_add (:953-979), and their autograd backward. */
typedef struct mpv_gemm_epilogue {
  int a_group, a_stride, a_offset; /* row map on A rows (transA=0 only)                     */
  int c_group, c_stride, c_offset; /* row map on C / residual / preact rows                  */
  int k_group, k_stride, k_offset; /* row map on the reduction rows (transposed operands)    */
  const void* bias;                /* bf16 [N] or NULL (16-byte aligned for the 256x256 kernel, which fetches a tile's slice by
                                      LDS-DMA; any other alignment is taken by the 128x128 kernel) */
  int act;                         /* MPV_ACT_*: y = act(bf16(acc + bias))                    */
  void* preact_out;                /* optional bf16 copy of (acc + bias) before act (ldc)     */
  const void* residual;            /* optional bf16 [.,N] added last (leading dim ldr)        */
  int64_t ldr;
  const void* act_bwd_z; /* optional bf16 z [M][ldz]: multiply by act'(z) (GELU backward)    */
  int64_t ldz;
  int act_bwd;     /* MPV_ACT_* selecting act' for act_bwd_z                                  */
  float dropout_p; /* dropout on (acc + bias) before the residual add                         */
  uint64_t seed, offset;
  const float* alpha_dev; /* optional device scalar multiplier                                */
  float alpha;            /* host scalar multiplier (0 -> 1)                                  */
  int out_f32;            /* C is fp32 (no epilogue besides alpha / accumulate)               */
  int accumulate;         /* C += result                                                      */
  void* colsum_out;       /* wgrad only: bf16 [M] = column sums of A over the (mapped) reduction rows,
                             i.e. the bias gradient that goes with dW = dY^T X, fused into the same pass */
  int tile_hint;          /* 0: the library picks the tile kernel per problem; 128 / 256 pin the 128x128 or the
                             256x256 eight-phase kernel where it applies, 192 / 160 its 192- and 160-row tile variants
                             (tests and measurements)                                                          */
  void* row_tap_out;      /* optional bf16 [ceil(M / row_tap_group)][N]: rows m with m % row_tap_group == 0 ALSO store
                             bf16(acc * alpha + bias) -- before activation / dropout / residual -- at row m / group.
                             The ViT block uses it to take the per-frame cls rows of the spatial projection out of the
                             GEMM whose residual epilogue writes x + proj(.) for all rows (vision_transformer.py:263-270) */
  int row_tap_group;
  int split_hint;         /* measurements: > 0 pins the split-K count of a wgrad product (0: the library picks)              */
  int gm_hint;            /* measurements: > 0 pins the m-tiles per n-tile of an XCD's tile walk in the 256x256 kernel       */
  int preact_deriv;       /* with act (GELU kinds) and preact_out: preact_out receives bf16(act'(bf16(acc + bias))) instead of the
                             pre-activation itself -- what the matching dgrad multiplies by (act_bwd = MPV_ACT_DERIV)               */
  int keep_output;        /* cache hint.  0 (default): bf16 output tiles leave as NON-TEMPORAL stores -- a tile is written once, and as
                             plain stores it pushes the operand panels its neighbours share out of the XCD's L2 (round 5: -1.2 ms per
                             step).  1: plain stores -- for a SMALL output that a latency-bound kernel reads right behind this launch
                             (the decoder's qkv product in front of its attention: 26 vs 34 us per layer)                             */
  float colscale;         /* with colscale_cols > 0 (round 6): output columns n < colscale_cols leave as bf16(bf16(acc + bias) * colscale)  */
  int colscale_cols;      /* -- a second rounding, exactly `q = q * self.scale` on the q third of the packed qkv product
                             (models/vision_transformer.py:175-179): the attention kernels then take q as already scaled
                             (mpv_attn_desc.scale_q_bf16 = 2) and none of the three of them re-scales its q rows per work item.
                             Multiple of 8; bias-only epilogue (no activation / residual / dropout / accumulate / fp32 output)      */
} mpv_gemm_epilogue;

size_t mpv_gemm_workspace_size(int64_t M, int64_t N, int64_t K, int transA, int transB);
int mpv_gemm_bf16(const void* A, const void* B, void* C, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb,
                  int64_t ldc, int transA, int transB, const mpv_gemm_epilogue* ep, void* workspace,
                  size_t workspace_bytes, mpv_stream_t stream);

/* Test / measurement hook: the row-band plan mpv_gemm_bf16 uses for an M x N x K forward or dgrad product on a chip of
 * `ncu` compute units (csrc/gemm256.hip: rows are cut into up to three bands of 256-, 192- and 160-row tiles, one launch
 * each, so that no launch ends in a mostly empty round of workgroups).  preact / ext_rows: the epilogue also writes the
 * pre-activation / reads residual or GELU' rows.  out6[2i] = tile rows, out6[2i+1] = m-tiles of band i (row order);
 * returns the number of bands.  Host-only, launches nothing. */
int mpv_gemm_plan_bands(int64_t M, int64_t N, int64_t K, int ncu, int preact, int ext_rows, int* out6);

/* ------------------------------------------------------------------------------------------
 * LayerNorm, fp32 statistics, bf16 in/out.  Replaces LayerNormWithForceFP32
 * (models/vision_transformer.py:69-71) and megatron MixedFusedLayerNorm
 * (models/modeling_distributed_gpt3.py:1002,1016,1131).  x rows use xmap, y rows use ymap. */
int mpv_layernorm_fwd(const void* x, const void* gamma, const void* beta, void* y, float* mean, float* rstd,
                      int64_t rows, int64_t cols, int64_t ldx, int64_t ldy, float eps, int x_group, int x_stride,
                      int x_offset, int y_group, int y_stride, int y_offset, mpv_stream_t stream);
size_t mpv_layernorm_bwd_workspace_size(int64_t cols);
/* accumulate_dparams == MPV_LN_DPARAM_DEFER: the call leaves its per-workgroup dgamma/dbeta partials -- fp32
 * [mpv_layernorm_bwd_partial_rows(rows)][2][cols] at the start of `workspace`, which the caller then owns until the
 * finish -- and launches no reduction; mpv_layernorm_dparam_finish folds the partials of up to 8 such calls in two launches
 * (a ViT block has three LayerNorms: six reduce launches otherwise; it uses the 16 partial rows behind the call's own as
 * scratch, which the workspace size already covers).  dgamma / dbeta only select the mode in the deferred call (non-NULL). */
#define MPV_LN_DPARAM_DEFER 2
int mpv_layernorm_bwd_partial_rows(int64_t rows);
int mpv_layernorm_dparam_finish(const float* const* partials, const int* partial_rows, void* const* dgamma, void* const* dbeta,
                                const int* accumulate, int n, int64_t cols, mpv_stream_t stream);
/* dx[xmap(r)] = (dres ? dres[xmap(r)] : 0) + LNbwd(dy[ymap(r)]); dx may alias dres.
 * dx_drop (optional) = dx * keepmask/(1-p) with element index offset + r*cols + c.
 * dgamma/dbeta (bf16 [cols]) may be NULL (frozen GPT: dgrad only); workspace needed if not. */
int mpv_layernorm_bwd(const void* dy, const void* x, const void* gamma, const float* mean, const float* rstd,
                      const void* dres, void* dx, void* dx_drop, float drop_p, uint64_t seed, uint64_t offset,
                      void* dgamma, void* dbeta, int accumulate_dparams, int64_t rows, int64_t cols, int64_t ldx,
                      int64_t ldy, int x_group, int x_stride, int x_offset, int y_group, int y_stride, int y_offset,
                      void* workspace, size_t workspace_bytes, mpv_stream_t stream);

/* The decoder's residual stream in fp32 (models/modeling_distributed_gpt3.py:1038-1078: h + dropout(sublayer(LN(h)) + bias), twice
 * per layer).  The reference adds in bf16; 48 such roundings over 24 layers are what separates a bf16 run from the fp32 function
 * by 1.2e-2 in the logits (tools/parity_bisect.py), so here the sum lives in fp32:
 *   h_out[hmap(r)] = h_in[hmap(r)] + add[amap(r)]   (fp32; skipped when add == NULL),   y[ymap(r)] = LN(h_out row) in bf16.
 * h_in is fp32 [.,ldh], or bf16 when h_in_bf16 (the first LayerNorm behind the embedding); add is a sublayer output in bf16 with
 * bias and dropout already applied by its GEMM.  mpv_ln_stream_bwd is mpv_layernorm_bwd (no parameter gradients: frozen
 * decoder) with x read from the fp32 stream; dy / dres / dx / dx_drop stay bf16 and share x's row map. */
int mpv_ln_stream_fwd(const void* h_in, int h_in_bf16, const void* add, float* h_out, const void* gamma, const void* beta, void* y,
                      float* mean, float* rstd, int64_t rows, int64_t cols, int64_t ldh, int64_t lda, int64_t ldy, float eps,
                      int h_group, int h_stride, int h_offset, int a_group, int a_stride, int a_offset, int y_group, int y_stride,
                      int y_offset, mpv_stream_t stream);
/* The same with the dropout of `add` applied HERE (round 4): add_dropout_p > 0 drops the added tensor -- element index offset + (row of
 * `add`) * cols + column, the index, threshold and bf16 rounding of mpv_gemm_bf16's dropout epilogue, so that a sublayer's GEMM can store
 * bf16(acc + bias) with a plain epilogue and bias_dropout_add (models/modeling_distributed_gpt3.py:953-979, 1059-1078) completes in the
 * LayerNorm that adds it into the stream: an HBM-bound kernel with idle VALU slots instead of a store-bound epilogue.  Bit-identical to
 * dropping in the GEMM. */
int mpv_ln_stream_fwd_drop(const void* h_in, int h_in_bf16, const void* add, float* h_out, const void* gamma, const void* beta, void* y,
                           float* mean, float* rstd, int64_t rows, int64_t cols, int64_t ldh, int64_t lda, int64_t ldy, float eps,
                           int h_group, int h_stride, int h_offset, int a_group, int a_stride, int a_offset, int y_group, int y_stride,
                           int y_offset, float add_dropout_p, uint64_t seed, uint64_t offset, mpv_stream_t stream);
int mpv_ln_stream_bwd(const void* dy, const float* x, const void* gamma, const float* mean, const float* rstd, const void* dres,
                      void* dx, void* dx_drop, float drop_p, uint64_t seed, uint64_t offset, int64_t rows, int64_t cols, int64_t ldx,
                      int64_t ldy, int x_group, int x_stride, int x_offset, int y_group, int y_stride, int y_offset, mpv_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Fused attention (scores never materialised), exact two-pass softmax in fp32, MFMA QK^T / PV.
 * Replaces: ViT Attention core (models/vision_transformer.py:179-204: q*scale rounded to bf16,
 * fp32 scores+softmax, bf16 probs), GPT3CoreAttention (models/modeling_distributed_gpt3.py:
 * 757-804: baddbmm + scale/causal-mask(-10000)/softmax + dropout + bmm), nn.MultiheadAttention
 * core of AttentionPool (models/vision_transformer.py:353,371).
 * Element (b,h,row,d) of tensor X lives at X + b*x_bs + h*x_hs + row*x_rs + d. head_dim in
 * {64,80,96}.  lse is fp32 [batch][heads][sq]. */
typedef struct mpv_attn_desc {
  const void *q, *k, *v;
  void* o;
  float* lse;
  int64_t q_bs, q_hs, q_rs, k_bs, k_hs, k_rs, v_bs, v_hs, v_rs, o_bs, o_hs, o_rs;
  int batch, heads, sq, sk, head_dim;
  int causal;       /* key j visible to query i iff j <= i + (sk - sq) */
  float scale;      /* softmax(scale * q.k)                            */
  int scale_q_bf16; /* 1: q' = bf16(q*scale) first (ViT numerics); 2: q IS that q' already (written by the qkv product's
                       colscale epilogue): scores use it as it is, dQ is still the gradient of the UNSCALED q (x scale)  */
  float dropout_p;  /* on the probabilities                            */
  uint64_t seed, offset;
} mpv_attn_desc;
int mpv_attn_fwd(const mpv_attn_desc* d, mpv_stream_t stream);
/* dO has o's strides; dq/dk/dv have q/k/v's strides; delta is fp32 scratch [batch][heads][sq]. */
int mpv_attn_bwd(const mpv_attn_desc* d, const void* dO, void* dq, void* dk, void* dv, float* delta,
                 mpv_stream_t stream);

/* Divided space-time "temporal" attention: T<=16 tokens per (b,n) sequence, many tiny
 * problems (models/vision_transformer.py:247-248 with Attention :169-207).  qkv is the packed
 * [rows][3*D] projection in stream layout; sequence (o,i) (o<n_outer, i<n_inner) has its token
 * t at row o*outer_stride + inner_offset + i + t*t_stride.  out/dout are [rows][D]. */
int mpv_temporal_attn_fwd(const void* qkv, void* out, int n_outer, int64_t outer_stride, int n_inner,
                          int64_t inner_offset, int64_t t_stride, int T, int heads, int head_dim, float scale,
                          mpv_stream_t stream);
int mpv_temporal_attn_bwd(const void* qkv, const void* dout, void* dqkv, int n_outer, int64_t outer_stride,
                          int n_inner, int64_t inner_offset, int64_t t_stride, int T, int heads, int head_dim,
                          float scale, mpv_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Patch-embed fold: video [B,3,T,H,W] bf16 -> im2col rows (b,t,ph,pw) x (c,i,j) padded to kpad
 * columns (models/vision_transformer.py:546-548,392-398; models/eva_vit.py:197-203 for P=14). */
int mpv_im2col_patches(const void* video, void* cols, int B, int C, int T, int H, int W, int P, int kpad,
                       mpv_stream_t stream);
/* Token assembly: X[b,t,0] = cls + pos[0]; X[b,t,1+n] = patch[(b,t,n)] + pos[1+n] + temporal[t]
 * (models/vision_transformer.py:555-565) into the [B,T,1+N,D] stream. */
int mpv_vit_embed_assemble_fwd(const void* patch, const void* cls_token, const void* pos_embed,
                               const void* temporal_embed, void* x, int B, int T, int N, int D, mpv_stream_t stream);
/* dpatch [B*T*N, D], dcls [D], dpos [1+N, D], dtemporal [T, D] from dx [B,T,1+N,D]. */
int mpv_vit_embed_assemble_bwd(const void* dx, void* dpatch, void* dcls, void* dpos, void* dtemporal, int B, int T,
                               int N, int D, mpv_stream_t stream);
/* cls merge after spatial attention (models/vision_transformer.py:263-270): for every b,
 * m = mean_t a[b,t,0,:];  y[b,t,0,:] = xt[b,t,0,:] + m.  Token rows: y = xt + a (all rows). */
int mpv_vit_cls_merge_fwd(const void* xt, const void* a, void* y, int B, int T, int N1, int D, mpv_stream_t stream);
/* backward: dxt = dy (all rows); da[token rows] = dy; da[b,t,0] = (sum_t' dy[b,t',0]) / T. */
/* The same merge without the full-tensor passes: after a GEMM with residual = xt and a row tap of group N1
 * (y = xt + a everywhere, tap = a on the cls rows), fix the cls slots in place: y[b,t,0] = xt[b,t,0] + bf16(mean_t' tap[b,t']). */
int mpv_vit_cls_fix_fwd(const void* xt, const void* tap, void* y, int B, int T, int N1, int D, mpv_stream_t stream);
/* Backward in place: saved[b,t] = dy[b,t,0] (bf16 [B*T][D], to be copied back with mpv_copy_rows once the projection's
 * gradients are done), then dy[b,t,0] = bf16(mean_t' dy[b,t',0]); token rows are untouched. */
int mpv_vit_cls_merge_bwd_inplace(void* dy, void* saved, int B, int T, int N1, int D, mpv_stream_t stream);
int mpv_vit_cls_merge_bwd(const void* dy, void* da, int B, int T, int N1, int D, mpv_stream_t stream);

/* copy `rows` rows of `cols` bf16 between row-mapped buffers */
int mpv_copy_rows(const void* src, void* dst, int64_t rows, int64_t cols, int64_t lds, int64_t ldd, int s_group,
                  int s_stride, int s_offset, int d_group, int d_stride, int d_offset, mpv_stream_t stream);
/* out[c] = sum_r in[map(r)][c] (bf16 in, fp32 accumulate, bf16 out, optional accumulate):
 * bias gradients and broadcast-parameter gradients. */
size_t mpv_colsum_workspace_size(int64_t cols);
int mpv_colsum(const void* in, void* out, int64_t rows, int64_t cols, int64_t ld, int group, int stride, int offset,
               int accumulate, void* workspace, size_t workspace_bytes, mpv_stream_t stream);
/* out = a + b (bf16, n elements); used for gradient joins */
int mpv_add(const void* a, const void* b, void* out, int64_t n, mpv_stream_t stream);
/* acc (fp32) += g (bf16), n elements: the gradient-accumulation window sum (DeepSpeed's bf16 optimizer keeps it in fp32,
 * run_pretrain_distributed_gpt3.py:46-53 with --update_freq); mpv_f32_to_bf16 rounds the window sum once at the boundary */
int mpv_accum_f32(float* acc, const void* g, int64_t n, int first, mpv_stream_t stream);
/* dst[0..n) <- words[0..n) (32-bit, n <= 32), the values travelling BY VALUE in the launch: the host's way to hand per-step
 * scalars (MPV_SEED_FROM_DEVICE seeds, mpv_adamw_step_grouped_dev's hyper_dev) to the device without a host buffer that must
 * outlive an asynchronous copy.  Stream-ordered like every other entry point. */
int mpv_store_words(void* dst, const uint32_t* words, int n, mpv_stream_t stream);
int mpv_f32_to_bf16(const float* src, void* dst, int64_t n, mpv_stream_t stream);
/* n independent bf16 copies dst[i][0:count[i]] = src[i][0:count[i]] in ONE launch (host arrays of device pointers): the
 * q / v halves of the packed qkv bias (models/vision_transformer.py:173) of every attention of the tower, and the
 * q_bias / v_bias gradients out of the packed bias-gradient of the qkv weight-gradient products */
int mpv_copy_segments(const void* const* src, void* const* dst, const int64_t* count, int n, mpv_stream_t stream);
/* Finishing step of the composed temporal_attn.proj + temporal_fc backward (models/vision_transformer.py:199-200, 250;
 * Wc = Wf Wp): dWf = bf16(float(dWc Wp^T) + d(bc) bp^T) and d(bp) = Wf^T d(bc), all operands bf16 [D,D] / [D]. */
int mpv_vit_compose_bwd_finish(const void* dwc_wpT, const void* dbc, const void* bp, const void* wf, void* dwf, void* dbp, int D,
                               mpv_stream_t stream);
/* The composed temporal projection of EVERY block of the tower in one launch each (round 6; the [D, D] products of a block depend on
 * parameters and on per-block reduced gradients only, so the 12 blocks' worth goes out together: Wc and bc at the head of the step,
 * dWf / dWp / d(bp) at the end of the tower's backward -- 5 launches per step instead of 60).  Pointer tables are HOST arrays of
 * `batch` device pointers (copied by value into the launches; any batch, 16 problems per launch).
 *   mpv_vit_compose_bias_batched:       bc[b] = bf16(Wf[b] bp[b] + bf[b])             (models/vision_transformer.py:199-200, 250)
 *   mpv_vit_compose_bwd_finish_batched: mpv_vit_compose_bwd_finish for every b
 *   mpv_gemm_bf16_batched:              C[b] = A[b] op B[b], mpv_gemm_bf16's operand forms (transA / transB), ONE shape, plain epilogue
 *                                       (bf16 out, no bias); replaces torch.bmm-shaped loops over blocks, which the reference does not
 *                                       have (it runs proj and temporal_fc as two token-row products, :199-200, 250). */
int mpv_vit_compose_bias_batched(const void* const* wf, const void* const* bp, const void* const* bf, void* const* bc, int batch, int D,
                                 mpv_stream_t stream);
int mpv_vit_compose_bwd_finish_batched(const void* const* dwc_wpT, const void* const* dbc, const void* const* bp, const void* const* wf,
                                       void* const* dwf, void* const* dbp, int batch, int D, mpv_stream_t stream);
int mpv_gemm_bf16_batched(const void* const* A, const void* const* B, void* const* C, int batch, int64_t M, int64_t N, int64_t K,
                          int64_t lda, int64_t ldb, int64_t ldc, int transA, int transB, mpv_stream_t stream);
/* Labels and loss weights of the L text positions behind the Q query slots (models/distributed_gpt3.py:142-159,
 * 348-351; the masked mean of models/modeling_distributed_gpt3.py:1615-1617 as per-position weights):
 * labels[b][l] = ids[b][l+1] (ids[b][1] at l = L-1), weights[b][l] = attention_mask[b][l+1] / sum(attention_mask[:,1:])
 * with positions l < prompt_len[b] (optional) zeroed, weights[b][L-1] = 0. */
int mpv_caption_targets(const int64_t* ids, const int64_t* attention_mask, const int64_t* prompt_len, int B, int L,
                        int64_t* labels, float* weights, mpv_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * GPT-3 embedding front: h[b,s,:] = (s < Q ? query[b,s,:] : wte[ids[b,s-Q],:]) + wpe[s,:], then
 * dropout (models/distributed_gpt3.py:155-156; models/modeling_distributed_gpt3.py:640-666). */
int mpv_gpt_embed_fwd(const void* query, const int64_t* ids, const void* wte, const void* wpe, void* h, int B,
                      int Q, int L, int H, float dropout_p, uint64_t seed, uint64_t offset, mpv_stream_t stream);
/* dquery[b,q,:] = dh[b,q,:] * keepmask/(1-p) */
int mpv_gpt_embed_bwd(const void* dh, void* dquery, int B, int Q, int L, int H, float dropout_p, uint64_t seed,
                      uint64_t offset, mpv_stream_t stream);
/* The same dropout mask applied to the gradient of EVERY row of the embedding output: dfull[rows, H] = dropout_mask * dh (rows = B * S).
 * A trainable decoder (freeze_text_decoder: false, models/distributed_gpt3.py:91-93) needs it for the word- and position-embedding
 * gradients (GPT3Embedding, models/modeling_distributed_gpt3.py:640-666). */
int mpv_gpt_embed_bwd_full(const void* dh, void* dfull, int64_t rows, int H, float dropout_p, uint64_t seed, uint64_t offset,
                           mpv_stream_t stream);

/* Masked cross-entropy over bf16 logits in fp32 (models/modeling_distributed_gpt3.py:1352-1359,
 * 1615-1617): losses[r] = lse(logits[r]) - logits[r][labels[r]];  if dlogits != NULL it is
 * filled with (softmax - onehot) * weight[r] (weight = loss_mask / sum(loss_mask)), may alias
 * logits.  loss_sum (fp32 scalar) receives sum_r losses[r]*weight[r], reduced in a fixed order
 * (deterministic; needs `losses`). */
int mpv_cross_entropy(const void* logits, const int64_t* labels, const float* weight, float* losses, float* loss_sum,
                      void* dlogits, int64_t rows, int64_t vocab, int64_t ld, mpv_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * ITC retrieval head (models/distributed_gpt3.py:938-980).
 * F.normalize(dim=-1) forward/backward (:947,960); norm is fp32 [rows] (saved for backward). */
int mpv_l2norm_fwd(const void* x, void* y, float* norm, int64_t rows, int64_t cols, float eps, mpv_stream_t stream);
int mpv_l2norm_bwd(const void* dy, const void* x, const float* norm, void* dx, int64_t rows, int64_t cols,
                   mpv_stream_t stream);
/* dst[r] = src[idx[r]] (last-valid-token pooling of the text hidden state, :958-959) */
int mpv_gather_rows(const void* src, const int64_t* idx, void* dst, int64_t rows, int64_t cols, int64_t ld,
                    mpv_stream_t stream);
/* dst[idx[r]] = src[r], idx distinct, dst rows of stride ld (backward of that pooling where the decoder input
 * carries gradient: the prompt pass of cls_head, models/distributed_gpt3.py:583-585, 1149-1151) */
int mpv_scatter_rows(const void* src, const int64_t* idx, void* dst, int64_t rows, int64_t cols, int64_t ld,
                     mpv_stream_t stream);
/* ------------------------------------------------------------------------------------------
 * Generation (KV-cache decode, models/modeling_distributed_gpt3.py:1620-1886).  The forward kernels above serve
 * prefill and the single-row decode steps (mpv_attn_fwd with sq = 1 over the cached keys); two helpers:
 * dst[r][0:cols] = src[idx[r]][0:cols] with row pitches lds / ldd: the beam re-order of a layer's KV cache
 * (InferenceParams.swap_key_value_dict :1459-1473). */
int mpv_gather_rows_ld(const void* src, const int64_t* idx, void* dst, int64_t rows, int64_t cols, int64_t lds, int64_t ldd,
                       mpv_stream_t stream);
/* out_val[r][j], out_idx[r][j] = the j-th largest of log_softmax(logits[r]) + add[r] (add optional), fp32 statistics over
 * bf16 logits, ties broken towards the lower index; k <= 64.  Serves beam_search (:1790-1805: log_softmax, + scores,
 * sort, top 2*beam) and greedy sample (:1411-1413, k = 1). */
size_t mpv_logprob_topk_workspace_size(int64_t rows, int k);
int mpv_logprob_topk(const void* logits, const float* add, int64_t rows, int64_t vocab, int64_t ld, int k, float* out_val,
                     int64_t* out_idx, void* workspace, size_t workspace_bytes, mpv_stream_t stream);
/* One incremental decoder forward over per-layer KV caches, all launches issued from C (one FFI call per step):
 * appends the n = Q + L new positions of every sequence (query [batch*Q][hidden] rows first, then wte[tokens]) at
 * cache rows [pos0, pos0 + n), attends causally over rows [0, pos0 + n) and writes the logits of the LAST new position
 * of every sequence to logits [batch][vocab] (bf16).  kv_cache[l] = [batch][max_len][3*hidden] bf16 in the decoder's
 * head-interleaved q|k|v row format (models/modeling_distributed_gpt3.py:895-902).  Weight tables are host arrays of
 * device pointers.  Covers :640-666, :868-938 (KV memory), :1034-1078, :1184, :1348-1350. */
typedef struct mpv_gpt_layer_weights {
  const void *ln1_w, *ln1_b, *qkv_w, *qkv_b, *dense_w, *dense_b, *ln2_w, *ln2_b, *fc1_w, *fc1_b, *fc2_w, *fc2_b;
} mpv_gpt_layer_weights;
typedef struct mpv_gpt_weights {
  int layers, hidden, heads, ffn, vocab;
  float ln_eps;
  const mpv_gpt_layer_weights* layer;
  const void *wte, *wpe, *lnf_w, *lnf_b;
  int max_positions; /* rows of wpe (max_position_embeddings); 0 = unchecked */
} mpv_gpt_weights;
size_t mpv_gpt_decode_workspace_size(const mpv_gpt_weights* w, int batch, int n_new);
int mpv_gpt_decode_step(const mpv_gpt_weights* w, void* const* kv_cache, int batch, int max_len, int pos0, const void* query,
                        int Q, const int64_t* tokens, int L, void* workspace, size_t workspace_bytes, void* logits,
                        mpv_stream_t stream);
/* beam re-order of all layers: dst[l][j] = src[l][idx[j]] for the first `rows` cached positions */
int mpv_kv_reorder(void* const* src, void* const* dst, int layers, const int64_t* idx, int batch, int max_len, int rows,
                   int hidden, mpv_stream_t stream);
/* ------------------------------------------------------------------------------------------
 * Device-side video input transform (configs' video_pretrain/train/test transforms,
 * dataset/__init__.py:60-85): the decoder's uint8 clip [T][H][W][3] -> crop (i, j, h, w) ->
 * torch.nn.functional.interpolate to out_h x out_w (mode 0 nearest / 1 bilinear / 2 bicubic, align_corners=False:
 * dataset/video_utils/functional.py:51-72, 95-112) -> .long() truncation -> optional horizontal flip
 * (video_transforms.py:933-936) -> /255 and (x - mean) / std (volume_transforms.py:40-42, functional.py:125-136) ->
 * bf16, written as planes: element (c, t, y, x) at out[c*c_stride + t*t_stride + y*out_w + x] (a [B,3,T,H,W] batch slot).
 * mean3 / std3 are host arrays.  (TemporalConsistentRandomAugment is cv2-based and not part of this entry point.) */
int mpv_video_resized_crop_normalize(const uint8_t* clip, int T, int H, int W, int crop_i, int crop_j, int crop_h, int crop_w,
                                     int out_h, int out_w, int mode, int flip, const float* mean3, const float* std3, void* out,
                                     int64_t c_stride, int64_t t_stride, mpv_stream_t stream);

/* The clip-consistent RandAugment of the loaders (dataset/video_utils/randaugment_video.py; dataset/__init__.py:65-66,75-76) on the
 * device, on uint8 clips [T][H][W][3].  mpv_video_resized_crop_u8 is mpv_video_resized_crop_normalize stopped at the point where
 * the augmentation takes over: the .long() value as uint8 with numpy's wrapping cast (randaugment_video.py:343), [T][oh][ow][3].
 * mpv_video_aug_pointwise: op 0 = contrast_func (:120-130; the per-frame luminance mean needs sums_ws, 3 * T uint64), op 1 =
 * brightness_func (:133-139), in place.  mpv_video_aug_sharpness: sharpness_func (:142-160; cv2.filter2D restated), out of place.
 * mpv_video_aug_warp_affine: cv2.warpAffine(INTER_LINEAR, BORDER_CONSTANT) with the INVERTED 2 x 3 matrix (the host inverts as
 * opencv does) -- shear_x/y :163-167,:198-202, translate_x/y :170-187, rotate :67-75 -- out of place.  mpv_video_u8_normalize:
 * ClipToTensor + Normalize (dataset/__init__.py:68-69) into the [3][T][H][W] slot of the batch.  Bit-exact against
 * oracle/augment.py; what that restatement pins is stated in its header (opencv itself is not available to compare with). */
int mpv_video_resized_crop_u8(const uint8_t* clip, int T, int H, int W, int crop_i, int crop_j, int crop_h, int crop_w, int out_h,
                              int out_w, int mode, int flip, uint8_t* out, mpv_stream_t stream);
int mpv_video_aug_pointwise(uint8_t* frames, int T, int H, int W, int op, double factor, uint64_t* sums_ws, mpv_stream_t stream);
int mpv_video_aug_sharpness(const uint8_t* in, uint8_t* out, int T, int H, int W, double factor, mpv_stream_t stream);
int mpv_video_aug_warp_affine(const uint8_t* in, uint8_t* out, int T, int H, int W, const double* inverse_matrix6, const uint8_t* fill3,
                              mpv_stream_t stream);
int mpv_video_u8_normalize(const uint8_t* frames, int T, int H, int W, const float* mean3, const float* std3, void* out,
                           int64_t c_stride, int64_t t_stride, mpv_stream_t stream);
/* Soft-target contrastive cross-entropy over fp32 similarities sim[rows][cols] (:966-978):
 * targets[i][j] = [row_ids[i]==col_ids[j]] / count_i; losses[i] = -sum_j log_softmax(sim_i)[j] targets[i][j];
 * dsim (bf16, optional) = (softmax - targets) * scale; dts[i] (optional) = sum_j dsim[i][j] * sim[i][j]. */
int mpv_soft_target_ce(const float* sim, const int64_t* row_ids, const int64_t* col_ids, float scale, float* losses,
                       void* dsim, float* dts, int64_t rows, int64_t cols, mpv_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Optimizer (DeepSpeed FusedAdam + global-norm clip, run_pretrain_distributed_gpt3.py:137;
 * math of optim/adamw.py:66-115; torch.nn.utils.clip_grad_norm_ semantics, utils.py:308).
 * sumsq (fp32 scalar, pre-zeroed) += sum g^2 over n bf16 gradients -- BIT-REPRODUCIBLE since round 6: per-workgroup partial sums in
 * `workspace` (mpv_grad_sumsq_workspace_size() bytes), added up in a fixed order by a second launch (the clip coefficient, hence every
 * update of a clipped step, hangs on the last bits of this scalar; an atomicAdd per workgroup made them depend on arrival order). */
size_t mpv_grad_sumsq_workspace_size(void);
int mpv_grad_sumsq(const void* grad, int64_t n, float* sumsq, void* workspace, size_t workspace_bytes, mpv_stream_t stream);
/* One AdamW step over a flat range: master/m/v fp32, grad bf16, param bf16 written back.
 * clip coefficient = min(1, max_norm / (sqrt(*sumsq) * inv_world ... ) computed in-kernel from
 * the device scalar; grad_scale multiplies g first (1/world for averaged all-reduce sums). */
int mpv_adamw_step(void* param_bf16, float* master, float* exp_avg, float* exp_avg_sq, const void* grad_bf16,
                   int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                   float grad_scale, const float* sumsq, float max_norm, mpv_stream_t stream);

/* Grouped AdamW over a flat buffer laid out in backward-completion order (so data-parallel
 * gradient buckets are contiguous slices): tile_group[i] (device, uint8) is the parameter-group
 * id (0..ngroups-1; >= 8 = skip) of the i-th 256-element tile; lrs/wds are HOST arrays of the
 * per-group learning rate (lr_schedule * lr_scale, run_pretrain_distributed_gpt3.py:88-96) and
 * weight decay (optim/optim_factory.py:219-265). */
int mpv_adamw_step_grouped(void* param_bf16, float* master, float* exp_avg, float* exp_avg_sq, const void* grad_bf16,
                           int64_t n, const uint8_t* tile_group, const float* lrs, const float* wds, int ngroups,
                           float beta1, float beta2, float eps, int step, float grad_scale, const float* sumsq,
                           float max_norm, mpv_stream_t stream);

/* The same step with every STEP-DEPENDENT hyper-parameter read from device memory at kernel run time: hyper_dev[18] =
 * { lr[8], weight_decay[8], 1 / (1 - beta1^step), 1 / sqrt(1 - beta2^step) }, as written by mpv_adamw_hyper_pack (host; the
 * bias corrections are computed exactly as mpv_adamw_step_grouped computes them, so both entry points update bit-identically).
 * Kernel arguments are frozen when a training step is captured into a HIP graph; with this entry point a replay applies the
 * learning rate of ITS step (the host rewrites the 72 bytes before each replay). */
int mpv_adamw_hyper_pack(const float* lrs, const float* wds, int ngroups, float beta1, float beta2, int step, float* out18);
int mpv_adamw_step_grouped_dev(void* param_bf16, float* master, float* exp_avg, float* exp_avg_sq, const void* grad_bf16,
                               int64_t n, const uint8_t* tile_group, const float* hyper_dev, float beta1, float beta2, float eps,
                               float grad_scale, const float* sumsq, float max_norm, mpv_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MPV_H_ */
