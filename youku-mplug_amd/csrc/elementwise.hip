// elementwise.hip -- HBM-bound glue kernels of the mPLUG-Video path for gfx950: patch
// im2col, token assembly (+cls/pos/temporal embeddings), cls merge, row copies, column sums
// (bias / broadcast-parameter gradients), GPT embedding front and masked cross-entropy.
// All of them move 8-16 bytes per lane per access, rows contiguous across a wave.
#include "mpv_common.h"
#include "../../include/mpv.h"

namespace {

// ---------------------------------------------------------------- im2col for k=s=P patch conv
// out row r = (b*T + t)*N + ph*PW + pw ; out col = c*P*P + i*P + j ; zero pad to kpad.
__global__ void im2col_kernel(const bf16* __restrict__ video, bf16* __restrict__ cols, int B, int C, int T, int H, int W,
                              int P, int kpad) {
  const int PH = H / P, PW = W / P;
  const long long rows = (long long)B * T * PH * PW;
  const int K = C * P * P;
  const long long total = rows * kpad;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const long long r = idx / kpad;
    const int kk = (int)(idx - r * kpad);
    bf16 v = f2bf(0.f);
    if (kk < K) {
      const int c = kk / (P * P), ij = kk - c * P * P, i = ij / P, j = ij - i * P;
      const int pw = (int)(r % PW);
      const long long r2 = r / PW;
      const int ph = (int)(r2 % PH);
      const long long bt = r2 / PH;
      const int t = (int)(bt % T);
      const long long b = bt / T;
      v = video[(((b * C + c) * T + t) * H + (ph * P + i)) * (long long)W + (pw * P + j)];
    }
    cols[idx] = v;
  }
}

// P % 8 == 0 (the 16-pixel patches of the video tower) and 16-byte aligned rows: a thread moves 8 consecutive pixels of one
// patch row (one 16-byte load, one 16-byte store) and pays the index arithmetic once per 8 elements -- the element-wise form
// above is bound by its six integer divisions per bf16 (169 us for 32 x 8 frames of 224^2; this form is HBM-bound).
__global__ void im2col8_kernel(const bf16* __restrict__ video, bf16* __restrict__ cols, int B, int C, int T, int H, int W, int P,
                               int kpad) {
  const int PH = H / P, PW = W / P, P8 = P / 8, K8 = C * P * P8, kp8 = kpad / 8;
  const long long total = (long long)B * T * PH * PW * kp8;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const long long r = idx / kp8;
    const int k8 = (int)(idx - r * kp8);
    bf16x8 v = bf16x8{};
    if (k8 < K8) {
      const int c = k8 / (P * P8), ij = k8 - c * P * P8, i = ij / P8, j8 = ij - i * P8;
      const int pw = (int)(r % PW);
      const long long r2 = r / PW;
      const int ph = (int)(r2 % PH);
      const long long bt = r2 / PH;
      const int t = (int)(bt % T);
      const long long b = bt / T;
      v = *(const bf16x8*)(video + (((b * C + c) * T + t) * H + (ph * P + i)) * (long long)W + (pw * P + j8 * 8));
    }
    *(bf16x8*)(cols + idx * 8) = v;
  }
}

// ---------------------------------------------------------------- token assembly
__global__ void embed_assemble_fwd_kernel(const bf16* __restrict__ patch, const bf16* __restrict__ cls,
                                          const bf16* __restrict__ pos, const bf16* __restrict__ temporal,
                                          bf16* __restrict__ x, int B, int T, int N, int D) {
  const int D4 = D / 4;
  const long long total = (long long)B * T * (N + 1) * D4;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int c4 = (int)(idx % D4);
    const long long row = idx / D4;
    const int s = (int)(row % (N + 1));
    const long long bt = row / (N + 1);
    const int t = (int)(bt % T);
    f32x4 v = cvt4(*(const bf16x4*)(pos + (long long)s * D + c4 * 4));
    if (s == 0) {
      v += cvt4(*(const bf16x4*)(cls + c4 * 4));
    } else {
      // reference order: (tile_pos + tile_temporal) rounded to bf16, then added to the token (:563-565)
      v += cvt4(*(const bf16x4*)(temporal + (long long)t * D + c4 * 4));
      v = cvt4(cvt4(v));
      v += cvt4(*(const bf16x4*)(patch + (bt * N + (s - 1)) * (long long)D + c4 * 4));
    }
    *(bf16x4*)(x + row * D + c4 * 4) = cvt4(v);
  }
}

// dpos[s] = sum_{b,t} dx[b,t,s] (dcls = the s = 0 row): columns of the flattened [(1+N) * D] row, 8 per thread (16-byte
// loads), 64 column groups x 4 row lanes per workgroup, the B*T rows looped per lane and folded through LDS.
constexpr int EBP_RL = 16;      // row lanes (round 6: 16 waves per workgroup instead of 4 -- 64 dependent-accumulate trips per lane became 16)
__global__ __launch_bounds__(64 * EBP_RL) void embed_bwd_pos_kernel(const bf16* __restrict__ dx, bf16* __restrict__ dcls, bf16* __restrict__ dpos,
                                                                    int BT, long long C, int D) {
  __shared__ float red[EBP_RL][64][8];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const long long c8 = (long long)blockIdx.x * 64 + cl;     // column group
  f32x8 a = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (c8 * 8 < C)
    for (int r = rl; r < BT; r += EBP_RL) a += cvt8(*(const bf16x8*)(dx + (long long)r * C + c8 * 8));
#pragma unroll
  for (int e = 0; e < 8; ++e) red[rl][cl][e] = a[e];
  __syncthreads();
  if (rl == 0 && c8 * 8 < C) {
    f32x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float t = 0.f;
#pragma unroll
      for (int l = 0; l < EBP_RL; ++l) t += red[l][cl][e];
      o[e] = t;
    }
    const bf16x8 ob = cvt8(o);
    *(bf16x8*)(dpos + c8 * 8) = ob;
    if (c8 * 8 < D) *(bf16x8*)(dcls + c8 * 8) = ob;
  }
}
// dtemporal[t] = sum_{b,n} dx[b,t,1+n]: one workgroup per (t, 64-column stripe) = 8 column groups x 32 row lanes, every
// lane looping over its share of the B*N token rows with 16-byte loads (a row's stripe is one 128-byte line); no scratch.
// (round 6: 128 row lanes instead of 32 -- the launch is T x D/64 = 96 workgroups on 256 CUs, each walking B*N rows: with 256 threads a
// lane made 196 dependent-accumulate trips, 82 us for 77 MB; the sum order of a column changes, not its precision)
constexpr int EBT_RL = 128;
__global__ __launch_bounds__(8 * EBT_RL) void embed_bwd_temporal_kernel(const bf16* __restrict__ dx, bf16* __restrict__ dtemporal, int B, int T,
                                                                        int N, int D) {
  __shared__ float red[EBT_RL][8][8];
  const int t = blockIdx.x;
  const int cg = threadIdx.x & 7, rl = threadIdx.x >> 3;
  const int c = blockIdx.y * 64 + cg * 8;
  f32x8 a = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (c < D) {
    const int rows = B * N;
    for (int r = rl; r < rows; r += EBT_RL) {
      const int b = r / N, n = r - b * N;
      a += cvt8(*(const bf16x8*)(dx + (((long long)b * T + t) * (N + 1) + 1 + n) * D + c));
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[rl][cg][e] = a[e];
  __syncthreads();
  if (threadIdx.x < 64 && blockIdx.y * 64 + (int)threadIdx.x < D) {
    const int g = threadIdx.x >> 3, e = threadIdx.x & 7;
    float o = 0.f;
#pragma unroll 8
    for (int l = 0; l < EBT_RL; ++l) o += red[l][g][e];
    dtemporal[(long long)t * D + blockIdx.y * 64 + threadIdx.x] = f2bf(o);
  }
}

// ---------------------------------------------------------------- cls merge
// y = xt + a on token rows; on cls slots y[b,t,0] = xt[b,t,0] + mean_t' a[b,t',0].
__global__ void cls_merge_fwd_kernel(const bf16* __restrict__ xt, const bf16* __restrict__ a, bf16* __restrict__ y, int B,
                                     int T, int N1, int D) {
  const int D4 = D / 4;
  const long long total = (long long)B * T * N1 * D4;
  const float invT = 1.0f / (float)T;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int c4 = (int)(idx % D4);
    const long long row = idx / D4;
    const int s = (int)(row % N1);
    f32x4 v;
    if (s != 0) {
      v = cvt4(*(const bf16x4*)(a + row * D + c4 * 4));
    } else {
      const long long b = row / ((long long)N1 * T);
      f32x4 m = {0.f, 0.f, 0.f, 0.f};
      for (int t = 0; t < T; ++t) m += cvt4(*(const bf16x4*)(a + ((b * T + t) * N1) * (long long)D + c4 * 4));
      v = cvt4(cvt4(m * invT));  // torch.mean rounds to bf16 before the residual add (:265,270)
    }
    v += cvt4(*(const bf16x4*)(xt + row * D + c4 * 4));
    *(bf16x4*)(y + row * D + c4 * 4) = cvt4(v);
  }
}
__global__ void cls_merge_bwd_kernel(const bf16* __restrict__ dy, bf16* __restrict__ da, int B, int T, int N1, int D) {
  const int D4 = D / 4;
  const long long total = (long long)B * T * N1 * D4;
  const float invT = 1.0f / (float)T;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int c4 = (int)(idx % D4);
    const long long row = idx / D4;
    const int s = (int)(row % N1);
    bf16x4 o;
    if (s != 0) {
      o = *(const bf16x4*)(dy + row * D + c4 * 4);
    } else {
      const long long b = row / ((long long)N1 * T);
      f32x4 m = {0.f, 0.f, 0.f, 0.f};
      for (int t = 0; t < T; ++t) m += cvt4(*(const bf16x4*)(dy + ((b * T + t) * N1) * (long long)D + c4 * 4));
      o = cvt4(m * invT);
    }
    *(bf16x4*)(da + row * D + c4 * 4) = o;
  }
}

// The merge without full-tensor passes (one thread per (b, 8 columns)).  Forward: y already holds xt + a on every row
// (residual epilogue of the projection GEMM), tap holds a on the cls rows; only the B*T cls slots are rewritten.
__global__ void cls_fix_fwd_kernel(const bf16* __restrict__ xt, const bf16* __restrict__ tap, bf16* __restrict__ y, int B, int T,
                                   int N1, int D) {
  const int D8 = D / 8;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * D8) return;
  const int b = idx / D8, c = (idx - b * D8) * 8;
  f32x8 m = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int t = 0; t < T; ++t) m += cvt8(*(const bf16x8*)(tap + (long long)(b * T + t) * D + c));
  const f32x8 mb = cvt8(cvt8(m * (1.0f / (float)T)));      // torch.mean rounds to bf16 before the residual add (:265,270)
  for (int t = 0; t < T; ++t) {
    const long long off = (long long)(b * T + t) * N1 * D + c;
    *(bf16x8*)(y + off) = cvt8(cvt8(*(const bf16x8*)(xt + off)) + mb);
  }
}
// Backward: the cls rows of dy are saved, then replaced by their mean over t (what the projection's gradients see).
__global__ void cls_merge_bwd_inplace_kernel(bf16* __restrict__ dy, bf16* __restrict__ saved, int B, int T, int N1, int D) {
  const int D8 = D / 8;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * D8) return;
  const int b = idx / D8, c = (idx - b * D8) * 8;
  f32x8 m = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int t = 0; t < T; ++t) {
    const bf16x8 v = *(const bf16x8*)(dy + (long long)(b * T + t) * N1 * D + c);
    *(bf16x8*)(saved + (long long)(b * T + t) * D + c) = v;
    m += cvt8(v);
  }
  const bf16x8 o = cvt8(m * (1.0f / (float)T));
  for (int t = 0; t < T; ++t) *(bf16x8*)(dy + (long long)(b * T + t) * N1 * D + c) = o;
}

// ---------------------------------------------------------------- row copy / add / colsum
__global__ void copy_rows_kernel(const bf16* __restrict__ src, bf16* __restrict__ dst, long long rows, int cols, long long lds_,
                                 long long ldd, RowMap sm, RowMap dm) {
  const int C4 = cols / 4;
  const long long total = rows * C4;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int c4 = (int)(idx % C4);
    const long long r = idx / C4;
    *(bf16x4*)(dst + map_row(dm, r) * ldd + c4 * 4) = *(const bf16x4*)(src + map_row(sm, r) * lds_ + c4 * 4);
  }
}
__global__ void add_kernel(const bf16* __restrict__ a, const bf16* __restrict__ b, bf16* __restrict__ o, long long n4) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x)
    *(bf16x4*)(o + i * 4) = cvt4(cvt4(*(const bf16x4*)(a + i * 4)) + cvt4(*(const bf16x4*)(b + i * 4)));
}
constexpr int COLSUM_ROWS_SPLIT = 128;
// partial sums: grid (ceil(cols/256), nsplit); a workgroup = 64 column-quads x 4 row lanes, each
// thread keeps 4 independent row loads in flight; the 4 row lanes are reduced through LDS.
__global__ __launch_bounds__(256) void colsum_partial_kernel(const bf16* __restrict__ in, float* __restrict__ part, long long rows,
                                                             int cols, long long ld, RowMap m) {
  __shared__ f32x4 red[4][64];
  const int cq = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = (blockIdx.x * 64 + cq) * 4;
  const bool ok = c < cols;
  f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
  const long long step = (long long)gridDim.y * 4;
  long long r = (long long)blockIdx.y * 4 + rl;
  if (ok) {
    for (; r + 3 * step < rows; r += 4 * step) {
      const bf16x4 v0 = *(const bf16x4*)(in + map_row(m, r) * ld + c);
      const bf16x4 v1 = *(const bf16x4*)(in + map_row(m, r + step) * ld + c);
      const bf16x4 v2 = *(const bf16x4*)(in + map_row(m, r + 2 * step) * ld + c);
      const bf16x4 v3 = *(const bf16x4*)(in + map_row(m, r + 3 * step) * ld + c);
      a0 += cvt4(v0);
      a1 += cvt4(v1);
      a2 += cvt4(v2);
      a3 += cvt4(v3);
    }
    for (; r < rows; r += step) a0 += cvt4(*(const bf16x4*)(in + map_row(m, r) * ld + c));
  }
  red[rl][cq] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (rl == 0 && ok) *(f32x4*)(part + (long long)blockIdx.y * cols + c) = (red[0][cq] + red[1][cq]) + (red[2][cq] + red[3][cq]);
}
// final: workgroup = 64 columns x 4 partial lanes
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ part, bf16* __restrict__ out, int nsplit, int cols,
                                                           int accumulate) {
  __shared__ float red[4][64];
  const int cl = threadIdx.x & 63, pl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  float a = 0.f;
  if (c < cols)
    for (int i = pl; i < nsplit; i += 4) a += part[(long long)i * cols + c];
  red[pl][cl] = a;
  __syncthreads();
  if (pl == 0 && c < cols) {
    a = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
    if (accumulate) a += bf2f(out[c]);
    out[c] = f2bf(a);
  }
}

// ---------------------------------------------------------------- GPT embedding front
__global__ void gpt_embed_fwd_kernel(const bf16* __restrict__ query, const int64_t* __restrict__ ids, const bf16* __restrict__ wte,
                                     const bf16* __restrict__ wpe, bf16* __restrict__ h, int B, int Q, int L, int H,
                                     float drop_scale, uint32_t thr, uint64_t seed, uint64_t offset) {
  const int H8 = H / 8, S = Q + L;
  const long long total = (long long)B * S * H8;
  const uint64_t seed_r = thr ? mpv_resolve_seed(seed) : 0;      // (bit 63 set: the seed lives in device memory, mpv_common.h)
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int c8 = (int)(idx % H8);
    const long long row = idx / H8;
    const int s = (int)(row % S);
    const long long b = row / S;
    const bf16* src = s < Q ? query + (b * Q + s) * (long long)H : wte + ids[b * L + (s - Q)] * (long long)H;
    // reference: words_embeddings + position_embeddings in bf16 (one rounding), then dropout
    f32x8 v = cvt8(*(const bf16x8*)(src + c8 * 8)) + cvt8(*(const bf16x8*)(wpe + (long long)s * H + c8 * 8));
    if (thr) {
      v = cvt8(cvt8(v));
      const uint64_t base = offset + (uint64_t)row * (uint64_t)H + (uint64_t)(c8 * 8);
      v = mpv_dropout_vec<f32x8, 8>(v, seed_r, base, thr, drop_scale);
    }
    *(bf16x8*)(h + row * H + c8 * 8) = cvt8(v);
  }
}
__global__ void gpt_embed_bwd_kernel(const bf16* __restrict__ dh, bf16* __restrict__ dquery, int B, int Q, int L, int H,
                                     float drop_scale, uint32_t thr, uint64_t seed, uint64_t offset) {
  const int H8 = H / 8, S = Q + L;
  const long long total = (long long)B * Q * H8;
  const uint64_t seed_r = thr ? mpv_resolve_seed(seed) : 0;      // (bit 63 set: the seed lives in device memory, mpv_common.h)
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int c8 = (int)(idx % H8);
    const long long qrow = idx / H8;
    const int s = (int)(qrow % Q);
    const long long b = qrow / Q;
    const long long row = b * S + s;
    f32x8 v = cvt8(*(const bf16x8*)(dh + row * H + c8 * 8));
    if (thr) {
      const uint64_t base = offset + (uint64_t)row * (uint64_t)H + (uint64_t)(c8 * 8);
      v = mpv_dropout_vec<f32x8, 8>(v, seed_r, base, thr, drop_scale);
    }
    *(bf16x8*)(dquery + qrow * H + c8 * 8) = cvt8(v);
  }
}

// dropout mask of the embedding front applied to the gradient of EVERY row (trainable decoder: word / position embedding gradients)
__global__ void gpt_embed_bwd_full_kernel(const bf16* __restrict__ dh, bf16* __restrict__ dfull, long long rows, int H, float drop_scale, uint32_t thr,
                                          uint64_t seed, uint64_t offset) {
  const int H8 = H / 8;
  const long long total = rows * H8;
  const uint64_t seed_r = thr ? mpv_resolve_seed(seed) : 0;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int c8 = (int)(idx % H8);
    const long long row = idx / H8;
    f32x8 v = cvt8(*(const bf16x8*)(dh + row * H + c8 * 8));
    if (thr) {
      const uint64_t base = offset + (uint64_t)row * (uint64_t)H + (uint64_t)(c8 * 8);
      v = mpv_dropout_vec<f32x8, 8>(v, seed_r, base, thr, drop_scale);
    }
    *(bf16x8*)(dfull + row * H + c8 * 8) = cvt8(v);
  }
}

// ---------------------------------------------------------------- masked cross-entropy
// one workgroup per row; two sweeps over the row (second one hits L2): stats, then gradient.
// THREADS: 1024 for the LM head's rows (51200 logits: 25 trips of a 256-thread sweep were three latency-bound passes per row, 64 us for the
// 1024 loss-window rows; round 6), 256 for short rows
template <int THREADS>
__global__ __launch_bounds__(THREADS) void cross_entropy_kernel(const bf16* __restrict__ logits, const int64_t* __restrict__ labels,
                                                            const float* __restrict__ weight, float* __restrict__ losses,
                                                            float* __restrict__ loss_sum, bf16* dlogits, int vocab,
                                                            long long ld) {
  constexpr int NW = THREADS / 64;
  __shared__ float red[2 * NW + 1];      // [max per wave | sum per wave | target logit]
  const long long r = blockIdx.x;
  const bf16* row = logits + r * ld;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int V8 = vocab / 8;
  // a label outside [0, vocab) (e.g. an ignore index) contributes neither loss nor gradient; the reference's
  // nn.Embedding / vocab-parallel CE would fault on it (models/modeling_distributed_gpt3.py:1352-1359)
  const int64_t tgt_raw = labels[r];
  const bool tgt_ok = tgt_raw >= 0 && tgt_raw < (int64_t)vocab;
  const long long tgt = tgt_ok ? (long long)tgt_raw : -1;
  float mx = -INFINITY;
  for (int c = tid; c < V8; c += THREADS) {
    const f32x8 v = cvt8(*(const bf16x8*)(row + c * 8));
#pragma unroll
    for (int e = 0; e < 8; ++e) mx = fmaxf(mx, v[e]);
    // the target logit is captured here, by the thread that owns its chunk: with dlogits aliasing logits the gradient
    // sweep below overwrites the row, so it must not be re-read after that sweep has started anywhere in the workgroup
    if (tgt_ok && (long long)c == (tgt >> 3)) red[2 * NW] = v[(int)(tgt & 7)];
  }
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = red[0];
#pragma unroll
  for (int w2 = 1; w2 < NW; ++w2) mx = fmaxf(mx, red[w2]);
  float sum = 0.f;
  for (int c = tid; c < V8; c += THREADS) {
    const f32x8 v = cvt8(*(const bf16x8*)(row + c * 8));
#pragma unroll
    for (int e = 0; e < 8; ++e) sum += __expf(v[e] - mx);
  }
  sum = wave_sum(sum);
  if (lane == 0) red[NW + wave] = sum;
  __syncthreads();
  sum = red[NW];
#pragma unroll
  for (int w2 = 1; w2 < NW; ++w2) sum += red[NW + w2];
  const float w = (weight ? weight[r] : 1.0f) * (tgt_ok ? 1.0f : 0.0f);
  const float lse = mx + __logf(sum);
  if (tid == 0 && losses) losses[r] = tgt_ok ? lse - red[2 * NW] : 0.f;
  if (dlogits) {
    bf16* drow = dlogits + r * ld;
    const float inv = 1.0f / sum;
    for (int c = tid; c < V8; c += THREADS) {
      const f32x8 v = cvt8(*(const bf16x8*)(row + c * 8));
      f32x8 g;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float pr = __expf(v[e] - mx) * inv;
        g[e] = (pr - ((long long)(c * 8 + e) == tgt ? 1.0f : 0.f)) * w;
      }
      *(bf16x8*)(drow + c * 8) = cvt8(g);
    }
  }
}

// deterministic sum_r losses[r] * weight[r] (fixed reduction order: no float atomics)
__global__ __launch_bounds__(256) void weighted_sum_kernel(const float* __restrict__ losses, const float* __restrict__ weight,
                                                           float* __restrict__ out, long long rows) {
  __shared__ float red[4];
  float s = 0.f;
  for (long long r = threadIdx.x; r < rows; r += 256) s += losses[r] * (weight ? weight[r] : 1.0f);
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) *out = (red[0] + red[1]) + (red[2] + red[3]);
}

// ---------------------------------------------------------------- ITC head (retrieval)
// F.normalize(x, dim=-1): y = x / max(||x||_2, eps); one wave per row, cols <= 2048
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const bf16* __restrict__ x, bf16* __restrict__ y, float* __restrict__ nrm,
                                                         long long rows, int cols, float eps) {
  const int lane = threadIdx.x & 63;
  const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  float ss = 0.f;
  for (int c = lane * 4; c < cols; c += 256) {
    const f32x4 v = cvt4(*(const bf16x4*)(x + r * cols + c));
    ss += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
  }
  const float n = fmaxf(sqrtf(wave_sum(ss)), eps);
  const float inv = 1.0f / n;
  for (int c = lane * 4; c < cols; c += 256) *(bf16x4*)(y + r * cols + c) = cvt4(cvt4(*(const bf16x4*)(x + r * cols + c)) * inv);
  if (lane == 0) nrm[r] = n;
}
// dx = (dy - y * <y, dy>) / n   (y = x / n recomputed in fp32)
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const bf16* __restrict__ dy, const bf16* __restrict__ x,
                                                         const float* __restrict__ nrm, bf16* __restrict__ dx, long long rows,
                                                         int cols) {
  const int lane = threadIdx.x & 63;
  const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const float inv = 1.0f / nrm[r];
  float dot = 0.f;
  for (int c = lane * 4; c < cols; c += 256) {
    const f32x4 xv = cvt4(*(const bf16x4*)(x + r * cols + c)) * inv, dv = cvt4(*(const bf16x4*)(dy + r * cols + c));
    dot += xv[0] * dv[0] + xv[1] * dv[1] + xv[2] * dv[2] + xv[3] * dv[3];
  }
  dot = wave_sum(dot);
  for (int c = lane * 4; c < cols; c += 256) {
    const f32x4 xv = cvt4(*(const bf16x4*)(x + r * cols + c)) * inv, dv = cvt4(*(const bf16x4*)(dy + r * cols + c));
    *(bf16x4*)(dx + r * cols + c) = cvt4((dv - xv * dot) * inv);
  }
}
__global__ void gather_rows_kernel(const bf16* __restrict__ src, const int64_t* __restrict__ idx, bf16* __restrict__ dst,
                                   long long rows, int cols, long long ld) {
  const int C4 = cols / 4;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < rows * C4; i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / C4;
    const int c4 = (int)(i - r * C4);
    *(bf16x4*)(dst + r * cols + c4 * 4) = *(const bf16x4*)(src + idx[r] * ld + c4 * 4);
  }
}
// dst[idx[r]] = src[r] (rows of idx are distinct): backward of the last-valid-token pooling
__global__ void scatter_rows_kernel(const bf16* __restrict__ src, const int64_t* __restrict__ idx, bf16* __restrict__ dst,
                                    long long rows, int cols, long long ld) {
  const int C4 = cols / 4;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < rows * C4; i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / C4;
    const int c4 = (int)(i - r * C4);
    *(bf16x4*)(dst + idx[r] * ld + c4 * 4) = *(const bf16x4*)(src + r * cols + c4 * 4);
  }
}
// ---------------------------------------------------------------- generation (KV-cache decode)
// beam re-order of a KV cache (InferenceParams.swap_key_value_dict, models/modeling_distributed_gpt3.py:1459-1473):
// dst[r][0:cols] = src[idx[r]][0:cols] with separate source / destination row pitches
__global__ void gather_rows_ld_kernel(const bf16* __restrict__ src, const int64_t* __restrict__ idx, bf16* __restrict__ dst,
                                      long long rows, long long cols, long long lds, long long ldd) {
  const long long C8 = cols / 8;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < rows * C8; i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / C8, c8 = i - r * C8;
    *(bf16x8*)(dst + r * ldd + c8 * 8) = *(const bf16x8*)(src + idx[r] * lds + c8 * 8);
  }
}
// log_softmax(logits[r]) + add[r], then the k largest entries of the row (descending; ties -> lower index first), k <= 64.
// Two stages so that a handful of rows still fills the chip: (1) TOPK_CHUNKS workgroups per row each take a slice of
// the vocabulary: slice max / sum-exp and the slice's k best by k rounds of a block-wide arg-max over the entries that
// come after the previous winner in (value desc, index asc) order; (2) one workgroup per row merges the statistics
// and the TOPK_CHUNKS*k candidates the same way.
constexpr int TOPK_CHUNKS = 32;

__device__ __forceinline__ void block_argmax(float& bv, int& bi, float* redf, int* redi, int tid) {
  const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(bv, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
  }
  if (lane == 0) { redf[wave] = bv; redi[wave] = bi; }
  __syncthreads();
  bv = redf[0];
  bi = redi[0];
#pragma unroll
  for (int w = 1; w < 4; ++w)
    if (redf[w] > bv || (redf[w] == bv && redi[w] < bi)) { bv = redf[w]; bi = redi[w]; }
  __syncthreads();
}

__global__ __launch_bounds__(256) void logprob_topk_part_kernel(const bf16* __restrict__ logits, int k, float* __restrict__ cval,
                                                                int* __restrict__ cidx, float* __restrict__ pmax,
                                                                float* __restrict__ psum, int vocab, long long ld) {
  __shared__ float redf[4];
  __shared__ int redi[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int chunk = blockIdx.x, rowi = blockIdx.y;
  const int per = (vocab + TOPK_CHUNKS - 1) / TOPK_CHUNKS;
  const int c0 = chunk * per, c1 = min(vocab, c0 + per);
  const bf16* row = logits + (long long)rowi * ld;
  float mx = -INFINITY;
  for (int c = c0 + tid; c < c1; c += 256) mx = fmaxf(mx, bf2f(row[c]));
  mx = wave_max(mx);
  if (lane == 0) redf[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(redf[0], redf[1]), fmaxf(redf[2], redf[3]));
  __syncthreads();
  float se = 0.f;
  for (int c = c0 + tid; c < c1; c += 256) se += __expf(bf2f(row[c]) - mx);
  se = wave_sum(se);
  if (lane == 0) redf[wave] = se;
  __syncthreads();
  if (tid == 0) {
    pmax[rowi * TOPK_CHUNKS + chunk] = mx;
    psum[rowi * TOPK_CHUNKS + chunk] = (redf[0] + redf[1]) + (redf[2] + redf[3]);
  }
  __syncthreads();
  float pv = INFINITY;
  int pi = -1;
  for (int j = 0; j < k; ++j) {
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int c = c0 + tid; c < c1; c += 256) {
      const float v = bf2f(row[c]);
      const bool after = v < pv || (v == pv && c > pi);          // not yet taken
      if (after && (v > bv || (v == bv && c < bi))) { bv = v; bi = c; }
    }
    block_argmax(bv, bi, redf, redi, tid);
    if (tid == 0) {
      cval[((long long)rowi * TOPK_CHUNKS + chunk) * k + j] = bv;     // -inf / 0x7fffffff once the slice is exhausted
      cidx[((long long)rowi * TOPK_CHUNKS + chunk) * k + j] = bi;
    }
    pv = bv;
    pi = bi;
  }
}

__global__ __launch_bounds__(256) void logprob_topk_merge_kernel(const float* __restrict__ cval, const int* __restrict__ cidx,
                                                                 const float* __restrict__ pmax, const float* __restrict__ psum,
                                                                 const float* __restrict__ add, int k, float* __restrict__ out_val,
                                                                 int64_t* __restrict__ out_idx) {
  __shared__ float redf[4];
  __shared__ int redi[4];
  const int tid = threadIdx.x, rowi = blockIdx.x;
  float gmax = -INFINITY;
  for (int c = 0; c < TOPK_CHUNKS; ++c) gmax = fmaxf(gmax, pmax[rowi * TOPK_CHUNKS + c]);
  float se = 0.f;
  for (int c = 0; c < TOPK_CHUNKS; ++c) se += psum[rowi * TOPK_CHUNKS + c] * __expf(pmax[rowi * TOPK_CHUNKS + c] - gmax);
  const float lse = gmax + __logf(se) - (add ? add[rowi] : 0.f);
  const int ncand = TOPK_CHUNKS * k;
  const float* cv = cval + (long long)rowi * ncand;
  const int* ci = cidx + (long long)rowi * ncand;
  float pv = INFINITY;
  int pi = -1;
  for (int j = 0; j < k; ++j) {
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int c = tid; c < ncand; c += 256) {
      const float v = cv[c];
      const int i = ci[c];
      const bool after = v < pv || (v == pv && i > pi);
      if (after && (v > bv || (v == bv && i < bi))) { bv = v; bi = i; }
    }
    block_argmax(bv, bi, redf, redi, tid);
    if (tid == 0) {
      out_val[(long long)rowi * k + j] = bv - lse;
      out_idx[(long long)rowi * k + j] = bi;
    }
    pv = bv;
    pi = bi;
  }
}
// ---------------------------------------------------------------- device-side video input transform
// crop -> F.interpolate (nearest / bilinear / bicubic A=-0.75, align_corners=False: dataset/video_utils/functional.py:
// 51-72, 95-112) -> .long() (truncation, NO clamp: bicubic overshoot survives) -> flip(W) (video_transforms.py:933-936)
// -> /255, permute to C,T,H,W (volume_transforms.py:40-42) -> (x - mean) / std (functional.py:125-136) -> bf16
// (run_pretrain_distributed_gpt3.py:121).  One thread per output pixel, the three interleaved channels together.
struct VideoTfArgs {
  const uint8_t* clip;    // [T][H][W][3]
  bf16* out;              // element (c, t, y, x) at c*c_stride + t*t_stride + y*out_w + x
  int T, H, W, ci, cj, ch, cw, oh, ow, mode, flip;
  long long c_stride, t_stride;
  float mean[3], stdv[3];
  uint8_t* out_u8;        // when set: the .long() value as uint8 (numpy's wrapping astype) at [t][y][x][c] instead of the normalised bf16
};
__device__ __forceinline__ float cubic1(float x, float A) { return ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f; }
__device__ __forceinline__ float cubic2(float x, float A) { return ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A; }

__global__ void video_transform_kernel(const VideoTfArgs p) {
#pragma clang fp contract(off)
  const long long n = (long long)p.T * p.oh * p.ow;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int ox = (int)(i % p.ow);
    const int oy = (int)((i / p.ow) % p.oh);
    const int t = (int)(i / ((long long)p.ow * p.oh));
    const uint8_t* frame = p.clip + ((long long)t * p.H + p.ci) * p.W * 3 + (long long)p.cj * 3;   // crop origin
    const long long rowb = (long long)p.W * 3;
    float v[3] = {0.f, 0.f, 0.f};
    if (p.mode == 0) {                       // nearest: min(floor(dst * in/out), in - 1)
      const int sy = p.oh == p.ch ? oy : min((int)floorf(oy * ((float)p.ch / p.oh)), p.ch - 1);
      const int sx = p.ow == p.cw ? ox : min((int)floorf(ox * ((float)p.cw / p.ow)), p.cw - 1);
      const uint8_t* q = frame + sy * rowb + sx * 3;
      for (int c = 0; c < 3; ++c) v[c] = (float)q[c];
    } else {
      const float sch = (float)p.ch / p.oh, scw = (float)p.cw / p.ow;
      float fy = sch * (oy + 0.5f) - 0.5f, fx = scw * (ox + 0.5f) - 0.5f;
      if (p.mode == 1) {                     // bilinear
        fy = fmaxf(fy, 0.f);
        fx = fmaxf(fx, 0.f);
        const int y0 = min((int)floorf(fy), p.ch - 1), x0 = min((int)floorf(fx), p.cw - 1);
        const float ly = fminf(fmaxf(fy - y0, 0.f), 1.f), lx = fminf(fmaxf(fx - x0, 0.f), 1.f);
        const int y1 = min(y0 + 1, p.ch - 1), x1 = min(x0 + 1, p.cw - 1);
        const uint8_t *r0 = frame + y0 * rowb, *r1 = frame + y1 * rowb;
        for (int c = 0; c < 3; ++c) {
          const float a = (1.f - lx) * r0[x0 * 3 + c] + lx * r0[x1 * 3 + c];
          const float b = (1.f - lx) * r1[x0 * 3 + c] + lx * r1[x1 * 3 + c];
          v[c] = (1.f - ly) * a + ly * b;
        }
      } else {                               // bicubic, A = -0.75
        const float A = -0.75f;
        const int iy = min((int)floorf(fy), p.ch - 1), ix = min((int)floorf(fx), p.cw - 1);
        const float ly = fminf(fmaxf(fy - iy, 0.f), 1.f), lx = fminf(fmaxf(fx - ix, 0.f), 1.f);
        const float wy[4] = {cubic2(ly + 1.f, A), cubic1(ly, A), cubic1(1.f - ly, A), cubic2(2.f - ly, A)};
        const float wx[4] = {cubic2(lx + 1.f, A), cubic1(lx, A), cubic1(1.f - lx, A), cubic2(2.f - lx, A)};
        int xs[4];
        for (int k = 0; k < 4; ++k) xs[k] = min(max(ix + k - 1, 0), p.cw - 1) * 3;
        for (int c = 0; c < 3; ++c) {
          float acc = 0.f;
          for (int ky = 0; ky < 4; ++ky) {
            const uint8_t* r = frame + (long long)min(max(iy + ky - 1, 0), p.ch - 1) * rowb;
            float rv = wx[0] * r[xs[0] + c];
            for (int kx = 1; kx < 4; ++kx) rv += wx[kx] * r[xs[kx] + c];
            acc = ky == 0 ? wy[0] * rv : acc + wy[ky] * rv;
          }
          v[c] = acc;
        }
      }
    }
    const int dx = p.flip ? p.ow - 1 - ox : ox;
    if (p.out_u8) {      // the view TemporalConsistentRandomAugment takes of the clip: frames.numpy().astype(np.uint8) (randaugment_video.py:343)
      for (int c = 0; c < 3; ++c) p.out_u8[(((long long)t * p.oh + oy) * p.ow + dx) * 3 + c] = (uint8_t)(long long)truncf(v[c]);
      continue;
    }
    for (int c = 0; c < 3; ++c) {
      const float q = truncf(v[c]);                              // .long()
      const float y = (q / 255.f - p.mean[c]) / p.stdv[c];
      p.out[c * p.c_stride + t * p.t_stride + (long long)oy * p.ow + dx] = f2bf(y);
    }
  }
}
// Soft-target contrastive CE (models/distributed_gpt3.py:966-978): targets[i][j] = [ids_r[i]==ids_c[j]] / count_i;
// loss_i = -sum_j log_softmax(sim_i)[j] * targets[i][j];  dsim = (softmax - targets) * scale (bf16);
// dts[i] = sum_j dsim[i][j] * sim[i][j] (for the temperature gradient).  One wave per row.
__global__ __launch_bounds__(256) void soft_ce_kernel(const float* __restrict__ sim, const int64_t* __restrict__ ids_r,
                                                      const int64_t* __restrict__ ids_c, float scale, float* __restrict__ losses,
                                                      bf16* __restrict__ dsim, float* __restrict__ dts, int rows, int cols) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const float* row = sim + (long long)r * cols;
  const int64_t me = ids_r[r];
  float mx = -INFINITY, cnt = 0.f;
  for (int c = lane; c < cols; c += 64) {
    mx = fmaxf(mx, row[c]);
    cnt += ids_c[c] == me ? 1.f : 0.f;
  }
  mx = wave_max(mx);
  cnt = wave_sum(cnt);
  float se = 0.f;
  for (int c = lane; c < cols; c += 64) se += __expf(row[c] - mx);
  se = wave_sum(se);
  const float lse = mx + __logf(se), invc = 1.0f / cnt;
  float loss = 0.f, dt = 0.f;
  for (int c = lane; c < cols; c += 64) {
    const float t = ids_c[c] == me ? invc : 0.f;
    const float g = (__expf(row[c] - lse) - t) * scale;
    loss -= (row[c] - lse) * t;
    dt += g * row[c];
    if (dsim) dsim[(long long)r * cols + c] = f2bf(g);
  }
  loss = wave_sum(loss);
  dt = wave_sum(dt);
  if (lane == 0) {
    losses[r] = loss;
    if (dts) dts[r] = dt;
  }
}

inline int ew_grid(long long work_items, int threads = 256) {
  long long b = (work_items + threads - 1) / threads;
  return (int)(b < 1 ? 1 : (b > 16384 ? 16384 : b));
}

}  // namespace

extern "C" int mpv_im2col_patches(const void* video, void* cols, int B, int C, int T, int H, int W, int P, int kpad,
                                  hipStream_t stream) {
  MPV_REQUIRE(video && cols, MPV_E_ARG, "mpv_im2col_patches: null pointer");
  MPV_REQUIRE(B > 0 && C > 0 && T > 0 && P > 0 && H % P == 0 && W % P == 0 && kpad >= C * P * P, MPV_E_SHAPE,
              "mpv_im2col_patches: bad shape");
  const long long total = (long long)B * T * (H / P) * (W / P) * kpad;
  if (P % 8 == 0 && W % 8 == 0 && kpad % 8 == 0 && (((uintptr_t)video | (uintptr_t)cols) & 15) == 0)
    hipLaunchKernelGGL(im2col8_kernel, dim3(ew_grid(total / 8)), dim3(256), 0, stream, (const bf16*)video, (bf16*)cols, B, C, T, H, W, P,
                       kpad);
  else
    hipLaunchKernelGGL(im2col_kernel, dim3(ew_grid(total)), dim3(256), 0, stream, (const bf16*)video, (bf16*)cols, B, C, T, H, W, P,
                       kpad);
  return mpv_check_launch("mpv_im2col_patches");
}

extern "C" int mpv_vit_embed_assemble_fwd(const void* patch, const void* cls_token, const void* pos_embed,
                                          const void* temporal_embed, void* x, int B, int T, int N, int D,
                                          hipStream_t stream) {
  MPV_REQUIRE(patch && cls_token && pos_embed && temporal_embed && x, MPV_E_ARG, "mpv_vit_embed_assemble_fwd: null pointer");
  MPV_REQUIRE(D % 4 == 0 && B > 0 && T > 0 && N > 0, MPV_E_SHAPE, "mpv_vit_embed_assemble_fwd: bad shape");
  const long long total = (long long)B * T * (N + 1) * (D / 4);
  hipLaunchKernelGGL(embed_assemble_fwd_kernel, dim3(ew_grid(total)), dim3(256), 0, stream, (const bf16*)patch,
                     (const bf16*)cls_token, (const bf16*)pos_embed, (const bf16*)temporal_embed, (bf16*)x, B, T, N, D);
  return mpv_check_launch("mpv_vit_embed_assemble_fwd");
}

extern "C" int mpv_vit_embed_assemble_bwd(const void* dx, void* dpatch, void* dcls, void* dpos, void* dtemporal, int B,
                                          int T, int N, int D, hipStream_t stream) {
  MPV_REQUIRE(dx && dpatch && dcls && dpos && dtemporal, MPV_E_ARG, "mpv_vit_embed_assemble_bwd: null pointer");
  MPV_REQUIRE(D % 8 == 0 && B > 0 && T > 0 && N > 0, MPV_E_SHAPE, "mpv_vit_embed_assemble_bwd: bad shape (D must be a multiple of 8)");
  const long long rows = (long long)B * T * N;
  hipLaunchKernelGGL(copy_rows_kernel, dim3(ew_grid(rows * (D / 4))), dim3(256), 0, stream, (const bf16*)dx, (bf16*)dpatch, rows, D,
                     (long long)D, (long long)D, RowMap{N, N + 1, 1}, RowMap{0, 0, 0});
  const long long C = (long long)(N + 1) * D;
  hipLaunchKernelGGL(embed_bwd_pos_kernel, dim3((unsigned)((C / 8 + 63) / 64)), dim3(64 * EBP_RL), 0, stream, (const bf16*)dx, (bf16*)dcls, (bf16*)dpos,
                     B * T, C, D);
  hipLaunchKernelGGL(embed_bwd_temporal_kernel, dim3(T, (D + 63) / 64), dim3(8 * EBT_RL), 0, stream, (const bf16*)dx, (bf16*)dtemporal, B, T,
                     N, D);
  return mpv_check_launch("mpv_vit_embed_assemble_bwd");
}

extern "C" int mpv_vit_cls_merge_fwd(const void* xt, const void* a, void* y, int B, int T, int N1, int D,
                                     hipStream_t stream) {
  MPV_REQUIRE(xt && a && y, MPV_E_ARG, "mpv_vit_cls_merge_fwd: null pointer");
  MPV_REQUIRE(D % 4 == 0 && B > 0 && T > 0 && N1 > 1, MPV_E_SHAPE, "mpv_vit_cls_merge_fwd: bad shape");
  hipLaunchKernelGGL(cls_merge_fwd_kernel, dim3(ew_grid((long long)B * T * N1 * (D / 4))), dim3(256), 0, stream, (const bf16*)xt,
                     (const bf16*)a, (bf16*)y, B, T, N1, D);
  return mpv_check_launch("mpv_vit_cls_merge_fwd");
}

extern "C" int mpv_vit_cls_fix_fwd(const void* xt, const void* tap, void* y, int B, int T, int N1, int D, hipStream_t stream) {
  MPV_REQUIRE(xt && tap && y, MPV_E_ARG, "mpv_vit_cls_fix_fwd: null pointer");
  MPV_REQUIRE(D % 8 == 0 && B > 0 && T > 0 && N1 > 1, MPV_E_SHAPE, "mpv_vit_cls_fix_fwd: bad shape");
  hipLaunchKernelGGL(cls_fix_fwd_kernel, dim3((unsigned)((B * (D / 8) + 255) / 256)), dim3(256), 0, stream, (const bf16*)xt,
                     (const bf16*)tap, (bf16*)y, B, T, N1, D);
  return mpv_check_launch("mpv_vit_cls_fix_fwd");
}

extern "C" int mpv_vit_cls_merge_bwd_inplace(void* dy, void* saved, int B, int T, int N1, int D, hipStream_t stream) {
  MPV_REQUIRE(dy && saved, MPV_E_ARG, "mpv_vit_cls_merge_bwd_inplace: null pointer");
  MPV_REQUIRE(D % 8 == 0 && B > 0 && T > 0 && N1 > 1, MPV_E_SHAPE, "mpv_vit_cls_merge_bwd_inplace: bad shape");
  hipLaunchKernelGGL(cls_merge_bwd_inplace_kernel, dim3((unsigned)((B * (D / 8) + 255) / 256)), dim3(256), 0, stream, (bf16*)dy,
                     (bf16*)saved, B, T, N1, D);
  return mpv_check_launch("mpv_vit_cls_merge_bwd_inplace");
}

extern "C" int mpv_vit_cls_merge_bwd(const void* dy, void* da, int B, int T, int N1, int D, hipStream_t stream) {
  MPV_REQUIRE(dy && da, MPV_E_ARG, "mpv_vit_cls_merge_bwd: null pointer");
  MPV_REQUIRE(D % 4 == 0 && B > 0 && T > 0 && N1 > 1, MPV_E_SHAPE, "mpv_vit_cls_merge_bwd: bad shape");
  hipLaunchKernelGGL(cls_merge_bwd_kernel, dim3(ew_grid((long long)B * T * N1 * (D / 4))), dim3(256), 0, stream, (const bf16*)dy,
                     (bf16*)da, B, T, N1, D);
  return mpv_check_launch("mpv_vit_cls_merge_bwd");
}

extern "C" int mpv_copy_rows(const void* src, void* dst, int64_t rows, int64_t cols, int64_t lds, int64_t ldd,
                             int s_group, int s_stride, int s_offset, int d_group, int d_stride, int d_offset,
                             hipStream_t stream) {
  MPV_REQUIRE(src && dst, MPV_E_ARG, "mpv_copy_rows: null pointer");
  MPV_REQUIRE(cols > 0 && cols % 4 == 0 && lds % 4 == 0 && ldd % 4 == 0 && rows >= 0, MPV_E_SHAPE, "mpv_copy_rows: bad shape");
  if (rows == 0) return MPV_OK;
  hipLaunchKernelGGL(copy_rows_kernel, dim3(ew_grid(rows * (cols / 4))), dim3(256), 0, stream, (const bf16*)src, (bf16*)dst,
                     (long long)rows, (int)cols, (long long)lds, (long long)ldd, RowMap{s_group, s_stride, s_offset},
                     RowMap{d_group, d_stride, d_offset});
  return mpv_check_launch("mpv_copy_rows");
}

extern "C" size_t mpv_colsum_workspace_size(int64_t cols) { return (size_t)COLSUM_ROWS_SPLIT * cols * sizeof(float); }

extern "C" int mpv_colsum(const void* in, void* out, int64_t rows, int64_t cols, int64_t ld, int group, int stride,
                          int offset, int accumulate, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  MPV_REQUIRE(in && out && workspace, MPV_E_ARG, "mpv_colsum: null pointer");
  MPV_REQUIRE(cols > 0 && cols % 4 == 0 && ld % 4 == 0 && rows > 0, MPV_E_SHAPE, "mpv_colsum: bad shape");
  int nsplit = (int)(rows < COLSUM_ROWS_SPLIT ? rows : COLSUM_ROWS_SPLIT);
  MPV_REQUIRE(workspace_bytes >= (size_t)nsplit * cols * sizeof(float), MPV_E_ARG, "mpv_colsum: workspace too small");
  nsplit = (int)((rows + 3) / 4 < COLSUM_ROWS_SPLIT ? (rows + 3) / 4 : COLSUM_ROWS_SPLIT);
  dim3 grid((unsigned)((cols / 4 + 63) / 64), nsplit);
  hipLaunchKernelGGL(colsum_partial_kernel, grid, dim3(256), 0, stream, (const bf16*)in, (float*)workspace, (long long)rows,
                     (int)cols, (long long)ld, RowMap{group, stride, offset});
  hipLaunchKernelGGL(colsum_final_kernel, dim3((unsigned)((cols + 63) / 64)), dim3(256), 0, stream, (const float*)workspace,
                     (bf16*)out, nsplit, (int)cols, accumulate);
  return mpv_check_launch("mpv_colsum");
}

extern "C" int mpv_add(const void* a, const void* b, void* out, int64_t n, hipStream_t stream) {
  MPV_REQUIRE(a && b && out, MPV_E_ARG, "mpv_add: null pointer");
  MPV_REQUIRE(n >= 0 && n % 4 == 0, MPV_E_SHAPE, "mpv_add: n must be a multiple of 4");
  if (n == 0) return MPV_OK;
  hipLaunchKernelGGL(add_kernel, dim3(ew_grid(n / 4)), dim3(256), 0, stream, (const bf16*)a, (const bf16*)b, (bf16*)out,
                     (long long)(n / 4));
  return mpv_check_launch("mpv_add");
}

extern "C" int mpv_gpt_embed_fwd(const void* query, const int64_t* ids, const void* wte, const void* wpe, void* h, int B,
                                 int Q, int L, int H, float dropout_p, uint64_t seed, uint64_t offset,
                                 hipStream_t stream) {
  MPV_REQUIRE(ids && wte && wpe && h && (query || Q == 0), MPV_E_ARG, "mpv_gpt_embed_fwd: null pointer");
  MPV_REQUIRE(H % 8 == 0 && B > 0 && Q >= 0 && L >= 0 && Q + L > 0, MPV_E_SHAPE, "mpv_gpt_embed_fwd: bad shape");
  MPV_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, MPV_E_ARG, "mpv_gpt_embed_fwd: bad dropout_p");
  const long long total = (long long)B * (Q + L) * (H / 8);
  hipLaunchKernelGGL(gpt_embed_fwd_kernel, dim3(ew_grid(total)), dim3(256), 0, stream, (const bf16*)query, ids, (const bf16*)wte,
                     (const bf16*)wpe, (bf16*)h, B, Q, L, H, 1.0f / (1.0f - dropout_p),
                     dropout_p > 0.f ? mpv_drop_threshold(dropout_p) : 0u, seed, offset);
  return mpv_check_launch("mpv_gpt_embed_fwd");
}

extern "C" int mpv_gpt_embed_bwd(const void* dh, void* dquery, int B, int Q, int L, int H, float dropout_p, uint64_t seed,
                                 uint64_t offset, hipStream_t stream) {
  MPV_REQUIRE(dh && dquery, MPV_E_ARG, "mpv_gpt_embed_bwd: null pointer");
  MPV_REQUIRE(H % 8 == 0 && B > 0 && Q > 0 && L >= 0, MPV_E_SHAPE, "mpv_gpt_embed_bwd: bad shape");
  const long long total = (long long)B * Q * (H / 8);
  hipLaunchKernelGGL(gpt_embed_bwd_kernel, dim3(ew_grid(total)), dim3(256), 0, stream, (const bf16*)dh, (bf16*)dquery, B, Q, L, H,
                     1.0f / (1.0f - dropout_p), dropout_p > 0.f ? mpv_drop_threshold(dropout_p) : 0u, seed, offset);
  return mpv_check_launch("mpv_gpt_embed_bwd");
}

extern "C" int mpv_cross_entropy(const void* logits, const int64_t* labels, const float* weight, float* losses,
                                 float* loss_sum, void* dlogits, int64_t rows, int64_t vocab, int64_t ld,
                                 hipStream_t stream) {
  MPV_REQUIRE(logits && labels, MPV_E_ARG, "mpv_cross_entropy: null pointer");
  MPV_REQUIRE(rows > 0 && vocab > 0 && vocab % 8 == 0 && ld % 8 == 0, MPV_E_SHAPE, "mpv_cross_entropy: vocab/ld must be multiples of 8");
  MPV_REQUIRE(!loss_sum || losses, MPV_E_ARG, "mpv_cross_entropy: loss_sum needs the per-row losses buffer");
if (vocab >= 16384)   hipLaunchKernelGGL(cross_entropy_kernel<1024>, dim3((unsigned)rows), dim3(1024), 0, stream, (const bf16*)logits, labels, weight, losses,
                     loss_sum, (bf16*)dlogits, (int)vocab, (long long)ld);
  else   hipLaunchKernelGGL(cross_entropy_kernel<256>, dim3((unsigned)rows), dim3(256), 0, stream, (const bf16*)logits, labels, weight, losses,
                     loss_sum, (bf16*)dlogits, (int)vocab, (long long)ld);
  if (loss_sum)
    hipLaunchKernelGGL(weighted_sum_kernel, dim3(1), dim3(256), 0, stream, (const float*)losses, weight, loss_sum, (long long)rows);
  return mpv_check_launch("mpv_cross_entropy");
}

extern "C" int mpv_l2norm_fwd(const void* x, void* y, float* norm, int64_t rows, int64_t cols, float eps, hipStream_t stream) {
  MPV_REQUIRE(x && y && norm, MPV_E_ARG, "mpv_l2norm_fwd: null pointer");
  MPV_REQUIRE(rows > 0 && cols > 0 && cols % 4 == 0, MPV_E_SHAPE, "mpv_l2norm_fwd: cols must be a multiple of 4");
  hipLaunchKernelGGL(l2norm_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, (const bf16*)x, (bf16*)y, norm,
                     (long long)rows, (int)cols, eps);
  return mpv_check_launch("mpv_l2norm_fwd");
}

extern "C" int mpv_l2norm_bwd(const void* dy, const void* x, const float* norm, void* dx, int64_t rows, int64_t cols,
                              hipStream_t stream) {
  MPV_REQUIRE(dy && x && norm && dx, MPV_E_ARG, "mpv_l2norm_bwd: null pointer");
  MPV_REQUIRE(rows > 0 && cols > 0 && cols % 4 == 0, MPV_E_SHAPE, "mpv_l2norm_bwd: cols must be a multiple of 4");
  hipLaunchKernelGGL(l2norm_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, (const bf16*)dy, (const bf16*)x, norm,
                     (bf16*)dx, (long long)rows, (int)cols);
  return mpv_check_launch("mpv_l2norm_bwd");
}

extern "C" int mpv_gather_rows(const void* src, const int64_t* idx, void* dst, int64_t rows, int64_t cols, int64_t ld,
                               hipStream_t stream) {
  MPV_REQUIRE(src && idx && dst, MPV_E_ARG, "mpv_gather_rows: null pointer");
  MPV_REQUIRE(rows > 0 && cols > 0 && cols % 4 == 0 && ld % 4 == 0, MPV_E_SHAPE, "mpv_gather_rows: bad shape");
  hipLaunchKernelGGL(gather_rows_kernel, dim3(ew_grid(rows * (cols / 4))), dim3(256), 0, stream, (const bf16*)src, idx, (bf16*)dst,
                     (long long)rows, (int)cols, (long long)ld);
  return mpv_check_launch("mpv_gather_rows");
}

extern "C" int mpv_scatter_rows(const void* src, const int64_t* idx, void* dst, int64_t rows, int64_t cols, int64_t ld,
                                hipStream_t stream) {
  MPV_REQUIRE(src && idx && dst, MPV_E_ARG, "mpv_scatter_rows: null pointer");
  MPV_REQUIRE(rows > 0 && cols > 0 && cols % 4 == 0 && ld % 4 == 0, MPV_E_SHAPE, "mpv_scatter_rows: bad shape");
  hipLaunchKernelGGL(scatter_rows_kernel, dim3(ew_grid(rows * (cols / 4))), dim3(256), 0, stream, (const bf16*)src, idx, (bf16*)dst,
                     (long long)rows, (int)cols, (long long)ld);
  return mpv_check_launch("mpv_scatter_rows");
}

extern "C" int mpv_gather_rows_ld(const void* src, const int64_t* idx, void* dst, int64_t rows, int64_t cols, int64_t lds,
                                  int64_t ldd, hipStream_t stream) {
  MPV_REQUIRE(src && idx && dst, MPV_E_ARG, "mpv_gather_rows_ld: null pointer");
  MPV_REQUIRE(rows > 0 && cols > 0 && cols % 8 == 0 && lds % 8 == 0 && ldd % 8 == 0, MPV_E_SHAPE, "mpv_gather_rows_ld: bad shape");
  MPV_REQUIRE((((uintptr_t)src | (uintptr_t)dst) & 15) == 0, MPV_E_ALIGN, "mpv_gather_rows_ld: buffers must be 16-byte aligned");
  hipLaunchKernelGGL(gather_rows_ld_kernel, dim3(ew_grid(rows * (cols / 8))), dim3(256), 0, stream, (const bf16*)src, idx, (bf16*)dst,
                     (long long)rows, (long long)cols, (long long)lds, (long long)ldd);
  return mpv_check_launch("mpv_gather_rows_ld");
}

extern "C" size_t mpv_logprob_topk_workspace_size(int64_t rows, int k) {
  return (size_t)rows * TOPK_CHUNKS * ((size_t)k * 8 + 8) + 256;
}

extern "C" int mpv_logprob_topk(const void* logits, const float* add, int64_t rows, int64_t vocab, int64_t ld, int k,
                                float* out_val, int64_t* out_idx, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  MPV_REQUIRE(logits && out_val && out_idx && workspace, MPV_E_ARG, "mpv_logprob_topk: null pointer");
  MPV_REQUIRE(rows > 0 && vocab > 0 && k > 0 && k <= 64 && k <= vocab, MPV_E_SHAPE, "mpv_logprob_topk: need 0 < k <= min(64, vocab)");
  MPV_REQUIRE(workspace_bytes >= mpv_logprob_topk_workspace_size(rows, k), MPV_E_ARG, "mpv_logprob_topk: workspace too small");
  const size_t nc = (size_t)rows * TOPK_CHUNKS;
  float* cval = (float*)workspace;
  int* cidx = (int*)(cval + nc * k);
  float* pmax = (float*)(cidx + nc * k);
  float* psum = pmax + nc;
  hipLaunchKernelGGL(logprob_topk_part_kernel, dim3(TOPK_CHUNKS, (unsigned)rows), dim3(256), 0, stream, (const bf16*)logits, k, cval,
                     cidx, pmax, psum, (int)vocab, (long long)ld);
  hipLaunchKernelGGL(logprob_topk_merge_kernel, dim3((unsigned)rows), dim3(256), 0, stream, (const float*)cval, (const int*)cidx,
                     (const float*)pmax, (const float*)psum, add, k, out_val, out_idx);
  return mpv_check_launch("mpv_logprob_topk");
}

extern "C" int mpv_video_resized_crop_normalize(const uint8_t* clip, int T, int H, int W, int crop_i, int crop_j, int crop_h, int crop_w,
                                               int out_h, int out_w, int mode, int flip, const float* mean3, const float* std3, void* out,
                                               int64_t c_stride, int64_t t_stride, hipStream_t stream) {
  MPV_REQUIRE(clip && out && mean3 && std3, MPV_E_ARG, "mpv_video_resized_crop_normalize: null pointer");
  MPV_REQUIRE(T > 0 && H > 0 && W > 0 && out_h > 0 && out_w > 0 && crop_h > 0 && crop_w > 0 && crop_i >= 0 && crop_j >= 0 &&
                  crop_i + crop_h <= H && crop_j + crop_w <= W,
              MPV_E_SHAPE, "mpv_video_resized_crop_normalize: crop (%d,%d,%d,%d) outside a %dx%d frame", crop_i, crop_j, crop_h, crop_w, H, W);
  MPV_REQUIRE(mode >= 0 && mode <= 2, MPV_E_ARG, "mpv_video_resized_crop_normalize: mode must be 0 (nearest), 1 (bilinear) or 2 (bicubic)");
  VideoTfArgs a = {};
  a.clip = clip; a.out = (bf16*)out;
  a.T = T; a.H = H; a.W = W; a.ci = crop_i; a.cj = crop_j; a.ch = crop_h; a.cw = crop_w; a.oh = out_h; a.ow = out_w;
  a.mode = mode; a.flip = flip; a.c_stride = c_stride; a.t_stride = t_stride;
  for (int c = 0; c < 3; ++c) { a.mean[c] = mean3[c]; a.stdv[c] = std3[c]; }
  hipLaunchKernelGGL(video_transform_kernel, dim3(ew_grid((long long)T * out_h * out_w)), dim3(256), 0, stream, a);
  return mpv_check_launch("mpv_video_resized_crop_normalize");
}

extern "C" int mpv_gpt_embed_bwd_full(const void* dh, void* dfull, int64_t rows, int H, float dropout_p, uint64_t seed, uint64_t offset,
                                     hipStream_t stream) {
  MPV_REQUIRE(dh && dfull, MPV_E_ARG, "mpv_gpt_embed_bwd_full: null pointer");
  MPV_REQUIRE(rows > 0 && H > 0 && H % 8 == 0, MPV_E_SHAPE, "mpv_gpt_embed_bwd_full: H must be a multiple of 8");
  MPV_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, MPV_E_ARG, "mpv_gpt_embed_bwd_full: bad dropout_p");
  hipLaunchKernelGGL(gpt_embed_bwd_full_kernel, dim3(ew_grid(rows * (H / 8))), dim3(256), 0, stream, (const bf16*)dh, (bf16*)dfull, (long long)rows, H,
                     1.0f / (1.0f - dropout_p), dropout_p > 0.f ? mpv_drop_threshold(dropout_p) : 0u, seed, offset);
  return mpv_check_launch("mpv_gpt_embed_bwd_full");
}

extern "C" int mpv_video_resized_crop_u8(const uint8_t* clip, int T, int H, int W, int crop_i, int crop_j, int crop_h, int crop_w, int out_h,
                                        int out_w, int mode, int flip, uint8_t* out, hipStream_t stream) {
  MPV_REQUIRE(clip && out, MPV_E_ARG, "mpv_video_resized_crop_u8: null pointer");
  MPV_REQUIRE(T > 0 && H > 0 && W > 0 && out_h > 0 && out_w > 0 && crop_h > 0 && crop_w > 0 && crop_i >= 0 && crop_j >= 0 &&
                  crop_i + crop_h <= H && crop_j + crop_w <= W,
              MPV_E_SHAPE, "mpv_video_resized_crop_u8: crop (%d,%d,%d,%d) outside a %dx%d frame", crop_i, crop_j, crop_h, crop_w, H, W);
  MPV_REQUIRE(mode >= 0 && mode <= 2, MPV_E_ARG, "mpv_video_resized_crop_u8: mode must be 0 (nearest), 1 (bilinear) or 2 (bicubic)");
  VideoTfArgs a = {};
  a.clip = clip; a.out_u8 = out;
  a.T = T; a.H = H; a.W = W; a.ci = crop_i; a.cj = crop_j; a.ch = crop_h; a.cw = crop_w; a.oh = out_h; a.ow = out_w;
  a.mode = mode; a.flip = flip;
  hipLaunchKernelGGL(video_transform_kernel, dim3(ew_grid((long long)T * out_h * out_w)), dim3(256), 0, stream, a);
  return mpv_check_launch("mpv_video_resized_crop_u8");
}

extern "C" int mpv_soft_target_ce(const float* sim, const int64_t* row_ids, const int64_t* col_ids, float scale, float* losses,
                                  void* dsim, float* dts, int64_t rows, int64_t cols, hipStream_t stream) {
  MPV_REQUIRE(sim && row_ids && col_ids && losses, MPV_E_ARG, "mpv_soft_target_ce: null pointer");
  MPV_REQUIRE(rows > 0 && cols > 0, MPV_E_SHAPE, "mpv_soft_target_ce: empty problem");
  hipLaunchKernelGGL(soft_ce_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, sim, row_ids, col_ids, scale, losses,
                     (bf16*)dsim, dts, (int)rows, (int)cols);
  return mpv_check_launch("mpv_soft_target_ce");
}
