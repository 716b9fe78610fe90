// glue.hip -- the small fused kernels that take framework tensor ops (cat / zeros / slice-assign / foreach-copy /
// float round trips) off the hot path of the step: each replaces a chain of 4-12 generic element-wise launches of the
// host framework by ONE launch.  All of them move a few KB; what they save is launches (1.2-1.9 us per kernel boundary).
#include "mpv_common.h"
#include "../../include/mpv.h"

namespace {

// ---- multi-segment bf16 copy --------------------------------------------------------------------------------
constexpr int SEG_MAX = 64;
struct SegArgs {
  const bf16* src[SEG_MAX];
  bf16* dst[SEG_MAX];
  int count[SEG_MAX];
};
// one workgroup per segment; 16-byte pieces when both ends are 16-byte aligned, element-wise otherwise
__global__ __launch_bounds__(256) void copy_segments_kernel(const SegArgs a) {
  const int s = blockIdx.x;
  const bf16* __restrict__ src = a.src[s];
  bf16* __restrict__ dst = a.dst[s];
  const int n = a.count[s];
  if ((((uintptr_t)src | (uintptr_t)dst) & 15) == 0) {
    const int n8 = n >> 3;
    for (int i = threadIdx.x; i < n8; i += 256) *(bf16x8*)(dst + i * 8) = *(const bf16x8*)(src + i * 8);
    for (int i = (n8 << 3) + threadIdx.x; i < n; i += 256) dst[i] = src[i];
  } else {
    for (int i = threadIdx.x; i < n; i += 256) dst[i] = src[i];
  }
}

// ---- composed temporal-projection backward: the two [D, D] / [D] finishing steps ----------------------------
// dWf[i][j] = bf16( float(P[i][j]) + dbc[i] * bp[j] )      (P = dWc Wp^T from the GEMM, rounded to bf16 there)
// dbp[j]    = bf16( sum_i Wf[i][j] * dbc[i] )               (= Wf^T d(bc))
// grid.x < nrank: rank-1 update, one thread per 8 consecutive columns of a row (16-byte accesses; D % 8 == 0);
// the remaining D / 64 workgroups: column sums, 64 columns x 16 row lanes each (1024 threads), eight loads in flight per lane
// (round 4: with 4 row lanes and a rolled loop a lane walked 192 rows one dependent round trip at a time -- 25.6 us for 1.2 MB).
constexpr int CF_THREADS = 1024, CF_RL = CF_THREADS / 64;
__device__ __forceinline__ void compose_finish_body(const bf16* __restrict__ P, const bf16* __restrict__ dbc,
                                                    const bf16* __restrict__ bp, const bf16* __restrict__ wf,
                                                    bf16* __restrict__ dwf, bf16* __restrict__ dbp, int D, int nrank) {
  __shared__ float red[CF_RL][64];
  if ((int)blockIdx.x < nrank) {
    const int d8 = D >> 3;
    const int g = blockIdx.x * CF_THREADS + threadIdx.x;
    if (g < D * d8) {
      const int i = g / d8, c = (g - i * d8) * 8;
      const float a = bf2f(dbc[i]);
      const f32x8 b = cvt8(*(const bf16x8*)(bp + c));
      f32x8 v = cvt8(*(const bf16x8*)(P + (long long)i * D + c));
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += a * b[e];
      *(bf16x8*)(dwf + (long long)i * D + c) = cvt8(v);
    }
    return;
  }
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int j = (blockIdx.x - nrank) * 64 + cl;
  float acc = 0.f;
  if (j < D) {
    int i = rl;
    for (; i + 7 * CF_RL < D; i += 8 * CF_RL) {
      float w[8], d[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        w[u] = bf2f(wf[(long long)(i + u * CF_RL) * D + j]);
        d[u] = bf2f(dbc[i + u * CF_RL]);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += w[u] * d[u];
    }
    for (; i < D; i += CF_RL) acc += bf2f(wf[(long long)i * D + j]) * bf2f(dbc[i]);
  }
  red[rl][cl] = acc;
  __syncthreads();
  if (rl == 0 && j < D) {
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < CF_RL; ++r) t += red[r][cl];
    dbp[j] = f2bf(t);
  }
}

__global__ __launch_bounds__(CF_THREADS) void compose_finish_kernel(const bf16* __restrict__ P, const bf16* __restrict__ dbc,
                                                                    const bf16* __restrict__ bp, const bf16* __restrict__ wf,
                                                                    bf16* __restrict__ dwf, bf16* __restrict__ dbp, int D, int nrank) {
  compose_finish_body(P, dbc, bp, wf, dwf, dbp, D, nrank);
}

// The same for every block of the tower in ONE launch (blockIdx.y = block; the pointers travel by value), and the forward's
// bc = Wf bp + bf of every block in one launch: with the [D, D] products batched too (gemm.hip: gemm_bf16_batched_kernel) the composed
// temporal projection costs the step 5 small launches instead of 60.
constexpr int COMPOSE_BATCH_MAX = 16;
struct ComposeBatch {
  const bf16* p0[COMPOSE_BATCH_MAX];   // finish: dWc Wp^T        bias: Wf
  const bf16* p1[COMPOSE_BATCH_MAX];   //         d(bc)                 bp
  const bf16* p2[COMPOSE_BATCH_MAX];   //         bp                    bf
  const bf16* p3[COMPOSE_BATCH_MAX];   //         Wf                    --
  bf16* o0[COMPOSE_BATCH_MAX];         //         dWf                   bc
  bf16* o1[COMPOSE_BATCH_MAX];         //         d(bp)                 --
};
__global__ __launch_bounds__(CF_THREADS) void compose_finish_batched_kernel(const ComposeBatch b, int D, int nrank) {
  const int i = blockIdx.y;
  compose_finish_body(b.p0[i], b.p1[i], b.p2[i], b.p3[i], b.o0[i], b.o1[i], D, nrank);
}
// bc[r] = bf16( sum_j Wf[r][j] bp[j] + bf[r] ): one wave per output row, 16-byte loads, fp32 accumulation (D % 8 == 0)
__global__ __launch_bounds__(256) void compose_bias_batched_kernel(const ComposeBatch b, int D) {
  const int blk = blockIdx.y;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= D) return;
  const bf16* __restrict__ w = b.p0[blk] + (long long)r * D;
  const bf16* __restrict__ v = b.p1[blk];
  float acc = 0.f;
  for (int c = lane * 8; c < D; c += 512) {
    const f32x8 a = cvt8(*(const bf16x8*)(w + c)), x = cvt8(*(const bf16x8*)(v + c));
#pragma unroll
    for (int e = 0; e < 8; ++e) acc += a[e] * x[e];
  }
  acc = wave_sum(acc);
  if (lane == 0) b.o0[blk][r] = f2bf(acc + bf2f(b.p2[blk][r]));
}

// ---- caption targets of the loss window ---------------------------------------------------------------------
// models/distributed_gpt3.py:142-159 on the L text positions behind the Q query slots (the only ones whose loss weight can
// be non-zero): label[b][l] = ids[b][l+1] (l < L-1), ids[b][1] (l = L-1); weight[b][l] = m[b][l+1] / sum(m[:,1:]) with
// m = attention_mask, zeroed where l < prompt_len[b] (:348-351); weight of the last position = 0 (losses[:, :-1], :1615-1617).
__global__ __launch_bounds__(256) void caption_targets_kernel(const long long* __restrict__ ids, const long long* __restrict__ mask,
                                                              const long long* __restrict__ plen, int B, int L,
                                                              long long* __restrict__ labels, float* __restrict__ weights) {
  __shared__ float red[4];
  const int n = B * L;
  float cnt = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) {
    const int b = i / L, l = i - b * L;
    if (l < L - 1 && !(plen && l < plen[b])) cnt += (float)mask[(long long)b * L + l + 1];
  }
  cnt = wave_sum(cnt);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = cnt;
  __syncthreads();
  const float denom = (red[0] + red[1]) + (red[2] + red[3]);
  for (int i = threadIdx.x; i < n; i += 256) {
    const int b = i / L, l = i - b * L;
    labels[i] = L > 1 ? ids[(long long)b * L + (l < L - 1 ? l + 1 : 1)] : 0;
    float w = 0.f;
    if (l < L - 1 && !(plen && l < plen[b])) w = (float)mask[(long long)b * L + l + 1] / denom;
    weights[i] = w;
  }
}

// ---- deferred LayerNorm parameter-gradient reduction --------------------------------------------------------
// mpv_layernorm_bwd in deferred mode leaves its per-workgroup partials [nblk][2][cols] (fp32) in a caller-owned buffer; the
// finish folds up to LNF_MAX such buffers in TWO launches (a ViT block has three LayerNorms: six reduce launches before):
// level 1, grid (cols / 64, LNF_SLICES, n): every workgroup (64 columns x 4 row lanes) folds its slice of the partial rows into
// row `slice` of the buffer's tail area [nblk + slice]; level 2, grid (cols / 64, 1, n): folds the LNF_SLICES rows and writes bf16.
// Fixed summation order (deterministic).
constexpr int LNF_MAX = 8;
constexpr int LNF_SLICES = 16;
struct LnFinishArgs {
  float* part[LNF_MAX];
  bf16* dgamma[LNF_MAX];
  bf16* dbeta[LNF_MAX];
  int nblk[LNF_MAX];
  int accumulate[LNF_MAX];
  int cols;
};
template <bool FINAL>
__global__ __launch_bounds__(256) void ln_dparam_finish_kernel(const LnFinishArgs a) {
  __shared__ float red[2][4][64];
  const int s = blockIdx.z;
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const int nblk = a.nblk[s], cols = a.cols;
  const float* __restrict__ part = FINAL ? a.part[s] + (long long)nblk * 2 * cols : a.part[s];
  const int per = (nblk + LNF_SLICES - 1) / LNF_SLICES;
  const int i0 = FINAL ? 0 : blockIdx.y * per, i1 = FINAL ? LNF_SLICES : min(nblk, i0 + per);
  float g0 = 0.f, b0 = 0.f, g1 = 0.f, b1 = 0.f;
  if (c < cols) {
    int i = i0 + rl;
    for (; i + 4 < i1; i += 8) {
      g0 += part[(long long)i * 2 * cols + c];
      b0 += part[(long long)i * 2 * cols + cols + c];
      g1 += part[(long long)(i + 4) * 2 * cols + c];
      b1 += part[(long long)(i + 4) * 2 * cols + cols + c];
    }
    if (i < i1) {
      g0 += part[(long long)i * 2 * cols + c];
      b0 += part[(long long)i * 2 * cols + cols + c];
    }
  }
  red[0][rl][cl] = g0 + g1;
  red[1][rl][cl] = b0 + b1;
  __syncthreads();
  if (rl == 0 && c < cols) {
    float g = (red[0][0][cl] + red[0][1][cl]) + (red[0][2][cl] + red[0][3][cl]);
    float b = (red[1][0][cl] + red[1][1][cl]) + (red[1][2][cl] + red[1][3][cl]);
    if constexpr (!FINAL) {
      float* out = a.part[s] + (long long)(nblk + blockIdx.y) * 2 * cols;
      out[c] = g;
      out[cols + c] = b;
    } else {
      if (a.accumulate[s]) {
        g += bf2f(a.dgamma[s][c]);
        b += bf2f(a.dbeta[s][c]);
      }
      a.dgamma[s][c] = f2bf(g);
      a.dbeta[s][c] = f2bf(b);
    }
  }
}

// ---- gradient-accumulation window in fp32 -------------------------------------------------------------------
__global__ __launch_bounds__(256) void accum_f32_kernel(float* __restrict__ acc, const bf16* __restrict__ g, long long n8, int first) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
    f32x8 v = cvt8(*(const bf16x8*)(g + i * 8));
    if (!first) v += *(const f32x8*)(acc + i * 8);
    *(f32x8*)(acc + i * 8) = v;
  }
}
__global__ __launch_bounds__(256) void f32_to_bf16_kernel(const float* __restrict__ src, bf16* __restrict__ dst, long long n8) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256)
    *(bf16x8*)(dst + i * 8) = cvt8(*(const f32x8*)(src + i * 8));
}
unsigned stream_grid(long long n8) { return (unsigned)((n8 + 255) / 256 < 8192 ? (n8 + 255) / 256 : 8192); }

// up to WORDS_MAX 32-bit words, passed BY VALUE in the kernel arguments, to device memory: how the host hands a step's
// scalars (dropout seeds, AdamW hyper-parameters) to kernels that read them from memory (mpv_store_words).  No host buffer
// has to outlive the call, unlike an asynchronous copy from pinned memory issued by a host that runs steps ahead of the device.
constexpr int WORDS_MAX = 32;
struct WordArgs {
  uint32_t w[WORDS_MAX];
};
__global__ void store_words_kernel(uint32_t* __restrict__ dst, WordArgs a, int n) {
  if ((int)threadIdx.x < n) dst[threadIdx.x] = a.w[threadIdx.x];
}

}  // namespace

extern "C" int mpv_accum_f32(float* acc, const void* g, int64_t n, int first, hipStream_t stream) {
  MPV_REQUIRE(acc && g, MPV_E_ARG, "mpv_accum_f32: null pointer");
  MPV_REQUIRE(n >= 0 && n % 8 == 0, MPV_E_SHAPE, "mpv_accum_f32: n must be a multiple of 8");
  MPV_REQUIRE((((uintptr_t)acc & 31) | ((uintptr_t)g & 15)) == 0, MPV_E_ALIGN, "mpv_accum_f32: buffers must be 32- / 16-byte aligned");
  if (n == 0) return MPV_OK;
  hipLaunchKernelGGL(accum_f32_kernel, dim3(stream_grid(n / 8)), dim3(256), 0, stream, acc, (const bf16*)g, (long long)(n / 8), first);
  return mpv_check_launch("mpv_accum_f32");
}

extern "C" int mpv_f32_to_bf16(const float* src, void* dst, int64_t n, hipStream_t stream) {
  MPV_REQUIRE(src && dst, MPV_E_ARG, "mpv_f32_to_bf16: null pointer");
  MPV_REQUIRE(n >= 0 && n % 8 == 0, MPV_E_SHAPE, "mpv_f32_to_bf16: n must be a multiple of 8");
  MPV_REQUIRE((((uintptr_t)src & 31) | ((uintptr_t)dst & 15)) == 0, MPV_E_ALIGN, "mpv_f32_to_bf16: buffers must be 32- / 16-byte aligned");
  if (n == 0) return MPV_OK;
  hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(stream_grid(n / 8)), dim3(256), 0, stream, src, (bf16*)dst, (long long)(n / 8));
  return mpv_check_launch("mpv_f32_to_bf16");
}

extern "C" int mpv_store_words(void* dst, const uint32_t* words, int n, hipStream_t stream) {
  MPV_REQUIRE(dst && words && n >= 1 && n <= WORDS_MAX && ((uintptr_t)dst & 3) == 0, MPV_E_ARG, "mpv_store_words: 1..%d words to a 4-byte aligned address", WORDS_MAX);
  WordArgs a = {};
  for (int i = 0; i < n; ++i) a.w[i] = words[i];
  hipLaunchKernelGGL(store_words_kernel, dim3(1), dim3(64), 0, stream, (uint32_t*)dst, a, n);
  return mpv_check_launch("mpv_store_words");
}

extern "C" int mpv_copy_segments(const void* const* src, void* const* dst, const int64_t* count, int n, hipStream_t stream) {
  MPV_REQUIRE(n >= 0 && (n == 0 || (src && dst && count)), MPV_E_ARG, "mpv_copy_segments: null pointer");
  for (int base = 0; base < n; base += SEG_MAX) {
    SegArgs a = {};
    const int m = n - base < SEG_MAX ? n - base : SEG_MAX;
    for (int i = 0; i < m; ++i) {
      MPV_REQUIRE(src[base + i] && dst[base + i] && count[base + i] >= 0 && count[base + i] < (1LL << 31), MPV_E_ARG,
                  "mpv_copy_segments: bad segment %d", base + i);
      a.src[i] = (const bf16*)src[base + i];
      a.dst[i] = (bf16*)dst[base + i];
      a.count[i] = (int)count[base + i];
    }
    hipLaunchKernelGGL(copy_segments_kernel, dim3((unsigned)m), dim3(256), 0, stream, a);
  }
  return mpv_check_launch("mpv_copy_segments");
}

extern "C" int mpv_vit_compose_bwd_finish(const void* dwc_wpT, const void* dbc, const void* bp, const void* wf, void* dwf, void* dbp,
                                          int D, hipStream_t stream) {
  MPV_REQUIRE(dwc_wpT && dbc && bp && wf && dwf && dbp, MPV_E_ARG, "mpv_vit_compose_bwd_finish: null pointer");
  MPV_REQUIRE(D > 0, MPV_E_SHAPE, "mpv_vit_compose_bwd_finish: empty problem");
  MPV_REQUIRE(D % 8 == 0 && (((uintptr_t)dwc_wpT | (uintptr_t)bp | (uintptr_t)dwf) & 15) == 0, MPV_E_ALIGN,
              "mpv_vit_compose_bwd_finish: D must be a multiple of 8 and the matrices 16-byte aligned");
  const int nrank = (int)(((long long)D * (D / 8) + CF_THREADS - 1) / CF_THREADS);
  hipLaunchKernelGGL(compose_finish_kernel, dim3((unsigned)(nrank + (D + 63) / 64)), dim3(CF_THREADS), 0, stream, (const bf16*)dwc_wpT, (const bf16*)dbc,
                     (const bf16*)bp, (const bf16*)wf, (bf16*)dwf, (bf16*)dbp, D, nrank);
  return mpv_check_launch("mpv_vit_compose_bwd_finish");
}

extern "C" int mpv_caption_targets(const int64_t* ids, const int64_t* attention_mask, const int64_t* prompt_len, int B, int L,
                                   int64_t* labels, float* weights, hipStream_t stream) {
  MPV_REQUIRE(ids && attention_mask && labels && weights, MPV_E_ARG, "mpv_caption_targets: null pointer");
  MPV_REQUIRE(B > 0 && L > 0 && (long long)B * L < (1LL << 30), MPV_E_SHAPE, "mpv_caption_targets: bad shape");
  hipLaunchKernelGGL(caption_targets_kernel, dim3(1), dim3(256), 0, stream, (const long long*)ids, (const long long*)attention_mask,
                     (const long long*)prompt_len, B, L, (long long*)labels, weights);
  return mpv_check_launch("mpv_caption_targets");
}

extern "C" int mpv_layernorm_dparam_finish(const float* const* partials, const int* nblk, void* const* dgamma, void* const* dbeta,
                                           const int* accumulate, int n, int64_t cols, hipStream_t stream) {
  MPV_REQUIRE(n >= 0 && (n == 0 || (partials && nblk && dgamma && dbeta && accumulate)), MPV_E_ARG, "mpv_layernorm_dparam_finish: null pointer");
  MPV_REQUIRE(cols > 0, MPV_E_SHAPE, "mpv_layernorm_dparam_finish: cols must be positive");
  for (int base = 0; base < n; base += LNF_MAX) {
    LnFinishArgs a = {};
    const int m = n - base < LNF_MAX ? n - base : LNF_MAX;
    for (int i = 0; i < m; ++i) {
      MPV_REQUIRE(partials[base + i] && dgamma[base + i] && dbeta[base + i] && nblk[base + i] > 0, MPV_E_ARG,
                  "mpv_layernorm_dparam_finish: bad entry %d", base + i);
      for (int j = 0; j < i; ++j)
        MPV_REQUIRE(a.dgamma[j] != (bf16*)dgamma[base + i], MPV_E_ARG, "mpv_layernorm_dparam_finish: two entries of one launch share a dgamma");
      a.part[i] = const_cast<float*>(partials[base + i]);
      a.nblk[i] = nblk[base + i];
      a.dgamma[i] = (bf16*)dgamma[base + i];
      a.dbeta[i] = (bf16*)dbeta[base + i];
      a.accumulate[i] = accumulate[base + i];
    }
    a.cols = (int)cols;
    const unsigned gx = (unsigned)((cols + 63) / 64);
    hipLaunchKernelGGL(ln_dparam_finish_kernel<false>, dim3(gx, LNF_SLICES, (unsigned)m), dim3(256), 0, stream, a);
    hipLaunchKernelGGL(ln_dparam_finish_kernel<true>, dim3(gx, 1, (unsigned)m), dim3(256), 0, stream, a);
  }
  return mpv_check_launch("mpv_layernorm_dparam_finish");
}

extern "C" int mpv_vit_compose_bwd_finish_batched(const void* const* dwc_wpT, const void* const* dbc, const void* const* bp, const void* const* wf,
                                                  void* const* dwf, void* const* dbp, int batch, int D, hipStream_t stream) {
  MPV_REQUIRE(dwc_wpT && dbc && bp && wf && dwf && dbp && batch > 0, MPV_E_ARG, "mpv_vit_compose_bwd_finish_batched: null pointer table or empty batch");
  MPV_REQUIRE(D > 0 && D % 8 == 0, MPV_E_SHAPE, "mpv_vit_compose_bwd_finish_batched: D must be a positive multiple of 8");
  for (int i = 0; i < batch; ++i) {
    MPV_REQUIRE(dwc_wpT[i] && dbc[i] && bp[i] && wf[i] && dwf[i] && dbp[i], MPV_E_ARG, "mpv_vit_compose_bwd_finish_batched: null pointer in block %d", i);
    MPV_REQUIRE((((uintptr_t)dwc_wpT[i] | (uintptr_t)bp[i] | (uintptr_t)dwf[i]) & 15) == 0, MPV_E_ALIGN,
                "mpv_vit_compose_bwd_finish_batched: the matrices must be 16-byte aligned (block %d)", i);
  }
  const int nrank = (int)(((long long)D * (D / 8) + CF_THREADS - 1) / CF_THREADS);
  for (int b0 = 0; b0 < batch; b0 += COMPOSE_BATCH_MAX) {
    const int nb = batch - b0 < COMPOSE_BATCH_MAX ? batch - b0 : COMPOSE_BATCH_MAX;
    ComposeBatch cb = {};
    for (int i = 0; i < nb; ++i) {
      cb.p0[i] = (const bf16*)dwc_wpT[b0 + i]; cb.p1[i] = (const bf16*)dbc[b0 + i]; cb.p2[i] = (const bf16*)bp[b0 + i];
      cb.p3[i] = (const bf16*)wf[b0 + i]; cb.o0[i] = (bf16*)dwf[b0 + i]; cb.o1[i] = (bf16*)dbp[b0 + i];
    }
    hipLaunchKernelGGL(compose_finish_batched_kernel, dim3((unsigned)(nrank + (D + 63) / 64), (unsigned)nb), dim3(CF_THREADS), 0, stream, cb, D, nrank);
  }
  return mpv_check_launch("mpv_vit_compose_bwd_finish_batched");
}

extern "C" int mpv_vit_compose_bias_batched(const void* const* wf, const void* const* bp, const void* const* bf, void* const* bc, int batch, int D,
                                            hipStream_t stream) {
  MPV_REQUIRE(wf && bp && bf && bc && batch > 0, MPV_E_ARG, "mpv_vit_compose_bias_batched: null pointer table or empty batch");
  MPV_REQUIRE(D > 0 && D % 8 == 0, MPV_E_SHAPE, "mpv_vit_compose_bias_batched: D must be a positive multiple of 8");
  for (int i = 0; i < batch; ++i) {
    MPV_REQUIRE(wf[i] && bp[i] && bf[i] && bc[i], MPV_E_ARG, "mpv_vit_compose_bias_batched: null pointer in block %d", i);
    MPV_REQUIRE((((uintptr_t)wf[i] | (uintptr_t)bp[i]) & 15) == 0, MPV_E_ALIGN, "mpv_vit_compose_bias_batched: Wf and bp must be 16-byte aligned (block %d)", i);
  }
  for (int b0 = 0; b0 < batch; b0 += COMPOSE_BATCH_MAX) {
    const int nb = batch - b0 < COMPOSE_BATCH_MAX ? batch - b0 : COMPOSE_BATCH_MAX;
    ComposeBatch cb = {};
    for (int i = 0; i < nb; ++i) {
      cb.p0[i] = (const bf16*)wf[b0 + i]; cb.p1[i] = (const bf16*)bp[b0 + i]; cb.p2[i] = (const bf16*)bf[b0 + i]; cb.o0[i] = (bf16*)bc[b0 + i];
    }
    hipLaunchKernelGGL(compose_bias_batched_kernel, dim3((unsigned)((D + 3) / 4), (unsigned)nb), dim3(256), 0, stream, cb, D);
  }
  return mpv_check_launch("mpv_vit_compose_bias_batched");
}
