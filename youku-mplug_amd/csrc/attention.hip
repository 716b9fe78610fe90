// attention.hip -- fused attention for gfx950 (CDNA4): scores are never written to HBM.
//
// One kernel family serves the four shape classes of the path:
//   GPT causal self-attention   (Sq=Sk<=208, head_dim 64/80, prob-dropout)
//   ViT spatial attention       (Sq=Sk=197,  head_dim 96, q pre-scaled in bf16)
//   AttentionPool cross-attn    (Sq=128, Sk=1+T*196+1, head_dim 96)
// plus a VALU kernel pair for the divided space-time *temporal* attention (T<=16 tokens,
// hundreds of thousands of tiny problems, HBM-bound).
//
// Layout of the MFMA kernels: a workgroup = 4 waves = 128 query rows (forward, dQ) or 128
// key rows (dK/dV) of one (batch, head); every wave owns a 32-row tile whose operand
// fragments stay in registers; the other side streams through LDS in 64-row chunks
// (register-staged, double-buffered, one barrier per chunk).  QK^T is issued "swapped"
// (S^T = K Q^T) so a lane owns one query row of the 32x32 score tile: the softmax is
// lane-local plus one cross-half shuffle, and the probabilities feed the PV MFMA as its
// B operand straight from registers (the reduction index of an MFMA may be permuted freely
// as long as both operands agree; V / K^T / Q^T / dO^T fragments are fetched with the gfx950
// transposing LDS read ds_read_b64_tr_b16 using the matching permutation).
// Softmax is exact two-pass (pass 1: row max / sum, pass 2: normalised P.V), fp32, which
// reproduces the reference's "softmax then cast to bf16 then @V" numerics and leaves a
// per-row log-sum-exp for the backward kernels.
#include <string.h>
#include <type_traits>

#include "mpv_common.h"
#include "../../include/mpv.h"

namespace {

constexpr int CH = 64;          // rows per LDS chunk
constexpr int ROWB = 208;       // LDS row pitch in bytes (96 bf16 + 16 pad: conflict-free ds_read_b128)
constexpr int CHUNK_BYTES = CH * ROWB;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
// a wave's DS ops retire in order; wait for them and stop the compiler from moving LDS traffic across
#define WAVE_SYNC()                                  \
  do {                                               \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
    __builtin_amdgcn_s_waitcnt(0xc07f);              \
    __builtin_amdgcn_wave_barrier();                 \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
  } while (0)

// v_exp_f32 directly: exp2f() wraps it in a denormal-range rescue (6 VALU instead of 1; measured 38% of the forward
// kernel's VALU instructions); probabilities below 2^-126 may flush to zero, far below bf16 resolution of the result
__device__ __forceinline__ float fexp2(float x) { return __builtin_amdgcn_exp2f(x); }
// attention-probability dropout: element (row, key) of the score matrix, pairs along the key axis (mpv_common.h)
__device__ __forceinline__ bool attn_keep(uint64_t seed, uint64_t row_base, int key, uint32_t thr) {
  return mpv_keep(seed, row_base + (uint64_t)(key & ~1), key & 1, thr);
}

#ifdef MPV_ATTN_TIMING   // measurement build only (tools/probe/attn_timeline.py)
__device__ long long mpv_attn_dbg[4096 * 8];
#define ASTAMP(i) do { if (threadIdx.x == 0 && blockIdx.y < 4096) mpv_attn_dbg[blockIdx.y * 8 + (i)] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define ASTAMP(i) do { } while (0)
#endif

struct AttnArgs {
  const bf16 *q, *k, *v;
  bf16* o;
  float* lse;
  long long q_bs, q_hs, q_rs, k_bs, k_hs, k_rs, v_bs, v_hs, v_rs, o_bs, o_hs, o_rs;
  int batch, heads, sq, sk;
  int hd;       // true head_dim (<= HD, multiple of 8); columns hd..HD-1 are zero-filled on load and never stored
  int causal;
  float scale;
  int scale_q_bf16;
  float drop_scale;
  uint32_t drop_thr;
  uint64_t seed, offset;
  // backward
  const bf16* dO;
  bf16 *dq, *dk, *dv;
  float* delta;
};

// ---- chunk staging: 64 rows x HD columns of a [rows][row_stride] bf16 matrix -> LDS -------
template <int HD>
struct ChunkStage {
  static constexpr int NDT = (HD + 31) / 32;
  static constexpr int CPR = NDT * 4;               // 16-byte chunks per (padded) row
  static constexpr int NLOAD = (CH * CPR + 255) / 256;
  i32x4 r[NLOAD];

  // nthr = blockDim.x (256..512): with more threads the later iterations are simply predicated off
  __device__ __forceinline__ void issue(__amdgpu_buffer_rsrc_t src, int tid, int row0, int nrows, long long rs, int hd = HD) {
    const int nthr = blockDim.x;
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) {
      const int c = tid + nthr * i;
      const int row = c / CPR, cc = c - row * CPR;
      const bool ok = (c < CH * CPR) && (row0 + row < nrows) && (cc * 8 < hd);
      const uint32_t off = ok ? (uint32_t)(((long long)(row0 + row) * rs + cc * 8) * 2) : 0x80000000u;
      r[i] = __builtin_amdgcn_raw_buffer_load_b128(src, off, 0, 0);
    }
  }
  // optional in-register transform q' = bf16(q * scale)
  __device__ __forceinline__ void scale_bf16(float s) {
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) {
      union { i32x4 i; bf16x8 b; } u;
      u.i = r[i];
      f32x8 f = cvt8(u.b);
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] *= s;
      u.b = cvt8(f);
      r[i] = u.i;
    }
  }
  __device__ __forceinline__ void commit(char* lds, int tid) {
    const int nthr = blockDim.x;
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) {
      const int c = tid + nthr * i;
      const int row = c / CPR, cc = c - row * CPR;
      if (c < CH * CPR) *(i32x4*)(lds + row * ROWB + cc * 16) = r[i];
    }
  }
};

// fragment with 8 consecutive d for row (lane&31) of 32-row tile `t`, d-step s (ds_read_b128)
template <int PITCH = ROWB>
__device__ __forceinline__ bf16x8 frag_rows(const char* lds, int t, int s, int lane) {
  return *(const bf16x8*)(lds + (t * 32 + (lane & 31)) * PITCH + (s * 16 + (lane >> 5) * 8) * 2);
}
// transposed fragment: column d = dt*32 + (lane&31); 8 rows (t*32 + ks*16 + {0..3}+4h, +8) (tr read)
template <int PITCH = ROWB>
__device__ __forceinline__ bf16x8 frag_cols(const char* lds, int t, int ks, int dt, int lane) {
  const int h = lane >> 5;
  const int rbase = t * 32 + ks * 16 + 4 * h + ((lane & 15) >> 2);
  const int col = dt * 32 + ((lane >> 4) & 1) * 16 + (lane & 3) * 4;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + rbase * PITCH + col * 2));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + (rbase + 8) * PITCH + col * 2));
  union { struct { s16x4 a, b; } s; bf16x8 v; } u;
  u.s.a = lo;
  u.s.b = hi;
  return u.v;
}
// clamped variants (resident kernels): rows past `last` re-read row `last` (finite data; the results of
// such rows are always masked), so unpadded LDS regions are never over-read
template <int PITCH>
__device__ __forceinline__ bf16x8 frag_rows_c(const char* lds, int t, int s, int lane, int last) {
  return *(const bf16x8*)(lds + min(t * 32 + (lane & 31), last) * PITCH + (s * 16 + (lane >> 5) * 8) * 2);
}
template <int PITCH>
__device__ __forceinline__ bf16x8 frag_cols_c(const char* lds, int t, int ks, int dt, int lane, int last) {
  const int h = lane >> 5;
  const int rbase = t * 32 + ks * 16 + 4 * h + ((lane & 15) >> 2);
  const int col = dt * 32 + ((lane >> 4) & 1) * 16 + (lane & 3) * 4;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + min(rbase, last) * PITCH + col * 2));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + min(rbase + 8, last) * PITCH + col * 2));
  union { struct { s16x4 a, b; } s; bf16x8 v; } u;
  u.s.a = lo;
  u.s.b = hi;
  return u.v;
}

// ---- fragment reads of the LDS-resident kernels on plain 32-bit LDS offsets: one per-lane base register per operand,
// tile / step / d-tile strides are immediates or a scalar.  (The clamped readers above recompute a 64-bit address with a
// min() per fragment: measured ~100 of the ~300 VALU instructions per q-tile of the dK/dV kernel, which is VALU-bound.)
// Regions hold whole 32-row tiles, the rows past the sequence end zero-filled by the LDS-DMA (out-of-range source), so
// a tile read never leaves defined data; every product with such a row is masked by a select or multiplied by an exact
// zero probability.
typedef __attribute__((address_space(3))) char lds_char;
typedef __attribute__((address_space(3))) bf16x8 lds_bf16x8;
template <int PITCH>
__device__ __forceinline__ int rows_lane_base(int lane) { return (lane & 31) * PITCH + (lane >> 5) * 16; }
template <int PITCH>
__device__ __forceinline__ int cols_lane_base(int lane) {
  return (4 * (lane >> 5) + ((lane & 15) >> 2)) * PITCH + (((lane >> 4) & 1) * 16 + (lane & 3) * 4) * 2;
}
// rows fragment: d-step s; `off` = region offset + tile * 32 * PITCH + lane base
__device__ __forceinline__ bf16x8 frag_rows_f(const lds_char* sm, int off, int s) { return *(const lds_bf16x8*)(sm + off + s * 32); }
template <int PITCH>
__device__ __forceinline__ bf16x8 frag_cols_f(const lds_char* sm, int off, int ks, int dt) {
  const lds_char* a = sm + off + ks * 16 * PITCH + dt * 64;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a + 8 * PITCH));
  union { struct { s16x4 a, b; } s; bf16x8 v; } u;
  u.s.a = lo;
  u.s.b = hi;
  return u.v;
}

// registers of a 32x32 accumulator that form the B-operand for reduction step ks (see header)
__device__ __forceinline__ bf16x8 acc_to_frag(const f32x16& a, int ks) {
  f32x8 f;
#pragma unroll
  for (int e = 0; e < 8; ++e) f[e] = a[8 * ks + e];
  return cvt8(f);
}
// row index inside a 32-row tile held by accumulator register `reg` of this lane
__device__ __forceinline__ int acc_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

// load this lane's B-operand fragments (row = row0 + lane&31, all d) straight from global
template <int HD>
__device__ __forceinline__ void load_row_frags(bf16x8 (&f)[HD / 16], const bf16* base, long long rs, int row, int nrows,
                                               int lane, int hd = HD) {
  // Branch-free (round 4): a row past the end re-reads the last row and is zeroed by a select.  Behind `if (row < nrows) load` the
  // compiler cannot count on the request having been issued, so every batch of fragments was waited for with vmcnt(0) before the next
  // operand's batch went out -- Q, dO, the statistics and the O rows of a dQ block were four round trips in a row.
  const bf16* rp = base + (long long)min(row, nrows - 1) * rs + (lane >> 5) * 8;
#pragma unroll
  for (int s = 0; s < HD / 16; ++s) {
    union { i32x4 i; bf16x8 b; } u;
    const bool cok = s * 16 + (lane >> 5) * 8 < hd;
    u.i = *(const i32x4*)(rp + (cok ? s * 16 : 0));
    if (!(row < nrows && cok)) u.i = i32x4{0, 0, 0, 0};
    f[s] = u.b;
  }
}
template <int N>
__device__ __forceinline__ void scale_frags_bf16(bf16x8 (&f)[N], float s) {
#pragma unroll
  for (int i = 0; i < N; ++i) {
    f32x8 x = cvt8(f[i]);
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] *= s;
    f[i] = cvt8(x);
  }
}

__device__ __forceinline__ int last_visible_key(const AttnArgs& p, int qrow) {
  return p.causal ? min(p.sk - 1, qrow + (p.sk - p.sq)) : p.sk - 1;
}

// =========================================================================== forward
template <int HD>
__global__ __launch_bounds__(512) void attn_fwd_kernel(const AttnArgs p) {
  const uint64_t seed_r = p.drop_thr ? mpv_resolve_seed(p.seed) : 0;      // (bit 63 set: the seed lives in device memory, mpv_common.h)
  constexpr int NS = HD / 16;
  constexpr int NDT = (HD + 31) / 32;
  __shared__ __attribute__((aligned(16))) char smem[4 * CHUNK_BYTES];  // [buf][K|V]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bh = blockIdx.y, b = bh / p.heads, h = bh - b * p.heads;
  const int rows_per_blk = (blockDim.x >> 6) * 32;
  const int q0 = blockIdx.x * rows_per_blk + wave * 32;
  const int qrow = q0 + (lane & 31);
  const bf16* qb = p.q + b * p.q_bs + h * p.q_hs;
  const bf16* kb = p.k + b * p.k_bs + h * p.k_hs;
  const bf16* vb = p.v + b * p.v_bs + h * p.v_hs;
  const __amdgpu_buffer_rsrc_t ksrc = make_rsrc(kb, (uint32_t)(((long long)(p.sk - 1) * p.k_rs + p.hd) * 2));
  const __amdgpu_buffer_rsrc_t vsrc = make_rsrc(vb, (uint32_t)(((long long)(p.sk - 1) * p.v_rs + p.hd) * 2));

  bf16x8 qf[NS];
  load_row_frags<HD>(qf, qb, p.q_rs, qrow, p.sq, lane, p.hd);
  float sc = p.scale;
  if (p.scale_q_bf16) {
    if (p.scale_q_bf16 == 1) scale_frags_bf16(qf, p.scale);      // (2: q arrives scaled, mpv.h)
    sc = 1.0f;
  }
  // keys needed by this workgroup / by this wave (wave-uniform: tiles beyond it are skipped)
  const int blk_last_q = min(p.sq - 1, blockIdx.x * rows_per_blk + rows_per_blk - 1);
  const int kmax = last_visible_key(p, blk_last_q);
  const int nchunk = kmax / CH + 1;
  const int my_last = last_visible_key(p, qrow);
  const int wave_last = q0 < p.sq ? last_visible_key(p, min(p.sq - 1, q0 + 31)) : -1;

  ChunkStage<HD> sk_, sv_;
  float m = -INFINITY, l = 0.f;

  auto scores = [&](const char* kl, int kt, f32x16& s) {
#pragma unroll
    for (int e = 0; e < 16; ++e) s[e] = 0.f;
#pragma unroll
    for (int st = 0; st < NS; ++st) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(kl, kt, st, lane), qf[st], s, 0, 0, 0);
  };

  // ---------------- pass 1: row max and sum
  sk_.issue(ksrc, tid, 0, p.sk, p.k_rs, p.hd);
  sk_.commit(smem, tid);
  __syncthreads();
  for (int c = 0; c < nchunk; ++c) {
    const int cur = c & 1;
    if (c + 1 < nchunk) sk_.issue(ksrc, tid, (c + 1) * CH, p.sk, p.k_rs, p.hd);
    const char* kl = smem + cur * 2 * CHUNK_BYTES;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      if (c * CH + kt * 32 > wave_last) continue;
      f32x16 s;
      scores(kl, kt, s);
      float mx = -INFINITY;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int key = c * CH + kt * 32 + acc_row(e, lane);
        s[e] = key <= my_last ? s[e] * sc : -INFINITY;
        mx = fmaxf(mx, s[e]);
      }
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float mn = fmaxf(m, mx);
      if (mn > -INFINITY) {
        float sum = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) sum += __expf(s[e] - mn);
        sum += __shfl_xor(sum, 32, 64);
        l = l * __expf(m - mn) + sum;
        m = mn;
      }
    }
    if (c + 1 < nchunk) sk_.commit(smem + (cur ^ 1) * 2 * CHUNK_BYTES, tid);
    __syncthreads();
  }
  const float inv_l = l > 0.f ? 1.0f / l : 0.f;
  if (qrow < p.sq && lane < 32 && p.lse) p.lse[(long long)bh * p.sq + qrow] = m + __logf(l);

  // ---------------- pass 2: O = softmax(S) V
  f32x16 oacc[NDT];
#pragma unroll
  for (int d = 0; d < NDT; ++d)
#pragma unroll
    for (int e = 0; e < 16; ++e) oacc[d][e] = 0.f;
  sk_.issue(ksrc, tid, 0, p.sk, p.k_rs, p.hd);
  sv_.issue(vsrc, tid, 0, p.sk, p.v_rs, p.hd);
  sk_.commit(smem, tid);
  sv_.commit(smem + CHUNK_BYTES, tid);
  __syncthreads();
  for (int c = 0; c < nchunk; ++c) {
    const int cur = c & 1;
    if (c + 1 < nchunk) {
      sk_.issue(ksrc, tid, (c + 1) * CH, p.sk, p.k_rs, p.hd);
      sv_.issue(vsrc, tid, (c + 1) * CH, p.sk, p.v_rs, p.hd);
    }
    const char* kl = smem + cur * 2 * CHUNK_BYTES;
    const char* vl = kl + CHUNK_BYTES;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      if (c * CH + kt * 32 > wave_last) continue;
      f32x16 s;
      scores(kl, kt, s);
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int key = c * CH + kt * 32 + acc_row(e, lane);
        float pr = key <= my_last ? __expf(s[e] * sc - m) * inv_l : 0.f;
        if (p.drop_thr) {
          const uint64_t rb = p.offset + ((uint64_t)bh * p.sq + (uint64_t)qrow) * (uint64_t)p.sk;
          pr = attn_keep(seed_r, rb, key, p.drop_thr) ? pr * p.drop_scale : 0.f;
        }
        s[e] = pr;
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const bf16x8 pf = acc_to_frag(s, ks);
#pragma unroll
        for (int d = 0; d < NDT; ++d)
          oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols(vl, kt, ks, d, lane), pf, oacc[d], 0, 0, 0);
      }
    }
    if (c + 1 < nchunk) {
      sk_.commit(smem + (cur ^ 1) * 2 * CHUNK_BYTES, tid);
      sv_.commit(smem + (cur ^ 1) * 2 * CHUNK_BYTES + CHUNK_BYTES, tid);
    }
    __syncthreads();
  }
  if (qrow < p.sq) {
    bf16* orow = p.o + b * p.o_bs + h * p.o_hs + (long long)qrow * p.o_rs;
#pragma unroll
    for (int d = 0; d < NDT; ++d)
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const int col = d * 32 + 8 * q4 + 4 * (lane >> 5);
        if (col < p.hd) {
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = oacc[d][4 * q4 + e];
          *(bf16x4*)(orow + col) = cvt4(v);
        }
      }
  }
}

// =========================================================================== delta = rowsum(dO * O)
// delta_i = sum_d dO[i][d] * O[i][d] from the row fragments a dQ wave already holds for dO (and loads once for O):
// lanes l and l^32 own the two halves of row (l & 31).  The dQ kernel also stores it for the dK/dV kernel that follows.
template <int HD>
__device__ __forceinline__ float row_delta(const bf16x8 (&dof)[HD / 16], const bf16* obase, long long rs, int row, int nrows, int lane,
                                           int hd) {
  bf16x8 of[HD / 16];
  load_row_frags<HD>(of, obase, rs, row, nrows, lane, hd);
  float s = 0.f;
#pragma unroll
  for (int st = 0; st < HD / 16; ++st) {
    const f32x8 a = cvt8(dof[st]), b = cvt8(of[st]);
#pragma unroll
    for (int e = 0; e < 8; ++e) s += a[e] * b[e];
  }
  return s + __shfl_xor(s, 32, 64);
}
// ... from O fragments the caller requested together with Q and dO (one round trip for the whole block instead of four)
template <int N>
__device__ __forceinline__ float row_delta_from(const bf16x8 (&dof)[N], const bf16x8 (&of)[N]) {
  float s = 0.f;
#pragma unroll
  for (int st = 0; st < N; ++st) {
    const f32x8 a = cvt8(dof[st]), b = cvt8(of[st]);
#pragma unroll
    for (int e = 0; e < 8; ++e) s += a[e] * b[e];
  }
  return s + __shfl_xor(s, 32, 64);
}

// =========================================================================== backward: dQ
template <int HD>
__global__ __launch_bounds__(512) void attn_bwd_dq_kernel(const AttnArgs p) {
  const uint64_t seed_r = p.drop_thr ? mpv_resolve_seed(p.seed) : 0;      // (bit 63 set: the seed lives in device memory, mpv_common.h)
  constexpr int NS = HD / 16;
  constexpr int NDT = (HD + 31) / 32;
  __shared__ __attribute__((aligned(16))) char smem[4 * CHUNK_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bh = blockIdx.y, b = bh / p.heads, h = bh - b * p.heads;
  const int rows_per_blk = (blockDim.x >> 6) * 32;
  const int q0 = blockIdx.x * rows_per_blk + wave * 32;
  const int qrow = q0 + (lane & 31);
  const bf16* qb = p.q + b * p.q_bs + h * p.q_hs;
  const bf16* kb = p.k + b * p.k_bs + h * p.k_hs;
  const bf16* vb = p.v + b * p.v_bs + h * p.v_hs;
  const bf16* dob = p.dO + b * p.o_bs + h * p.o_hs;
  const __amdgpu_buffer_rsrc_t ksrc = make_rsrc(kb, (uint32_t)(((long long)(p.sk - 1) * p.k_rs + p.hd) * 2));
  const __amdgpu_buffer_rsrc_t vsrc = make_rsrc(vb, (uint32_t)(((long long)(p.sk - 1) * p.v_rs + p.hd) * 2));

  bf16x8 qf[NS], dof[NS];
  load_row_frags<HD>(qf, qb, p.q_rs, qrow, p.sq, lane, p.hd);
  load_row_frags<HD>(dof, dob, p.o_rs, qrow, p.sq, lane, p.hd);
  float sc = p.scale;
  if (p.scale_q_bf16) {
    if (p.scale_q_bf16 == 1) scale_frags_bf16(qf, p.scale);      // (2: q arrives scaled, mpv.h)
    sc = 1.0f;
  }
  const bool qok = qrow < p.sq;
  const float lse = qok ? p.lse[(long long)bh * p.sq + qrow] : INFINITY;
  const float dl = row_delta<HD>(dof, p.o + b * p.o_bs + h * p.o_hs, p.o_rs, qrow, p.sq, lane, p.hd);
  if (qok && lane < 32) p.delta[(long long)bh * p.sq + qrow] = dl;
  const int blk_last_q = min(p.sq - 1, blockIdx.x * rows_per_blk + rows_per_blk - 1);
  const int nchunk = last_visible_key(p, blk_last_q) / CH + 1;
  const int my_last = last_visible_key(p, qrow);
  const int wave_last = q0 < p.sq ? last_visible_key(p, min(p.sq - 1, q0 + 31)) : -1;

  f32x16 dqacc[NDT];
#pragma unroll
  for (int d = 0; d < NDT; ++d)
#pragma unroll
    for (int e = 0; e < 16; ++e) dqacc[d][e] = 0.f;

  ChunkStage<HD> sk_, sv_;
  sk_.issue(ksrc, tid, 0, p.sk, p.k_rs, p.hd);
  sv_.issue(vsrc, tid, 0, p.sk, p.v_rs, p.hd);
  sk_.commit(smem, tid);
  sv_.commit(smem + CHUNK_BYTES, tid);
  __syncthreads();
  for (int c = 0; c < nchunk; ++c) {
    const int cur = c & 1;
    if (c + 1 < nchunk) {
      sk_.issue(ksrc, tid, (c + 1) * CH, p.sk, p.k_rs, p.hd);
      sv_.issue(vsrc, tid, (c + 1) * CH, p.sk, p.v_rs, p.hd);
    }
    const char* kl = smem + cur * 2 * CHUNK_BYTES;
    const char* vl = kl + CHUNK_BYTES;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      if (c * CH + kt * 32 > wave_last) continue;
      f32x16 s, dp;
#pragma unroll
      for (int e = 0; e < 16; ++e) s[e] = dp[e] = 0.f;
#pragma unroll
      for (int st = 0; st < NS; ++st) {
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(kl, kt, st, lane), qf[st], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(vl, kt, st, lane), dof[st], dp, 0, 0, 0);
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int key = c * CH + kt * 32 + acc_row(e, lane);
        const float pr = key <= my_last ? __expf(s[e] * sc - lse) : 0.f;
        float dpe = dp[e];
        if (p.drop_thr) {
          const uint64_t rb = p.offset + ((uint64_t)bh * p.sq + (uint64_t)qrow) * (uint64_t)p.sk;
          dpe = attn_keep(seed_r, rb, key, p.drop_thr) ? dpe * p.drop_scale : 0.f;
        }
        s[e] = pr * (dpe - dl);
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const bf16x8 dsf = acc_to_frag(s, ks);
#pragma unroll
        for (int d = 0; d < NDT; ++d)
          dqacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols(kl, kt, ks, d, lane), dsf, dqacc[d], 0, 0, 0);
      }
    }
    if (c + 1 < nchunk) {
      sk_.commit(smem + (cur ^ 1) * 2 * CHUNK_BYTES, tid);
      sv_.commit(smem + (cur ^ 1) * 2 * CHUNK_BYTES + CHUNK_BYTES, tid);
    }
    __syncthreads();
  }
  if (qok) {
    bf16* row = p.dq + b * p.q_bs + h * p.q_hs + (long long)qrow * p.q_rs;
#pragma unroll
    for (int d = 0; d < NDT; ++d)
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const int col = d * 32 + 8 * q4 + 4 * (lane >> 5);
        if (col < p.hd) {
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = dqacc[d][4 * q4 + e] * p.scale;
          *(bf16x4*)(row + col) = cvt4(v);
        }
      }
  }
}

// =========================================================================== backward: dK, dV
template <int HD>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(const AttnArgs p) {
  const uint64_t seed_r = p.drop_thr ? mpv_resolve_seed(p.seed) : 0;      // (bit 63 set: the seed lives in device memory, mpv_common.h)
  constexpr int NS = HD / 16;
  constexpr int NDT = (HD + 31) / 32;
  // [buf][Q|dO] chunks + [buf][lse|delta] rows
  __shared__ __attribute__((aligned(16))) char smem[4 * CHUNK_BYTES + 2 * 2 * CH * 4];
  float* stat = (float*)(smem + 4 * CHUNK_BYTES);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bh = blockIdx.y, b = bh / p.heads, h = bh - b * p.heads;
  const int rows_per_blk = (blockDim.x >> 6) * 32;
  const int k0 = blockIdx.x * rows_per_blk + wave * 32;
  const int krow = k0 + (lane & 31);
  const bf16* qb = p.q + b * p.q_bs + h * p.q_hs;
  const bf16* kb = p.k + b * p.k_bs + h * p.k_hs;
  const bf16* vb = p.v + b * p.v_bs + h * p.v_hs;
  const bf16* dob = p.dO + b * p.o_bs + h * p.o_hs;
  const __amdgpu_buffer_rsrc_t qsrc = make_rsrc(qb, (uint32_t)(((long long)(p.sq - 1) * p.q_rs + p.hd) * 2));
  const __amdgpu_buffer_rsrc_t dosrc = make_rsrc(dob, (uint32_t)(((long long)(p.sq - 1) * p.o_rs + p.hd) * 2));

  bf16x8 kf[NS], vf[NS];
  load_row_frags<HD>(kf, kb, p.k_rs, krow, p.sk, lane, p.hd);
  load_row_frags<HD>(vf, vb, p.v_rs, krow, p.sk, lane, p.hd);
  const float sc = p.scale_q_bf16 ? 1.0f : p.scale;
  const bool kok = krow < p.sk;

  // query rows that can see any key of this workgroup: q >= first_key - (sk - sq) when causal
  const int first_q = p.causal ? max(0, blockIdx.x * rows_per_blk - (p.sk - p.sq)) : 0;
  // first query row that can see any key of THIS wave (wave-uniform); whole q-tiles before it are skipped
  const int wave_first_q = k0 >= p.sk ? p.sq : (p.causal ? max(0, k0 - (p.sk - p.sq)) : 0);
  const int c_begin = first_q / CH;
  const int nchunk = (p.sq + CH - 1) / CH;

  f32x16 dkacc[NDT], dvacc[NDT];
#pragma unroll
  for (int d = 0; d < NDT; ++d)
#pragma unroll
    for (int e = 0; e < 16; ++e) dkacc[d][e] = dvacc[d][e] = 0.f;

  ChunkStage<HD> sq_, sd_;
  float st_l = 0.f, st_d = 0.f;
  auto issue_stats = [&](int c) {
    if (tid < CH) {
      const int qr = c * CH + tid;
      st_l = qr < p.sq ? p.lse[(long long)bh * p.sq + qr] : INFINITY;
      st_d = qr < p.sq ? p.delta[(long long)bh * p.sq + qr] : 0.f;
    }
  };
  auto commit_stats = [&](int buf) {
    if (tid < CH) {
      stat[buf * 2 * CH + tid] = st_l;
      stat[buf * 2 * CH + CH + tid] = st_d;
    }
  };
  if (c_begin < nchunk) {
    sq_.issue(qsrc, tid, c_begin * CH, p.sq, p.q_rs, p.hd);
    sd_.issue(dosrc, tid, c_begin * CH, p.sq, p.o_rs, p.hd);
    issue_stats(c_begin);
    if (p.scale_q_bf16 == 1) sq_.scale_bf16(p.scale);
    sq_.commit(smem, tid);
    sd_.commit(smem + CHUNK_BYTES, tid);
    commit_stats(0);
  }
  __syncthreads();
  for (int c = c_begin; c < nchunk; ++c) {
    const int cur = (c - c_begin) & 1;
    if (c + 1 < nchunk) {
      sq_.issue(qsrc, tid, (c + 1) * CH, p.sq, p.q_rs, p.hd);
      sd_.issue(dosrc, tid, (c + 1) * CH, p.sq, p.o_rs, p.hd);
      issue_stats(c + 1);
    }
    const char* ql = smem + cur * 2 * CHUNK_BYTES;
    const char* dl = ql + CHUNK_BYTES;
    const float* sl = stat + cur * 2 * CH;
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      if (c * CH + qt * 32 + 31 < wave_first_q) continue;
      f32x16 s, dp;
#pragma unroll
      for (int e = 0; e < 16; ++e) s[e] = dp[e] = 0.f;
      // S[q][key]: lane owns key (column), registers run over q rows
#pragma unroll
      for (int st = 0; st < NS; ++st) {
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(ql, qt, st, lane), kf[st], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(dl, qt, st, lane), vf[st], dp, 0, 0, 0);
      }
      f32x16 pd;  // dropped/scaled probabilities for dV
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int ql_ = qt * 32 + acc_row(e, lane);
        const int qr = c * CH + ql_;
        const int lastk = p.causal ? qr + (p.sk - p.sq) : p.sk - 1;
        const bool vis = kok && krow <= lastk;
        const float pr = vis ? __expf(s[e] * sc - sl[ql_]) : 0.f;
        float keep = 1.0f;
        if (p.drop_thr) {
          const uint64_t rb = p.offset + ((uint64_t)bh * p.sq + (uint64_t)qr) * (uint64_t)p.sk;
          keep = attn_keep(seed_r, rb, krow, p.drop_thr) ? p.drop_scale : 0.f;
        }
        pd[e] = pr * keep;
        s[e] = pr * (dp[e] * keep - sl[CH + ql_]);
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const bf16x8 pf = acc_to_frag(pd, ks);
        const bf16x8 dsf = acc_to_frag(s, ks);
#pragma unroll
        for (int d = 0; d < NDT; ++d) {
          dvacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols(dl, qt, ks, d, lane), pf, dvacc[d], 0, 0, 0);
          dkacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols(ql, qt, ks, d, lane), dsf, dkacc[d], 0, 0, 0);
        }
      }
    }
    if (c + 1 < nchunk) {
      if (p.scale_q_bf16 == 1) sq_.scale_bf16(p.scale);
      sq_.commit(smem + (cur ^ 1) * 2 * CHUNK_BYTES, tid);
      sd_.commit(smem + (cur ^ 1) * 2 * CHUNK_BYTES + CHUNK_BYTES, tid);
      commit_stats(cur ^ 1);
    }
    __syncthreads();
  }
  if (kok) {
    bf16* dkrow = p.dk + b * p.k_bs + h * p.k_hs + (long long)krow * p.k_rs;
    bf16* dvrow = p.dv + b * p.v_bs + h * p.v_hs + (long long)krow * p.v_rs;
#pragma unroll
    for (int d = 0; d < NDT; ++d)
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const int col = d * 32 + 8 * q4 + 4 * (lane >> 5);
        if (col < p.hd) {
          f32x4 a, c2;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            a[e] = dkacc[d][4 * q4 + e] * sc;
            c2[e] = dvacc[d][4 * q4 + e];
          }
          *(bf16x4*)(dkrow + col) = cvt4(a);
          *(bf16x4*)(dvrow + col) = cvt4(c2);
        }
      }
  }
}

// =========================================================================== resident variants
// Short sequences (<= 256 rows on the streamed side: GPT S<=208, ViT spatial 197): the whole K,V
// (forward, dQ) or Q,dO (dK/dV) of one (batch, head) is brought into LDS ONCE by LDS-DMA
// (buffer_load_dwordx4 ... lds; the [rows][208 B] image is lane-linear, 13 16-byte chunks per row, the
// pad chunk and out-of-range rows read as zero), every wave keeps its own 32-row tile in registers and
// then runs all tiles without further barriers.  The chunked kernels above pay one exposed HBM latency
// per 64-row chunk, which dominated at these sizes.
typedef __attribute__((address_space(3))) void lds_void_t;

// Workgroup y -> (batch, head) so that all heads of one sequence run on the SAME XCD (workgroup i lands on
// XCD i % 8): the 8 x 192-byte head slices of a qkv row share 128-byte lines, which are then fetched
// into one L2 once instead of once per XCD.  Falls back to the plain order when the grid has an x extent.
__device__ __forceinline__ bool decode_bh(const AttnArgs& p, int& b, int& h) {
  const int i = blockIdx.y;
  if (gridDim.x != 1) {
    b = i / p.heads;
    h = i - b * p.heads;
    return b < p.batch;
  }
  const int xcd = i & 7, j = i >> 3;
  h = j % p.heads;
  b = (j / p.heads) * 8 + xcd;
  return b < p.batch;
}

// PITCH = 208: conflict-free ds_read_b128 of row fragments; PITCH = 192: 16 bytes/row tighter (enough for
// ds_read_b64_tr_b16 column fragments), which is what lets two workgroups share a CU's 160 KiB at S=197.
// Only `nrows` rows are written (no padding to whole 32-row tiles): tile reads past them hit whatever
// follows in LDS -- the caller orders its regions so that data is finite where finiteness matters.
template <int HD, int PITCH>
__device__ __forceinline__ void dma_rows(const __amdgpu_buffer_rsrc_t src, char* lds, int nrows, long long rs, int wave, int nwaves,
                                         int lane, int hd = HD) {
  constexpr int CPR = PITCH / 16;
  const int prows = (nrows + 31) / 32 * 32;          // whole 32-row tiles: rows past nrows are zero-filled (out-of-range source)
  const int ninstr = (prows * CPR + 63) / 64;
  for (int i = wave; i < ninstr; i += nwaves) {
    const int g = i * 64 + lane;
    const int row = g / CPR, cc = g - row * CPR;
    const bool ok = row < nrows && cc * 8 < hd;
    const uint32_t off = ok ? (uint32_t)(((long long)row * rs + cc * 8) * 2) : 0x80000000u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(src, (lds_void_t*)(lds + i * 1024), 16, off, 0, 0, 0);
  }
}
// bytes of one LDS region of `rows` rows padded to whole tiles (device twin of the host's res_region)
__device__ __forceinline__ int region_bytes(int rows, int pitch) { return (((rows + 31) / 32 * 32) * pitch + 1023) / 1024 * 1024; }

// NTC > 0: non-causal with exactly NTC key tiles for every wave, known at compile time -- the tile loops become
// straight-line code (no per-tile branch), so the scheduler can hoist the next tile's LDS reads and MFMA chain above the
// current tile's VALU work.  NTC == 0: tile count per wave decided at run time (causal, other lengths).
// NTMAX / THREADS / WPE: an instance for short sequences (<= 32 * NTMAX keys) keeps only NTMAX score tiles and is compiled
// for WPE waves per SIMD, so that two workgroups share a CU (GPT: 160 keys, 5 waves, head_dim 64 -- one workgroup per CU
// left the kernel latency-bound: MFMA pipe 4 % busy).
template <int HD, int NTC = 0, int NTMAX = 8, int THREADS = 512, int WPE = 1>
__global__ __launch_bounds__(THREADS, WPE) void attn_fwd_res_kernel(const AttnArgs p) {
  const uint64_t seed_r = p.drop_thr ? mpv_resolve_seed(p.seed) : 0;      // (bit 63 set: the seed lives in device memory, mpv_common.h)
  constexpr int NS = HD / 16;
  constexpr int NDT = (HD + 31) / 32;
  extern __shared__ __attribute__((aligned(1024))) char rsm[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nwaves = blockDim.x >> 6;
  int b, h;
  if (!decode_bh(p, b, h)) return;
  const int bh = b * p.heads + h;
  char* vl = rsm;                                                         // [sk][208 B]
  char* kl = rsm + region_bytes(p.sk, ROWB);                               // [sk][208 B]; both regions hold whole zero-padded tiles
  const bf16* qb = p.q + b * p.q_bs + h * p.q_hs;
  const __amdgpu_buffer_rsrc_t ksrc = make_rsrc(p.k + b * p.k_bs + h * p.k_hs, (uint32_t)(((long long)(p.sk - 1) * p.k_rs + p.hd) * 2));
  const __amdgpu_buffer_rsrc_t vsrc = make_rsrc(p.v + b * p.v_bs + h * p.v_hs, (uint32_t)(((long long)(p.sk - 1) * p.v_rs + p.hd) * 2));
  dma_rows<HD, ROWB>(ksrc, kl, p.sk, p.k_rs, wave, nwaves, lane, p.hd);
  dma_rows<HD, ROWB>(vsrc, vl, p.sk, p.v_rs, wave, nwaves, lane, p.hd);
  const int q0 = (blockIdx.x * nwaves + wave) * 32;
  const int qrow = q0 + (lane & 31);
  bf16x8 qf[NS];
  load_row_frags<HD>(qf, qb, p.q_rs, qrow, p.sq, lane, p.hd);
  float sc = p.scale;
  if (p.scale_q_bf16) {
    if (p.scale_q_bf16 == 1) scale_frags_bf16(qf, p.scale);      // (2: q arrives scaled, mpv.h)
    sc = 1.0f;
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();
  if (q0 >= p.sq) return;
  const int my_last = last_visible_key(p, qrow);
  const int wave_last = last_visible_key(p, min(p.sq - 1, q0 + 31));
  const int wave_first_last = last_visible_key(p, q0);      // tiles entirely <= this need no masking
  const int nt = NTC ? NTC : wave_last / 32 + 1;
  const float c2 = sc * 1.4426950408889634f;               // work in the exp2 domain: p = 2^(s*c2 - m)
  const lds_char* sm = (const lds_char*)rsm;
  const int kbase = (int)(kl - rsm) + rows_lane_base<ROWB>(lane), vbase = cols_lane_base<ROWB>(lane);
  // One QK^T pass: all (<= 8) 32-key score tiles of the wave's 32 queries stay in registers (128 VGPRs), so the row
  // maximum is exact before the first exponential, every exponential is evaluated once, and K is read from LDS once
  // (the earlier form ran QK^T twice: once for the statistics, once for the probabilities).  Probabilities go to the
  // PV MFMAs unnormalised (<= 1, bf16); 1/l is applied to the 32 x hd output instead of the 32 x sk scores.
  constexpr int NTM = NTC ? NTC : NTMAX;
  f32x16 st[NTM];
  float mx = -INFINITY;
#pragma unroll
  for (int kt = 0; kt < NTM; ++kt) {
    if (NTC || kt < nt) {
      f32x16 s, s_odd;      // two interleaved partial sums: a 6-deep dependent MFMA chain waits on its own latency
#pragma unroll
      for (int e = 0; e < 16; ++e) s[e] = s_odd[e] = 0.f;
#pragma unroll
      for (int sx = 0; sx < NS; ++sx) {
        if (sx & 1) s_odd = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows_f(sm, kbase + kt * 32 * ROWB, sx), qf[sx], s_odd, 0, 0, 0);
        else s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows_f(sm, kbase + kt * 32 * ROWB, sx), qf[sx], s, 0, 0, 0);
      }
      s += s_odd;
      if (NTC ? kt == NTC - 1 : kt * 32 + 31 > wave_first_last) {
#pragma unroll
        for (int e = 0; e < 16; ++e) s[e] = kt * 32 + acc_row(e, lane) <= my_last ? s[e] : -INFINITY;
      }
#pragma unroll
      for (int e = 0; e < 16; e += 2) mx = fmaxf(fmaxf(mx, s[e]), s[e + 1]);      // v_max3_f32
      st[kt] = s;
    }
  }
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  const float m = mx > -INFINITY ? mx * c2 : 0.f;          // c2 > 0 (checked on the host): max(s * c2) = c2 * max(s)
  float l = 0.f;
  f32x16 oacc[NDT];
#pragma unroll
  for (int d = 0; d < NDT; ++d)
#pragma unroll
    for (int e = 0; e < 16; ++e) oacc[d][e] = 0.f;
#pragma unroll
  for (int kt = 0; kt < NTM; ++kt) {
    if (NTC || kt < nt) {
      f32x16 pr;
#pragma unroll
      for (int e = 0; e < 16; e += 2) {        // pairs: v_pk_fma_f32; masked scores are -inf -> 2^-inf = 0
        const f32x2 t = __builtin_elementwise_fma(f32x2{st[kt][e], st[kt][e + 1]}, f32x2{c2, c2}, f32x2{-m, -m});
        pr[e] = fexp2(t[0]);
        pr[e + 1] = fexp2(t[1]);
      }
      {
        f32x2 a2 = {0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 16; e += 2) a2 += f32x2{pr[e], pr[e + 1]};
        l += a2[0] + a2[1];
      }
      if (NTC == 0 && p.drop_thr) {      // (the NTC instances are launched without dropout)
        const uint64_t rb = p.offset + ((uint64_t)bh * p.sq + (uint64_t)qrow) * (uint64_t)p.sk;
#pragma unroll
        for (int e = 0; e < 16; e += 2) {      // registers e, e+1 hold keys k, k+1 with k even: one hash per pair
          const uint32_t r = mpv_rand_pair(seed_r, rb + (uint64_t)(kt * 32 + acc_row(e, lane)));
          pr[e] = (r & 0xffffu) >= p.drop_thr ? pr[e] * p.drop_scale : 0.f;
          pr[e + 1] = (r >> 16) >= p.drop_thr ? pr[e + 1] * p.drop_scale : 0.f;
        }
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const bf16x8 pf = acc_to_frag(pr, ks);
#pragma unroll
        for (int d = 0; d < NDT; ++d)
          oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols_f<ROWB>(sm, vbase + kt * 32 * ROWB, ks, d), pf, oacc[d], 0, 0, 0);
      }
    }
  }
  l += __shfl_xor(l, 32, 64);
  const float inv_l = l > 0.f ? 1.0f / l : 0.f;
  if (qrow < p.sq && lane < 32 && p.lse) p.lse[(long long)bh * p.sq + qrow] = (m + log2f(l)) * 0.6931471805599453f;
  if (qrow < p.sq) {
    bf16* orow = p.o + b * p.o_bs + h * p.o_hs + (long long)qrow * p.o_rs;
#pragma unroll
    for (int d = 0; d < NDT; ++d)
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const int col = d * 32 + 8 * q4 + 4 * (lane >> 5);
        if (col < p.hd) {
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = oacc[d][4 * q4 + e] * inv_l;
          *(bf16x4*)(orow + col) = cvt4(v);
        }
      }
  }
}

// NTC as in the forward kernel: compile-time key-tile count (non-causal, no dropout) -> straight-line tile loop.
template <int HD, int NTC = 0, int THREADS = 512, int WPE = 1>
__global__ __launch_bounds__(THREADS, WPE) void attn_bwd_dq_res_kernel(const AttnArgs p) {
  const uint64_t seed_r = p.drop_thr ? mpv_resolve_seed(p.seed) : 0;      // (bit 63 set: the seed lives in device memory, mpv_common.h)
  constexpr int NS = HD / 16;
  constexpr int NDT = (HD + 31) / 32;
  extern __shared__ __attribute__((aligned(1024))) char rsm[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nwaves = blockDim.x >> 6;
  int b, h;
  if (!decode_bh(p, b, h)) return;
  const int bh = b * p.heads + h;
  char* vl = rsm;                                                         // [sk][208 B]
  char* kl = rsm + region_bytes(p.sk, ROWB);                               // [sk][208 B]; both regions hold whole zero-padded tiles
  const __amdgpu_buffer_rsrc_t ksrc = make_rsrc(p.k + b * p.k_bs + h * p.k_hs, (uint32_t)(((long long)(p.sk - 1) * p.k_rs + p.hd) * 2));
  const __amdgpu_buffer_rsrc_t vsrc = make_rsrc(p.v + b * p.v_bs + h * p.v_hs, (uint32_t)(((long long)(p.sk - 1) * p.v_rs + p.hd) * 2));
  dma_rows<HD, ROWB>(ksrc, kl, p.sk, p.k_rs, wave, nwaves, lane, p.hd);
  dma_rows<HD, ROWB>(vsrc, vl, p.sk, p.v_rs, wave, nwaves, lane, p.hd);
  const int q0 = (blockIdx.x * nwaves + wave) * 32;
  const int qrow = q0 + (lane & 31);
  bf16x8 qf[NS], dof[NS], of[NS];
  load_row_frags<HD>(qf, p.q + b * p.q_bs + h * p.q_hs, p.q_rs, qrow, p.sq, lane, p.hd);
  load_row_frags<HD>(dof, p.dO + b * p.o_bs + h * p.o_hs, p.o_rs, qrow, p.sq, lane, p.hd);
  load_row_frags<HD>(of, p.o + b * p.o_bs + h * p.o_hs, p.o_rs, qrow, p.sq, lane, p.hd);
  const bool qok = qrow < p.sq;
  const float lse_any = p.lse[(long long)bh * p.sq + min(qrow, p.sq - 1)];      // (all requests of the block before the first use of any)
  float sc = p.scale;
  if (p.scale_q_bf16) {
    if (p.scale_q_bf16 == 1) scale_frags_bf16(qf, p.scale);      // (2: q arrives scaled, mpv.h)
    sc = 1.0f;
  }
  const float lse = qok ? lse_any : INFINITY;
  const float dl = row_delta_from(dof, of);
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();
  if (qok && lane < 32) p.delta[(long long)bh * p.sq + qrow] = dl;      // (behind the barrier: its acknowledgement is not waited for with the images)
  if (q0 >= p.sq) return;
  const int my_last = last_visible_key(p, qrow);
  const int nt = NTC ? NTC : last_visible_key(p, min(p.sq - 1, q0 + 31)) / 32 + 1;
  const int wave_first_last = last_visible_key(p, q0);
  const float c2 = sc * 1.4426950408889634f, lse2 = lse * 1.4426950408889634f;
  f32x16 dqacc[NDT];
#pragma unroll
  for (int d = 0; d < NDT; ++d)
#pragma unroll
    for (int e = 0; e < 16; ++e) dqacc[d][e] = 0.f;
  const lds_char* sm = (const lds_char*)rsm;
  const int kroff = (int)(kl - rsm) + rows_lane_base<ROWB>(lane), vroff = rows_lane_base<ROWB>(lane);
  const int kcoff = (int)(kl - rsm) + cols_lane_base<ROWB>(lane);
  auto tile = [&](int kt) {
    f32x16 s, dp;
#pragma unroll
    for (int e = 0; e < 16; ++e) s[e] = dp[e] = 0.f;
{   // four interleaved partial sums (see above)
  f32x16 s_odd, dp_odd;
#pragma unroll
  for (int e = 0; e < 16; ++e) s_odd[e] = dp_odd[e] = 0.f;
#pragma unroll
  for (int st = 0; st < NS; ++st) {
    if (st & 1) {
      s_odd = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows_f(sm, kroff + kt * 32 * ROWB, st), qf[st], s_odd, 0, 0, 0);
      dp_odd = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows_f(sm, vroff + kt * 32 * ROWB, st), dof[st], dp_odd, 0, 0, 0);
    } else {
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows_f(sm, kroff + kt * 32 * ROWB, st), qf[st], s, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows_f(sm, vroff + kt * 32 * ROWB, st), dof[st], dp, 0, 0, 0);
    }
  }
  s += s_odd;
  dp += dp_odd;
}
    if (NTC ? kt < NTC - 1 : (kt * 32 + 31 <= wave_first_last && !p.drop_thr)) {
#pragma unroll
      for (int e = 0; e < 16; e += 2) {      // pairs: v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32 around the two exponentials
        const f32x2 t = __builtin_elementwise_fma(f32x2{s[e], s[e + 1]}, f32x2{c2, c2}, f32x2{-lse2, -lse2});
        const f32x2 pr = {fexp2(t[0]), fexp2(t[1])};
        const f32x2 ds = pr * (f32x2{dp[e], dp[e + 1]} - f32x2{dl, dl});
        s[e] = ds[0];
        s[e + 1] = ds[1];
      }
    } else {
      const uint64_t rb = p.offset + ((uint64_t)bh * p.sq + (uint64_t)qrow) * (uint64_t)p.sk;
#pragma unroll
      for (int e = 0; e < 16; e += 2) {      // registers e, e+1 hold keys k, k+1 with k even: one hash word per pair (attn_keep)
        const int key = kt * 32 + acc_row(e, lane);
        uint32_t r = 0xffffffffu;
        if (NTC == 0 && p.drop_thr) r = mpv_rand_pair(seed_r, rb + (uint64_t)key);
        const float p0 = key <= my_last ? fexp2(s[e] * c2 - lse2) : 0.f;
        const float p1 = key + 1 <= my_last ? fexp2(s[e + 1] * c2 - lse2) : 0.f;
        const float d0 = (r & 0xffffu) >= p.drop_thr ? dp[e] * p.drop_scale : 0.f;      // drop_thr = 0: kept, drop_scale = 1
        const float d1 = (r >> 16) >= p.drop_thr ? dp[e + 1] * p.drop_scale : 0.f;
        s[e] = key <= my_last ? p0 * (d0 - dl) : 0.f;
        s[e + 1] = key + 1 <= my_last ? p1 * (d1 - dl) : 0.f;
      }
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const bf16x8 dsf = acc_to_frag(s, ks);
#pragma unroll
      for (int d = 0; d < NDT; ++d)
        dqacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols_f<ROWB>(sm, kcoff + kt * 32 * ROWB, ks, d), dsf, dqacc[d], 0, 0, 0);
    }
  };
  if constexpr (NTC > 0) {
#pragma unroll
    for (int kt = 0; kt < NTC; ++kt) tile(kt);
  } else {
    for (int kt = 0; kt < nt; ++kt) tile(kt);
  }
  if (qok) {
    bf16* row = p.dq + b * p.q_bs + h * p.q_hs + (long long)qrow * p.q_rs;
#pragma unroll
    for (int d = 0; d < NDT; ++d)
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const int col = d * 32 + 8 * q4 + 4 * (lane >> 5);
        if (col < p.hd) {
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = dqacc[d][4 * q4 + e] * p.scale;
          *(bf16x4*)(row + col) = cvt4(v);
        }
      }
  }
}

// One workgroup of ceil(sk / 32) waves per (batch, head): Q and dO are loaded once; the 512-thread bound keeps the kernel
// within 256 VGPRs (2 waves per SIMD: one 5..8-wave workgroup per CU; the <64, 320, 3> instance fits 168 for two 5-wave
// workgroups).  Both 32 x HD accumulator sets are live over ONE pass over the q-tiles (measured 80 us against 106 us for a
// two-pass form on the GPT shape).  The ViT shape (head_dim 96, 7 tiles) runs on attn_bwd_dkv_duo96_kernel instead.
template <int HD, int THREADS, int WPE = 1>
__global__ __launch_bounds__(THREADS, WPE) void attn_bwd_dkv_res_kernel(const AttnArgs p) {
  const MpvSeedKeys seed_k = mpv_seed_keys(p.drop_thr ? mpv_resolve_seed(p.seed) : 0);      // (bit 63 set: the seed lives in device memory, mpv_common.h)
  constexpr int NS = HD / 16;
  constexpr int NDT = (HD + 31) / 32;
  extern __shared__ __attribute__((aligned(1024))) char rsm[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nwaves = blockDim.x >> 6;
  int b, h;
  if (!decode_bh(p, b, h)) return;
  ASTAMP(0);
  const int bh = b * p.heads + h;
  const int qtiles = (p.sq + 31) / 32, qrows = qtiles * 32;
  // LDS order matters: tile over-reads of dO land in Q (finite), over-reads of Q run past the allocation
  // (hardware returns 0 for out-of-range LDS reads); the fp32 stats sit in front so no bf16 fragment read
  // can ever interpret them as (possibly Inf/NaN) bf16.
  float* sl = (float*)rsm;                                                  // [lse*log2e | delta] x qrows
  char* vl = rsm + ((2 * qrows * 4 + 1023) / 1024) * 1024;                  // V  [sk][208 B] (head_dim > 64 only: zero bytes otherwise)
  char* dl = vl + (HD > 64 ? region_bytes(p.sk, ROWB) : 0);                  // dO [sq][208 B]
  char* ql = dl + region_bytes(p.sq, ROWB);                                  // Q  [sq][208 B]
  const __amdgpu_buffer_rsrc_t qsrc = make_rsrc(p.q + b * p.q_bs + h * p.q_hs, (uint32_t)(((long long)(p.sq - 1) * p.q_rs + p.hd) * 2));
  const __amdgpu_buffer_rsrc_t dosrc = make_rsrc(p.dO + b * p.o_bs + h * p.o_hs, (uint32_t)(((long long)(p.sq - 1) * p.o_rs + p.hd) * 2));
  // The wave's K rows and the thread's first pair of statistics are requested BEFORE the LDS-DMA of the images: the compiler puts
  // s_waitcnt vmcnt(0) between a pending buffer_load..lds and everything it cannot disambiguate from it (the statistics' LDS stores,
  // and with them their loads) -- three round trips in a row where one does.
  const int k0 = (blockIdx.x * nwaves + wave) * 32;
  const int krow = k0 + (lane & 31);
  bf16x8 kf[NS];
  load_row_frags<HD>(kf, p.k + b * p.k_bs + h * p.k_hs, p.k_rs, krow, p.sk, lane, p.hd);
  const float lse_first = p.lse[(long long)bh * p.sq + min(tid, p.sq - 1)], delta_first = p.delta[(long long)bh * p.sq + min(tid, p.sq - 1)];
  dma_rows<HD, ROWB>(qsrc, ql, p.sq, p.q_rs, wave, nwaves, lane, p.hd);
  dma_rows<HD, ROWB>(dosrc, dl, p.sq, p.o_rs, wave, nwaves, lane, p.hd);
  if constexpr (HD > 64) {
    const __amdgpu_buffer_rsrc_t vsrc = make_rsrc(p.v + b * p.v_bs + h * p.v_hs, (uint32_t)(((long long)(p.sk - 1) * p.v_rs + p.hd) * 2));
    dma_rows<HD, ROWB>(vsrc, vl, p.sk, p.v_rs, wave, nwaves, lane, p.hd);
  }
  if (tid < qrows) {                                // lse pre-multiplied by log2(e): probabilities are 2^(s*c2 - lse2)
    sl[tid] = tid < p.sq ? lse_first * 1.4426950408889634f : 1e30f;
    sl[qrows + tid] = tid < p.sq ? delta_first : 0.f;
  }
  for (int r = tid + blockDim.x; r < qrows; r += blockDim.x) {
    sl[r] = r < p.sq ? p.lse[(long long)bh * p.sq + r] * 1.4426950408889634f : 1e30f;
    sl[qrows + r] = r < p.sq ? p.delta[(long long)bh * p.sq + r] : 0.f;
  }
  const float sc = p.scale_q_bf16 ? 1.0f : p.scale;
  const float c2 = sc * 1.4426950408889634f;
  const bool kok = krow < p.sk;
  ASTAMP(1);
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();
  ASTAMP(2);
  if (p.scale_q_bf16 == 1) {   // q' = bf16(q * scale) in place, once (2: q arrives scaled)
    for (int g = tid; g < p.sq * (HD / 8); g += blockDim.x) {
      const int row = g / (HD / 8), cc = g - row * (HD / 8);
      bf16x8* ptr = (bf16x8*)(ql + row * ROWB + cc * 16);
      f32x8 f = cvt8(*ptr);
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] *= p.scale;
      *ptr = cvt8(f);
    }
    __syncthreads();
  }
  ASTAMP(3);
  if (k0 >= p.sk) return;
  const int first_q = p.causal ? max(0, k0 - (p.sk - p.sq)) : 0;
  const lds_char* sm = (const lds_char*)rsm;
  const int qroff = (int)(ql - rsm) + rows_lane_base<ROWB>(lane), droff = (int)(dl - rsm) + rows_lane_base<ROWB>(lane);
  const int qcoff = (int)(ql - rsm) + cols_lane_base<ROWB>(lane), dcoff = (int)(dl - rsm) + cols_lane_base<ROWB>(lane);
  auto store_rows = [&](const f32x16 (&acc)[NDT], bf16* row, float mul) {
    if (kok) {
#pragma unroll
      for (int d = 0; d < NDT; ++d)
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          const int col = d * 32 + 8 * q4 + 4 * (lane >> 5);
          if (col < p.hd) {
            f32x4 a;
#pragma unroll
            for (int e = 0; e < 4; ++e) a[e] = acc[d][4 * q4 + e] * mul;
            *(bf16x4*)(row + col) = cvt4(a);
          }
        }
    }
  };
  // One pass over the q-tiles, S, P, dP and dS computed once, both 32 x HD accumulator sets live.  head_dim <= 64 keeps
  // the wave's K and V fragments in registers; above that the V fragments (24 more VGPRs at 96) are read from a third LDS
  // region instead (a round-2 decision, when the register form spilled inside the loop; the current code compiles to 220 VGPRs
  // without scratch in that form too -- attention_duo.inc's dK/dV kernel uses it -- but the one-shot instances were not
  // re-measured with it).
  constexpr bool V_LDS = HD > 64;
  bf16x8 vf[V_LDS ? 1 : NS];
  if constexpr (!V_LDS) load_row_frags<HD>(vf, p.v + b * p.v_bs + h * p.v_hs, p.v_rs, krow, p.sk, lane, p.hd);
  const int vfoff = (int)(vl - rsm) + rows_lane_base<ROWB>(lane) + k0 * ROWB;
  f32x16 dkacc[NDT], dvacc[NDT];
#pragma unroll
  for (int d = 0; d < NDT; ++d)
#pragma unroll
    for (int e = 0; e < 16; ++e) dkacc[d][e] = dvacc[d][e] = 0.f;
  for (int qt = first_q / 32; qt < qtiles; ++qt) {
    f32x16 s, dp;
#pragma unroll
    for (int e = 0; e < 16; ++e) s[e] = dp[e] = 0.f;
    int vo = vfoff, lane_t = lane;      // lane_t: the per-element row indices / dropout counters derived from it are loop-invariant
    asm volatile("" : "+v"(vo), "+v"(lane_t));      // opaque per iteration: otherwise the loop-invariant V fragment reads are hoisted back into registers
#pragma unroll
    for (int st = 0; st < NS; ++st) {
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows_f(sm, qroff + qt * 32 * ROWB, st), kf[st], s, 0, 0, 0);
      if constexpr (V_LDS) dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows_f(sm, droff + qt * 32 * ROWB, st), frag_rows_f(sm, vo, st), dp, 0, 0, 0);
      else dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows_f(sm, droff + qt * 32 * ROWB, st), vf[st], dp, 0, 0, 0);
    }
    // P and dS in one sweep (one exponential and one dropout draw per element), half a tile at a time: accumulator
    // registers 8*ks .. 8*ks+7 are exactly the B operand of reduction step ks, so P never exists as a whole fp32 tile.
    // A q-tile is "interior" for this wave when every (q,key) pair is visible and in range.
    const bool interior = !p.drop_thr && (k0 + 31 < p.sk) && (qt * 32 + 31 < p.sq) &&
                          (!p.causal || (k0 + 31 <= qt * 32 + (p.sk - p.sq)));
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      f32x8 pr8, ds8;
#pragma unroll
      for (int qh = 0; qh < 2; ++qh) {
        const int q4 = 2 * ks + qh;
        const int qb4 = qt * 32 + 8 * q4 + 4 * (lane_t >> 5);
        const f32x4 l4 = *(const f32x4*)(sl + qb4), d4 = *(const f32x4*)(sl + qrows + qb4);
        if (interior) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int e = 4 * q4 + j;
            const float pr = fexp2(s[e] * c2 - l4[j]);
            pr8[4 * qh + j] = pr;
            ds8[4 * qh + j] = pr * (dp[e] - d4[j]);
          }
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int e = 4 * q4 + j;
            const int qr = qb4 + j;
            const int lastk = p.causal ? qr + (p.sk - p.sq) : p.sk - 1;
            const bool vis = kok && krow <= lastk && qr < p.sq;
            const float pr = vis ? fexp2(s[e] * c2 - l4[j]) : 0.f;
            float keep = 1.0f;
            if (p.drop_thr) {
              const uint64_t rb = p.offset + ((uint64_t)bh * p.sq + (uint64_t)qr) * (uint64_t)p.sk;
              const uint32_t rw = mpv_rand_pair_k(seed_k, rb + (uint64_t)(krow & ~1));      // (= attn_keep)
              keep = ((krow & 1) ? rw >> 16 : rw & 0xffffu) >= p.drop_thr ? p.drop_scale : 0.f;
            }
            pr8[4 * qh + j] = vis ? pr * keep : 0.f;
            ds8[4 * qh + j] = vis ? pr * (dp[e] * keep - d4[j]) : 0.f;
          }
        }
      }
      const bf16x8 pf = cvt8(pr8), dsf = cvt8(ds8);
#pragma unroll
      for (int d = 0; d < NDT; ++d) {
        dvacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols_f<ROWB>(sm, dcoff + qt * 32 * ROWB, ks, d), pf, dvacc[d], 0, 0, 0);
        dkacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols_f<ROWB>(sm, qcoff + qt * 32 * ROWB, ks, d), dsf, dkacc[d], 0, 0, 0);
      }
    }
  }
  ASTAMP(4);
  // (row pointers formed only now: as loop-carried live values they were the first thing the allocator spilled)
  store_rows(dvacc, p.dv + b * p.v_bs + h * p.v_hs + (long long)krow * p.v_rs, 1.0f);
  store_rows(dkacc, p.dk + b * p.k_bs + h * p.k_hs + (long long)krow * p.k_rs, sc);
  ASTAMP(5);
}

// =========================================================================== persistent resident variants
// The resident kernels above run as rounds of workgroups that all load, then all compute, then all store: with one or two
// workgroups per CU nothing overlaps the load phase of a (batch, head) problem (every CU asks HBM for its 80-120 KB at the
// same moment, then the memory system idles while every CU computes).  Here ONE workgroup per CU walks a list of (batch,
// head) items and keeps two LDS operand sets: while item k is computed out of set `cur`, the operands of item k+1 arrive in
// the other set by LDS-DMA (inline asm: hipcc knows nothing of these loads, so it neither waits for them nor orders its own
// LDS reads behind them; the ONE wait is the explicit vmcnt(0) at the end of an item, placed before the item's output stores
// so that those stay in flight across the barrier).  Register operands of the next item (the wave's own Q / dO / K / V rows)
// are requested at the point where the current item's copy is dead, into the same registers.
//   item i -> (batch, head) exactly as decode_bh: i % 8 = XCD, so the heads of one sequence share an L2.
// Measured (profiles/r03_c1_kernel_trace.md, config B in-step): the 7-tile ViT forward 108 -> 87 us per layer.  The same
// structure for the decoder (head_dim 64, 5 waves per item) LOST to the one-shot kernels -- forward 34.9 vs 29.0 us, dQ 48.0 vs
// 48.9, dK/dV 62.3 vs 48.6: those items are bound by the dependent chain inside one wave (few tiles, a dropout hash per
// element), and two co-resident one-shot workgroups hide more of it than one workgroup with a hidden load -- so only the
// forward instances of the ViT shape are kept.
// LDS images are UNPADDED [rows][pitch] with only the region size rounded up to whole 1 KiB DMA pieces (the tail of the last
// piece is zero-filled): a 32-row tile read past the last row lands in the next region, which always holds finite data of
// some item (or, past the allocation, reads as zero) -- every product with such a row is masked by a select or multiplied
// by an exact zero probability.
template <int HD>
struct PresPitch {
  static constexpr int R = HD * 2 + 16;                     // row-fragment (ds_read_b128) images: conflict-free at 144 / 176 / 208 B
  static constexpr int C = HD > 80 ? HD * 2 : HD * 2 + 16;  // images that only serve transposing column reads: 192 B at head_dim 96
};
__device__ __forceinline__ int pres_region(int rows, int pitch) { return (rows * pitch + 1023) / 1024 * 1024; }
__device__ __forceinline__ i32x4 pres_rsrc(const void* ptr, uint32_t bytes) {
  const uint64_t a = (uint64_t)ptr;
  return i32x4{(int)(uint32_t)a, (int)((uint32_t)(a >> 32) & 0xffffu), (int)bytes, 0x00020000};
}
// one 1 KiB LDS-DMA piece: lane l's 16 bytes go to LDS byte lds_addr + 16 * l (m0 = wave-uniform base)
__device__ __forceinline__ void pres_dma16(i32x4 rsrc, uint32_t lds_addr, uint32_t voff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 3\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" : : "s"(lds_addr), "v"(voff), "s"(rsrc) : "memory");
}
template <int PITCH>
__device__ __forceinline__ void pres_dma_rows(i32x4 rsrc, uint32_t lds_base, int nrows, int rs, int wave, int nwaves, int lane, int hd) {
  constexpr int CPR = PITCH / 16;
  const int ninstr = (nrows * CPR + 63) >> 6;
  for (int i = wave; i < ninstr; i += nwaves) {
    const int g = i * 64 + lane;
    const int row = g / CPR, cc = g - row * CPR;
    const uint32_t off = (row < nrows && cc * 8 < hd) ? (uint32_t)((row * rs + cc * 8) * 2) : 0x80000000u;
    pres_dma16(rsrc, lds_base + (uint32_t)i * 1024u, off);
  }
}
// vmcnt(0) as the BUILTIN, not as asm text: the compiler's wait-count bookkeeping then knows that every load it has issued
// itself is complete here as well -- with an asm wait it went on to protect the loop-carried prefetch registers with its own
// s_waitcnt vmcnt(N) right behind the next item's DMA issue (N counts only the loads it knows: the wait drained the DMA)
__device__ __forceinline__ void pres_wait_all() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0), expcnt / lgkmcnt untouched
  asm volatile("" ::: "memory");
}
// item -> (batch, head); false for the padding items of a batch that is not a multiple of 8
__device__ __forceinline__ bool pres_item(const AttnArgs& p, int it, int& b, int& h) {
  const int j = it >> 3;
  h = j % p.heads;
  b = (j / p.heads) * 8 + (it & 7);
  return b < p.batch;
}
__device__ __forceinline__ int pres_next(const AttnArgs& p, int it, int step, int nitems) {
  int b, h;
  while (it < nitems && !pres_item(p, it, b, h)) it += step;
  return it;
}

// the lane id recomputed where it is needed (two VALU instructions) instead of a value kept live across the item loop (the
// allocator parked threadIdx.x in scratch and re-read it mid-item behind an s_waitcnt vmcnt(0) -- which also waits for the
// next item's LDS-DMA)
__device__ __forceinline__ int fresh_lane() {
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}
// row fragments through a buffer descriptor: out-of-range rows / columns read as zero without a branch around the load
template <int HD>
__device__ __forceinline__ void load_row_frags_buf(bf16x8 (&f)[HD / 16], const bf16* base, uint32_t bytes, int rs, int row, int nrows, int lane, int hd) {
  const __amdgpu_buffer_rsrc_t r = make_rsrc(base, bytes);
  const uint32_t r0 = (uint32_t)((row * rs + (lane >> 5) * 8) * 2);
#pragma unroll
  for (int s = 0; s < HD / 16; ++s) {
    const bool ok = row < nrows && s * 16 + (lane >> 5) * 8 < hd;
    union { i32x4 i; bf16x8 b; } u;
    u.i = __builtin_amdgcn_raw_buffer_load_b128(r, ok ? r0 + (uint32_t)(s * 32) : 0x80000000u, 0, 0);
    f[s] = u.b;
  }
}
// the whole dynamic LDS allocation cleared once per workgroup: tile over-reads then only ever see zeros or operand data
__device__ __forceinline__ void pres_clear_lds(char* rsm, int bytes) {
  for (int i = threadIdx.x * 16; i < bytes; i += blockDim.x * 16) *(i32x4*)(rsm + i) = i32x4{0, 0, 0, 0};
  __syncthreads();
}
// (batch, head) offset of an item in 32-bit elements
__device__ __forceinline__ uint32_t pres_off(int b, int h, long long bs, long long hs) { return (uint32_t)b * (uint32_t)bs + (uint32_t)h * (uint32_t)hs; }

// LEAN (with NTC): the last key tile holds at most 8 keys (ViT-B/16: 197 = 6 * 32 + 5), i.e. only accumulator registers 0..3 of
// its score tile can be visible -- the other 12 are not kept (registers), not exponentiated, and the second reduction step of
// its PV product (keys +16..+31 of the tile) is not issued.
template <int HD, int NTC = 0, int NTMAX = 8, int THREADS = 512, int WPE = 1, bool LEAN = false>
__global__ __launch_bounds__(THREADS, WPE) void attn_fwd_pres_kernel(const AttnArgs p, const int nitems) {
  constexpr int NS = HD / 16;
  constexpr int NDT = (HD + 31) / 32;
  constexpr int KP = PresPitch<HD>::R, VP = PresPitch<HD>::C;
  extern __shared__ __attribute__((aligned(1024))) char rsm[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nwaves = blockDim.x >> 6;
  const int vbytes = pres_region(p.sk, VP), setb = vbytes + pres_region(p.sk, KP);      // set = [V | K]
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void_t*)rsm;
  const int step = gridDim.x;
  const int q0 = wave * 32;
  const bool active = q0 < p.sq;
  auto issue = [&](int it, int buf) {
    int b, h;
    pres_item(p, it, b, h);
    const int lane = fresh_lane();
    const uint32_t kspan = (uint32_t)(((p.sk - 1) * (int)p.k_rs + p.hd) * 2), vspan = (uint32_t)(((p.sk - 1) * (int)p.v_rs + p.hd) * 2);
    pres_dma_rows<KP>(pres_rsrc(p.k + pres_off(b, h, p.k_bs, p.k_hs), kspan), lds0 + buf * setb + vbytes, p.sk, (int)p.k_rs, wave, nwaves, lane, p.hd);
    pres_dma_rows<VP>(pres_rsrc(p.v + pres_off(b, h, p.v_bs, p.v_hs), vspan), lds0 + buf * setb, p.sk, (int)p.v_rs, wave, nwaves, lane, p.hd);
  };
  bf16x8 qf[NS];
  auto load_q = [&](int it) {
    int b, h;
    pres_item(p, it, b, h);
    const int lane = fresh_lane();
    load_row_frags_buf<HD>(qf, p.q + pres_off(b, h, p.q_bs, p.q_hs), (uint32_t)(((p.sq - 1) * (int)p.q_rs + p.hd) * 2), (int)p.q_rs, q0 + (lane & 31), p.sq, lane, p.hd);
  };
  int it = pres_next(p, blockIdx.x, step, nitems);
  if (it >= nitems) return;
  pres_clear_lds(rsm, 2 * setb);
  issue(it, 0);
  load_q(it);
  pres_wait_all();
  __syncthreads();
  const lds_char* sm = (const lds_char*)rsm;
  constexpr int NTM = NTC ? NTC : NTMAX;
  int cur = 0;
  while (true) {
    const int nx = pres_next(p, it + step, step, nitems);
    if (nx < nitems) issue(nx, cur ^ 1);
    int b, h;
    pres_item(p, it, b, h);
    const int bh = b * p.heads + h;
    const int lane = fresh_lane();
    const int qrow = q0 + (lane & 31);
    if (active) {
      float l = 0.f, m = 0.f;
      const int my_last = last_visible_key(p, qrow);
      const int wave_last = last_visible_key(p, min(p.sq - 1, q0 + 31));
      const int wave_first_last = last_visible_key(p, q0);      // tiles entirely <= this need no masking
      const int nt = NTC ? NTC : wave_last / 32 + 1;
      const float c2 = (p.scale_q_bf16 ? 1.0f : p.scale) * 1.4426950408889634f;               // exp2 domain: p = 2^(s*c2 - m)
      if (p.scale_q_bf16 == 1) scale_frags_bf16(qf, p.scale);
      const int kbase = cur * setb + vbytes + rows_lane_base<KP>(lane), vbase = cur * setb + cols_lane_base<VP>(lane);
      static_assert(!LEAN || NTC > 1, "LEAN needs a compile-time tile count");
      f32x16 st[LEAN ? NTM - 1 : NTM];
      f32x4 st_last = {0.f, 0.f, 0.f, 0.f};
      float mx = -INFINITY;
#pragma unroll
      for (int kt = 0; kt < NTM; ++kt) {
        if (NTC || kt < nt) {
          f32x16 s;       // one accumulator chain per tile: the second wave of the SIMD covers its latency, and the 16 registers of
#pragma unroll  // a second partial sum are what tipped the 7-tile instance into scratch
          for (int e = 0; e < 16; ++e) s[e] = 0.f;
#pragma unroll
          for (int sx = 0; sx < NS; ++sx) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows_f(sm, kbase + kt * 32 * KP, sx), qf[sx], s, 0, 0, 0);
          if (NTC ? kt == NTC - 1 : kt * 32 + 31 > wave_first_last) {
#pragma unroll
            for (int e = 0; e < 16; ++e) s[e] = kt * 32 + acc_row(e, lane) <= my_last ? s[e] : -INFINITY;
          }
          if (LEAN && kt == NTC - 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) st_last[e] = s[e];
            mx = fmaxf(fmaxf(mx, s[0]), fmaxf(s[1], fmaxf(s[2], s[3])));
          } else {
#pragma unroll
            for (int e = 0; e < 16; e += 2) mx = fmaxf(fmaxf(mx, s[e]), s[e + 1]);
            st[LEAN && kt == NTC - 1 ? 0 : kt] = s;
          }
        }
      }
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      m = mx > -INFINITY ? mx * c2 : 0.f;
      f32x16 oacc[NDT];
#pragma unroll
      for (int d = 0; d < NDT; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) oacc[d][e] = 0.f;
#pragma unroll
      for (int kt = 0; kt < NTM; ++kt) {
        // the wave's Q rows are dead once the scores exist: the next item's go into the same registers under the PV phase --
        // requested only after the first score tile has been consumed (its 16 registers make room for the fragments in flight)
        if (kt == (NTC ? NTC / 2 : (NTM > 1 ? 1 : 0))) {
          asm volatile("" ::: "memory");
          if (nx < nitems) load_q(nx);
        }
        if (NTC || kt < nt) {
          f32x16 pr;
          if (LEAN && kt == NTC - 1) {
#pragma unroll
            for (int e = 0; e < 16; ++e) pr[e] = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) pr[e] = fexp2(fmaf(st_last[e], c2, -m));
            l += (pr[0] + pr[1]) + (pr[2] + pr[3]);
          } else {
            const f32x16& sc16 = st[LEAN && kt == NTC - 1 ? 0 : kt];
#pragma unroll
            for (int e = 0; e < 16; e += 2) {
              const f32x2 t = __builtin_elementwise_fma(f32x2{sc16[e], sc16[e + 1]}, f32x2{c2, c2}, f32x2{-m, -m});
              pr[e] = fexp2(t[0]);
              pr[e + 1] = fexp2(t[1]);
            }
            f32x2 a2 = {0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 16; e += 2) a2 += f32x2{pr[e], pr[e + 1]};
            l += a2[0] + a2[1];
          }
          if (NTC == 0 && p.drop_thr) {
            const uint64_t rb = p.offset + ((uint64_t)bh * p.sq + (uint64_t)qrow) * (uint64_t)p.sk;
#pragma unroll
            for (int e = 0; e < 16; e += 2) {
              const uint32_t r = mpv_rand_pair(p.seed, rb + (uint64_t)(kt * 32 + acc_row(e, lane)));
              pr[e] = (r & 0xffffu) >= p.drop_thr ? pr[e] * p.drop_scale : 0.f;
              pr[e + 1] = (r >> 16) >= p.drop_thr ? pr[e + 1] * p.drop_scale : 0.f;
            }
          }
#pragma unroll
          for (int ks = 0; ks < ((LEAN && kt == NTC - 1) ? 1 : 2); ++ks) {
            const bf16x8 pf = acc_to_frag(pr, ks);
#pragma unroll
            for (int d = 0; d < NDT; ++d)
              oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols_f<VP>(sm, vbase + kt * 32 * VP, ks, d), pf, oacc[d], 0, 0, 0);
          }
        }
      }
      l += __shfl_xor(l, 32, 64);
      pres_wait_all();          // the next item's operands have landed (this wave's pieces); older stores are done
      const int lane_e = fresh_lane();      // fresh: the store addresses are formed here, not at the top of the item and then kept live
      const int qrow_e = q0 + (lane_e & 31);
      if (qrow_e < p.sq) {
        const float inv_l = l > 0.f ? 1.0f / l : 0.f;
        if (lane_e < 32 && p.lse) p.lse[(uint32_t)(bh * p.sq + qrow_e)] = (m + log2f(l)) * 0.6931471805599453f;
        bf16* ob = p.o + pres_off(b, h, p.o_bs, p.o_hs);
        const uint32_t ro = (uint32_t)(qrow_e * (int)p.o_rs + 4 * (lane_e >> 5));
#pragma unroll
        for (int d = 0; d < NDT; ++d)
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4) {
            const int col = d * 32 + 8 * q4;
            if (col + 4 * (lane_e >> 5) < p.hd) {
              f32x4 v;
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = oacc[d][4 * q4 + e] * inv_l;
              *(bf16x4*)(ob + (ro + (uint32_t)col)) = cvt4(v);
            }
          }
      }
    } else {
      if (nx < nitems) load_q(nx);
      pres_wait_all();
    }
    __syncthreads();          // every wave's pieces are in, and nobody still reads set `cur`
    if (nx >= nitems) break;
    it = nx;
    cur ^= 1;
  }
}

// =========================================================================== temporal attention (VALU)
struct TempArgs {
  const bf16* qkv;
  const bf16* dout;
  bf16* out;
  bf16* dqkv;
  int n_outer, n_inner;
  long long outer_stride, inner_offset, t_stride;
  int T, heads, hd;
  float scale;
};

// one wave per (sequence, head); T <= 16, hd <= 96 (multiple of 4).  Rows live in LDS as fp32 with a pitch of
// hd+4 floats so every inner product runs on 16-byte ds_read_b128 operands (conflict-free across rows).
// LDS floats per wave: (3|4)*T*(hd+4) + (1|2)*T*(T+1).
// TC / HDC > 0: T and head_dim known at compile time (the ViT-B/16 instances: 4 frames -- the shipped pre-train YAML --, 8 -- the
// benchmark configuration -- and 16 -- the retrieval recipe --, head_dim 96).  With run-time trip
// counts none of the inner loops unrolls, every iteration waits on its own LDS reads (~130 clocks each: measured 15k
// clocks per problem for ~450 instructions); unrolled, the reads of 8 iterations are in flight together.
template <bool BWD, int TC = 0, int HDC = 0>
__global__ __launch_bounds__(256) void temporal_attn_kernel(const TempArgs p) {
  extern __shared__ __attribute__((aligned(16))) float tsm[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int T = TC ? TC : p.T, hd = HDC ? HDC : p.hd, ld = hd + 4, D = p.heads * hd, H4 = hd / 4;
  const int per_wave = (BWD ? 4 : 3) * T * ld + (BWD ? 2 : 1) * T * (T + 1);
  float* qs = tsm + wave * ((per_wave + 3) & ~3);
  float* ks = qs + T * ld;
  float* vs = ks + T * ld;
  float* dos = vs + T * ld;                         // BWD only: dO [T][ld]
  float* ps = (BWD ? dos + T * ld : dos);           // [T][T+1] probabilities
  float* dss = ps + T * (T + 1);                    // BWD only: dS [T][T+1]
  const long long nprob = (long long)p.n_outer * p.n_inner * p.heads;
  const int nwv = (int)(blockDim.x >> 6);           // waves (= problems in flight) per workgroup: 4, fewer where LDS packs better (T = 16)
  const long long pstride = (long long)gridDim.x * nwv;
  // The operands of a wave's NEXT problem are requested (into registers) before the current one is computed: a problem is a
  // load -> LDS -> three dependent LDS phases -> store chain, and with 12-16 waves per CU the memory system saw the loads of
  // one phase at a time (4.1-4.5 TB/s); compile-time instances only (the register arrays need constant trip counts).
  constexpr bool PF = TC > 0 && HDC > 0;
  constexpr int NX = PF ? (TC * (HDC / 4) + 63) / 64 : 1;
  bf16x4 rq[NX], rk[NX], rv[NX], rd[BWD ? NX : 1];
  auto row_of = [&](long long pr, int& h) {
    h = (int)(pr % p.heads);
    const long long seq = pr / p.heads;
    const long long o = seq / p.n_inner, i = seq % p.n_inner;
    return o * p.outer_stride + p.inner_offset + i;
  };
  auto request = [&](long long pr) {
    int h;
    const long long row0 = row_of(pr, h);
#pragma unroll
    for (int j = 0; j < NX; ++j) {
      const int x = lane + 64 * j;
      if (x < T * H4) {
        const int t = x / H4, c4 = x - t * H4;
        const bf16* src = p.qkv + (row0 + t * p.t_stride) * (3LL * D) + h * hd + c4 * 4;
        rq[j] = *(const bf16x4*)src;
        rk[j] = *(const bf16x4*)(src + D);
        rv[j] = *(const bf16x4*)(src + 2 * D);
        if constexpr (BWD) rd[j] = *(const bf16x4*)(p.dout + (row0 + t * p.t_stride) * (long long)D + h * hd + c4 * 4);
      }
    }
  };
  long long pr0 = (long long)blockIdx.x * nwv + wave;
  if (PF && pr0 < nprob) request(pr0);
  for (long long pr = pr0; pr < nprob; pr += pstride) {
    int h;
    const long long row0 = row_of(pr, h);
    if constexpr (PF) {
#pragma unroll
      for (int j = 0; j < NX; ++j) {
        const int x = lane + 64 * j;
        if (x < T * H4) {
          const int t = x / H4, c4 = x - t * H4;
          f32x4 qv = cvt4(rq[j]) * p.scale;
          qv = cvt4(cvt4(qv));                                      // q*scale rounds to bf16 (reference :179)
          *(f32x4*)(qs + t * ld + c4 * 4) = qv;
          *(f32x4*)(ks + t * ld + c4 * 4) = cvt4(rk[j]);
          *(f32x4*)(vs + t * ld + c4 * 4) = cvt4(rv[j]);
          if constexpr (BWD) *(f32x4*)(dos + t * ld + c4 * 4) = cvt4(rd[j]);
        }
      }
      if (pr + pstride < nprob) request(pr + pstride);
    } else {
#pragma unroll
    for (int x = lane; x < T * H4; x += 64) {
      const int t = x / H4, c4 = x - t * H4;
      const bf16* src = p.qkv + (row0 + t * p.t_stride) * (3LL * D) + h * hd + c4 * 4;
      f32x4 qv = cvt4(*(const bf16x4*)src) * p.scale;
      qv = cvt4(cvt4(qv));                                        // q*scale rounds to bf16 (reference :179)
      *(f32x4*)(qs + t * ld + c4 * 4) = qv;
      *(f32x4*)(ks + t * ld + c4 * 4) = cvt4(*(const bf16x4*)(src + D));
      *(f32x4*)(vs + t * ld + c4 * 4) = cvt4(*(const bf16x4*)(src + 2 * D));
      if constexpr (BWD)
        *(f32x4*)(dos + t * ld + c4 * 4) = cvt4(*(const bf16x4*)(p.dout + (row0 + t * p.t_stride) * (long long)D + h * hd + c4 * 4));
    }
    }
    WAVE_SYNC();
#pragma unroll
    for (int x = lane; x < T * T; x += 64) {
      const int a = x / T, bb = x - a * T;
      f32x4 acc4 = {0.f, 0.f, 0.f, 0.f}, dp4 = acc4;
#pragma unroll 8
      for (int c4 = 0; c4 < H4; ++c4) {
        const f32x4 kv = *(const f32x4*)(ks + bb * ld + c4 * 4);
        acc4 += *(const f32x4*)(qs + a * ld + c4 * 4) * kv;
        if constexpr (BWD) dp4 += *(const f32x4*)(dos + a * ld + c4 * 4) * *(const f32x4*)(vs + bb * ld + c4 * 4);
      }
      ps[a * (T + 1) + bb] = (acc4[0] + acc4[1]) + (acc4[2] + acc4[3]);
      if constexpr (BWD) dss[a * (T + 1) + bb] = (dp4[0] + dp4[1]) + (dp4[2] + dp4[3]);     // dP = dO V^T
    }
    WAVE_SYNC();
    if (lane < T) {
      float mx = -INFINITY;
#pragma unroll
      for (int j = 0; j < T; ++j) mx = fmaxf(mx, ps[lane * (T + 1) + j]);
      float sum = 0.f;
#pragma unroll
      for (int j = 0; j < T; ++j) sum += __expf(ps[lane * (T + 1) + j] - mx);
      const float inv = 1.0f / sum;
      float dl = 0.f;
#pragma unroll
      for (int j = 0; j < T; ++j) {
        const float pv = __expf(ps[lane * (T + 1) + j] - mx) * inv;
        ps[lane * (T + 1) + j] = BWD ? pv : bf2f(f2bf(pv));      // forward: probs cast to bf16 (:201)
        if constexpr (BWD) dl += pv * dss[lane * (T + 1) + j];
      }
      if constexpr (BWD)                                           // dS = P * (dP - rowsum(P*dP))
#pragma unroll
        for (int j = 0; j < T; ++j) dss[lane * (T + 1) + j] = ps[lane * (T + 1) + j] * (dss[lane * (T + 1) + j] - dl);
    }
    WAVE_SYNC();
#pragma unroll
    for (int x = lane; x < T * H4; x += 64) {
      const int a = x / H4, c4 = x - a * H4;
      if constexpr (!BWD) {
        f32x4 o4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < T; ++j) o4 += *(const f32x4*)(vs + j * ld + c4 * 4) * ps[a * (T + 1) + j];
        *(bf16x4*)(p.out + (row0 + a * p.t_stride) * (long long)D + h * hd + c4 * 4) = cvt4(o4);
      } else {
        f32x4 dq = {0.f, 0.f, 0.f, 0.f}, dk = dq, dv = dq;
#pragma unroll
        for (int j = 0; j < T; ++j) {
          dq += *(const f32x4*)(ks + j * ld + c4 * 4) * dss[a * (T + 1) + j];
          dk += *(const f32x4*)(qs + j * ld + c4 * 4) * dss[j * (T + 1) + a];
          dv += *(const f32x4*)(dos + j * ld + c4 * 4) * ps[j * (T + 1) + a];
        }
        bf16* dst = p.dqkv + (row0 + a * p.t_stride) * (3LL * D) + h * hd + c4 * 4;
        *(bf16x4*)dst = cvt4(dq * p.scale);
        *(bf16x4*)(dst + D) = cvt4(dk);
        *(bf16x4*)(dst + 2 * D) = cvt4(dv);
      }
    }
    WAVE_SYNC();
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Decode attention (sq == 1: one new token per sequence against its cached keys, models/modeling_distributed_gpt3.py:
// 905-929).  One wave per (sequence, head): phase 1 -- lane j scores keys j, j+64, ... (fp32 dot over the 128/160-byte
// key rows, q broadcast from LDS), softmax statistics by wave reduction, probabilities parked in LDS; phase 2 -- lane d
// accumulates sum_j p_j V[j][d] (each key's value row is one coalesced line).  VALU only: 2*sk*hd MACs per head.
constexpr int DEC_MAX_SK = 2048;
// One workgroup per (sequence, head).  LPK lanes share one key row (16 bytes each), so a wave instruction fetches
// 64/LPK whole rows and the 4 waves cover 256/LPK keys per pass, 8 passes unrolled (the step is latency-bound: every
// load that can be in flight is).
template <int LPK>
__global__ __launch_bounds__(256) void attn_decode_kernel(const AttnArgs p) {
  __shared__ float ps[DEC_MAX_SK];
  __shared__ float red[4];
  __shared__ float accs[4][LPK * 8];
  constexpr int KPI = 256 / LPK;                      // keys per workgroup pass
  constexpr int UN = 8;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bh = blockIdx.x;
  const int b = bh / p.heads, h = bh - b * p.heads;
  const int hd = p.hd;
  const int kg = tid / LPK, ch = tid - kg * LPK;      // key slot within a pass, 8-wide column chunk
  const bool cok = ch * 8 < hd;
  const bf16* kb = p.k + b * p.k_bs + h * p.k_hs + ch * 8;
  const bf16* vb = p.v + b * p.v_bs + h * p.v_hs + ch * 8;
  // q is requested unconditionally (a chunk past head_dim re-reads chunk 0 and is zeroed) and first USED behind the first batch of
  // K / V requests: inside `if (cok)` the compiler waited for it before anything else went out -- one more round trip in a chain of five
  const bf16x8 q_raw = *(const bf16x8*)(p.q + b * p.q_bs + h * p.q_hs + (cok ? ch * 8 : 0));
  f32x8 qv;
  bool q_ready = false;
  const float sc = p.scale_q_bf16 ? 1.0f : p.scale;
  const bf16x8 zero8 = cvt8(f32x8{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f});
  // ---- scores.  The V rows of the first NPF chunks (NPF * 256 keys at head_dim 64: a caption's whole context) are requested in
  // the same breath as their K rows (round 4): the step is a chain of dependent round trips -- q, K, [softmax], V, store -- and the
  // V trip ran after the softmax's two workgroup barriers; now it flies under them.
  constexpr int NPF = 2;
  bf16x8 vpre[NPF][UN];
  float mx = -INFINITY;
  for (int j0 = 0; j0 < p.sk; j0 += UN * KPI) {
    bf16x8 kv[UN];
    const int chunk = j0 / (UN * KPI);
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int j = j0 + u * KPI + kg;
      kv[u] = (cok && j < p.sk) ? *(const bf16x8*)(kb + (long long)j * p.k_rs) : zero8;
    }
    // (after the K rows: the counted wait for K then leaves these in flight)
#pragma unroll
    for (int c = 0; c < NPF; ++c)
      if (c == chunk) {
#pragma unroll
        for (int u = 0; u < UN; ++u) {
          const int j = j0 + u * KPI + kg;
          vpre[c][u] = (cok && j < p.sk) ? *(const bf16x8*)(vb + (long long)j * p.v_rs) : zero8;
        }
      }
    if (!q_ready) {
      qv = cvt8(q_raw);
      if (p.scale_q_bf16 == 1) qv = cvt8(cvt8(qv * p.scale));
      if (!cok) qv = f32x8{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      q_ready = true;
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const f32x8 kf = cvt8(kv[u]);
      float d = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) d += qv[e] * kf[e];
#pragma unroll
      for (int o = LPK / 2; o > 0; o >>= 1) d += __shfl_xor(d, o, 64);
      const int j = j0 + u * KPI + kg;
      if (ch == 0 && j < p.sk) {
        ps[j] = d * sc;
        mx = fmaxf(mx, d * sc);
      }
    }
  }
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float l = 0.f;
  for (int j = tid; j < p.sk; j += 256) {
    const float e = __expf(ps[j] - mx);
    ps[j] = e;
    l += e;
  }
  l = wave_sum(l);
  if (lane == 0) red[wave] = l;
  __syncthreads();
  l = (red[0] + red[1]) + (red[2] + red[3]);
  // ---- output: thread (kg, ch) accumulates its 8 columns over keys kg, kg + KPI, ...; then the key slots are summed
  f32x8 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int j0 = 0; j0 < p.sk; j0 += UN * KPI) {
    bf16x8 vv[UN];
    float pj[UN];
    const int chunk = j0 / (UN * KPI);
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int j = j0 + u * KPI + kg;
      const bool ok = cok && j < p.sk;
      if (chunk == 0) vv[u] = vpre[0][u];
      else if (chunk == 1) vv[u] = vpre[1][u];
      else vv[u] = ok ? *(const bf16x8*)(vb + (long long)j * p.v_rs) : zero8;
      pj[u] = ok ? ps[j] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) acc += cvt8(vv[u]) * pj[u];
  }
#pragma unroll
  for (int o = LPK; o < 64; o <<= 1)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] += __shfl_xor(acc[e], o, 64);
  if (lane < LPK)
#pragma unroll
    for (int e = 0; e < 8; ++e) accs[wave][lane * 8 + e] = acc[e];
  __syncthreads();
  if (tid < LPK && tid * 8 < hd) {
    f32x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = ((accs[0][tid * 8 + e] + accs[1][tid * 8 + e]) + (accs[2][tid * 8 + e] + accs[3][tid * 8 + e])) * (1.0f / l);
    *(bf16x8*)(p.o + b * p.o_bs + h * p.o_hs + tid * 8) = cvt8(o);
  }
  if (tid == 0 && p.lse) p.lse[bh] = mx + __logf(l);
}

#include "attention_pair.inc"
#include "attention_duo.inc"

// waves per workgroup: cover a whole sequence with one workgroup when it has <= 256 rows (no idle
// waves: 160 rows -> 5 waves, 197 -> 7), otherwise 8 waves = 256 rows per workgroup
int waves_for(int rows) {
  const int w = (rows + 31) / 32;
  return w < 4 ? 4 : (w > 8 ? 8 : w);
}

int fill_args(AttnArgs& a, const mpv_attn_desc* d) {
  a.q = (const bf16*)d->q;
  a.k = (const bf16*)d->k;
  a.v = (const bf16*)d->v;
  a.o = (bf16*)d->o;
  a.lse = d->lse;
  a.q_bs = d->q_bs; a.q_hs = d->q_hs; a.q_rs = d->q_rs;
  a.k_bs = d->k_bs; a.k_hs = d->k_hs; a.k_rs = d->k_rs;
  a.v_bs = d->v_bs; a.v_hs = d->v_hs; a.v_rs = d->v_rs;
  a.o_bs = d->o_bs; a.o_hs = d->o_hs; a.o_rs = d->o_rs;
  a.batch = d->batch;
  a.heads = d->heads;
  a.sq = d->sq;
  a.sk = d->sk;
  a.hd = d->head_dim;
  a.causal = d->causal;
  a.scale = d->scale;
  a.scale_q_bf16 = d->scale_q_bf16;
  a.drop_thr = d->dropout_p > 0.f ? mpv_drop_threshold(d->dropout_p) : 0;
  a.drop_scale = d->dropout_p > 0.f ? 1.0f / (1.0f - d->dropout_p) : 1.0f;
  a.seed = d->seed;
  a.offset = d->offset;
  return 0;
}

int check_desc(const mpv_attn_desc* d, const char* who) {
  MPV_REQUIRE(d && d->q && d->k && d->v && d->o, MPV_E_ARG, "%s: null pointer", who);
  MPV_REQUIRE(d->batch > 0 && d->heads > 0 && d->sq > 0 && d->sk > 0, MPV_E_SHAPE, "%s: empty problem", who);
  MPV_REQUIRE(d->head_dim >= 8 && d->head_dim <= 96 && d->head_dim % 8 == 0, MPV_E_SHAPE,
              "%s: head_dim %d must be a multiple of 8 in [8, 96]", who, d->head_dim);
  MPV_REQUIRE(d->q_rs % 8 == 0 && d->k_rs % 8 == 0 && d->v_rs % 8 == 0 && d->o_rs % 4 == 0 && d->q_hs % 8 == 0 &&
                  d->k_hs % 8 == 0 && d->v_hs % 8 == 0 && d->o_hs % 4 == 0 && d->q_bs % 8 == 0 && d->k_bs % 8 == 0 &&
                  d->v_bs % 8 == 0 && d->o_bs % 4 == 0,
              MPV_E_ALIGN, "%s: strides must keep rows 16-byte aligned", who);
  MPV_REQUIRE((((uintptr_t)d->q | (uintptr_t)d->k | (uintptr_t)d->v) & 15) == 0 && ((uintptr_t)d->o & 7) == 0, MPV_E_ALIGN,
              "%s: q/k/v must be 16-byte aligned", who);
  MPV_REQUIRE(d->dropout_p >= 0.f && d->dropout_p < 1.f, MPV_E_ARG, "%s: bad dropout_p", who);
  MPV_REQUIRE(d->scale > 0.f, MPV_E_ARG, "%s: scale must be positive (the row maximum is taken before scaling)", who);
  MPV_REQUIRE(!d->causal || d->sk >= d->sq, MPV_E_SHAPE, "%s: causal needs sk >= sq", who);
  return MPV_OK;
}

}  // namespace

constexpr int RES_MAX_ROWS = 256;
// whole zero-filled 32-row tiles per region (one workgroup per CU is register-bound anyway: 90 KiB at S = 197 costs nothing)
static size_t res_region(int rows, int pitch) { return ((size_t)((rows + 31) / 32 * 32) * pitch + 1023) / 1024 * 1024; }
static size_t res_lds_bytes(int rows, bool stats) {
  const int padded = (rows + 31) / 32 * 32;
  return 2 * res_region(rows, ROWB) + (stats ? (2 * (size_t)padded * sizeof(float) + 1023) / 1024 * 1024 : 0);
}
template <typename K>
static void allow_lds(K kernel, int bytes = 120 * 1024) {
  (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}
// ---- persistent variants: host side ----
constexpr size_t PRES_LDS_MAX = 160 * 1024;
static size_t pres_region_h(int rows, int pitch) { return ((size_t)rows * pitch + 1023) / 1024 * 1024; }
static int pres_pitch_r(int hdc) { return hdc * 2 + 16; }
static int pres_pitch_c(int hdc) { return hdc > 80 ? hdc * 2 : hdc * 2 + 16; }
static int attn_ncu() {
  static int ncu = 0;
  if (!ncu) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0) ncu = 256;
    ncu = ncu / 8 * 8 > 0 ? ncu / 8 * 8 : 8;      // whole XCD groups: item i and workgroup i % grid then share i % 8
  }
  return ncu;
}
// MPV_ATTN_PERSIST (measurement knob, read once): 0 = the one-shot resident kernels only; 1 (default) = the persistent
// double-buffered kernels where a workgroup gets more than one (batch, head) item; 2 = also for a single item per workgroup
static int pres_mode() {
  static int mode = -1;
  if (mode < 0) {
    const char* e = getenv("MPV_ATTN_PERSIST");
    mode = e ? atoi(e) : 1;
  }
  return mode;
}
// the persistent kernels address every operand with 32-bit element offsets from its base pointer
static bool pres_offsets_ok(const mpv_attn_desc* d) {
  auto fits = [&](int64_t bs, int64_t hs, int64_t rs, int rows) {
    return bs >= 0 && hs >= 0 && rs > 0 && rs < (1 << 20) &&
           (int64_t)(d->batch + 7) * bs + (int64_t)d->heads * hs + (int64_t)(rows + 32) * rs + 128 < (1LL << 31) / 2;
  };
  return fits(d->q_bs, d->q_hs, d->q_rs, d->sq) && fits(d->k_bs, d->k_hs, d->k_rs, d->sk) && fits(d->v_bs, d->v_hs, d->v_rs, d->sk) &&
         fits(d->o_bs, d->o_hs, d->o_rs, d->sq) && (int64_t)(d->batch + 7) * d->heads * d->sq < (1LL << 30);
}
static bool pres_wanted(int nitems) { return pres_mode() >= 2 || (pres_mode() == 1 && nitems > attn_ncu()); }
static void pres_attr_once() {
  static bool done = false;
  if (done) return;
  const int L = (int)PRES_LDS_MAX;
  allow_lds(attn_fwd_pres_kernel<96, 7>, L); allow_lds(attn_fwd_pres_kernel<96, 7, 8, 512, 1, true>, L);
  done = true;
}
// Instances exist where the structure measured faster than the one-shot kernels and the item loop fits the register file
// without scratch (a scratch reload sits behind an s_waitcnt vmcnt(0), which would drain the next item's DMA in the middle of
// the current one): the 7-tile non-causal head_dim-96 forward (ViT-B/16 spatial attention).
static bool pres_fwd_has(int hdc, const mpv_attn_desc* d) {
  return hdc == 96 && !d->causal && d->dropout_p == 0.f && (d->sk + 31) / 32 == 7;
}

static void res_attr_once() {
  static bool done = false;
  if (done) return;
  allow_lds(attn_fwd_res_kernel<64>); allow_lds(attn_fwd_res_kernel<64, 0, 5, 320, 3>); allow_lds(attn_fwd_res_kernel<80>); allow_lds(attn_fwd_res_kernel<96>); allow_lds(attn_fwd_res_kernel<96, 7>);
  allow_lds(attn_bwd_dq_res_kernel<64>); allow_lds(attn_bwd_dq_res_kernel<64, 0, 320, 3>); allow_lds(attn_bwd_dq_res_kernel<80>); allow_lds(attn_bwd_dq_res_kernel<96>); allow_lds(attn_bwd_dq_res_kernel<96, 7>);
  allow_lds(attn_bwd_dkv_res_kernel<64, 512>); allow_lds(attn_bwd_dkv_res_kernel<64, 320, 3>); allow_lds(attn_bwd_dkv_res_kernel<80, 512>); allow_lds(attn_bwd_dkv_res_kernel<96, 512>);
  done = true;
}

// ---- paired-block causal kernels (attention_pair.inc): host side ----
// MPV_ATTN_PAIR (measurement knob, read once): 0 = the one-shot resident kernels; 1 (default) = the paired forward and dQ kernels, the
// paired dK/dV kernel above 5 blocks; 2 = the paired dK/dV kernel everywhere
static int pair_mode() {
  static int mode = -1;
  if (mode < 0) {
    const char* e = getenv("MPV_ATTN_PAIR");
    mode = e ? atoi(e) : 1;
  }
  return mode;
}
// MPV_ATTN_DUO (measurement knob, read once): 0 = the one-shot 7-wave backward kernels of the ViT shape; 1 = dQ as two 4-wave items
// per CU; 2 (default) = dQ and dK/dV
static int duo_mode() {
  static int mode = -1;
  if (mode < 0) {
    const char* e = getenv("MPV_ATTN_DUO");
    mode = e ? atoi(e) : 2;
  }
  return mode;
}
constexpr int PAIR_MAX_ROWS = 224;
static bool pair_ok(const mpv_attn_desc* d) {
  return pair_mode() > 0 && d->head_dim == 64 && d->causal && d->sq == d->sk && d->sk <= PAIR_MAX_ROWS;
}
static size_t pair_lds(int rows) { return 2 * (size_t)((rows + 31) / 32) * 32 * 128; }      // two images of whole 32-row tiles (z_region)

extern "C" int mpv_attn_fwd(const mpv_attn_desc* d, hipStream_t stream) {
  int rc = check_desc(d, "mpv_attn_fwd");
  if (rc) return rc;
  AttnArgs a = {};
  fill_args(a, d);
  if (d->sq == 1 && d->dropout_p == 0.f && d->sk <= DEC_MAX_SK && (d->o_hs % 8) == 0 && ((uintptr_t)d->o & 15) == 0) {      // decode step of KV-cache generation
    const dim3 dg((unsigned)(d->batch * d->heads));
    if (d->head_dim <= 64) hipLaunchKernelGGL(attn_decode_kernel<8>, dg, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(attn_decode_kernel<16>, dg, dim3(256), 0, stream, a);
    return mpv_check_launch("mpv_attn_fwd");
  }
  if (d->sk <= RES_MAX_ROWS) {
    res_attr_once();
    const int nw = waves_for(d->sq);
    const int gy = (d->batch + 7) / 8 * 8 * d->heads;   // decode_bh() needs whole groups of 8 sequences
    dim3 grid((d->sq + 32 * nw - 1) / (32 * nw), gy), block(64 * nw);
    if (pair_ok(d)) {
      const int nb = (d->sk + 31) / 32;
      const dim3 pg(1, gy), pb(64 * ((nb + 1) / 2));
      if (nb <= 5) hipLaunchKernelGGL((attn_fwd_pair64_kernel<5, 192, 3>), pg, pb, pair_lds(d->sk), stream, a);
      else hipLaunchKernelGGL((attn_fwd_pair64_kernel<7, 256, 2>), pg, pb, pair_lds(d->sk), stream, a);
      return mpv_check_launch("mpv_attn_fwd");
    }
    const int hdc = d->head_dim <= 64 ? 64 : d->head_dim <= 80 ? 80 : 96;
    const size_t plds = 2 * (pres_region_h(d->sk, pres_pitch_r(hdc)) + pres_region_h(d->sk, pres_pitch_c(hdc)));
    if (grid.x == 1 && plds <= PRES_LDS_MAX && pres_wanted(gy) && pres_offsets_ok(d) && pres_fwd_has(hdc, d)) {
      pres_attr_once();
      const dim3 pg((unsigned)(gy < attn_ncu() ? gy : attn_ncu()));
      if (d->sk <= 200) {
        hipLaunchKernelGGL((attn_fwd_pres_kernel<96, 7, 8, 512, 1, true>), pg, block, plds, stream, a, gy);   // ViT-B/16: 197 keys
      } else {
        hipLaunchKernelGGL((attn_fwd_pres_kernel<96, 7>), pg, block, plds, stream, a, gy);
      }
      return mpv_check_launch("mpv_attn_fwd");
    }
    const size_t lds = res_lds_bytes(d->sk, false);
    switch (d->head_dim <= 64 ? 64 : d->head_dim <= 80 ? 80 : 96) {
      case 64:
        if (d->sk <= 160 && nw <= 5) hipLaunchKernelGGL((attn_fwd_res_kernel<64, 0, 5, 320, 3>), grid, block, lds, stream, a);
        else hipLaunchKernelGGL((attn_fwd_res_kernel<64>), grid, block, lds, stream, a);
        break;
      case 80: hipLaunchKernelGGL((attn_fwd_res_kernel<80>), grid, block, lds, stream, a); break;
      default:
        if (!d->causal && d->dropout_p == 0.f && (d->sk + 31) / 32 == 7) hipLaunchKernelGGL((attn_fwd_res_kernel<96, 7>), grid, block, lds, stream, a);   // ViT-B/16 spatial: 197 keys
        else hipLaunchKernelGGL((attn_fwd_res_kernel<96>), grid, block, lds, stream, a);
        break;
    }
    return mpv_check_launch("mpv_attn_fwd");
  }
  const int nw = waves_for(d->sq);
  dim3 grid((d->sq + 32 * nw - 1) / (32 * nw), d->batch * d->heads), block(64 * nw);
  switch (d->head_dim <= 64 ? 64 : d->head_dim <= 80 ? 80 : 96) {
    case 64: hipLaunchKernelGGL((attn_fwd_kernel<64>), grid, block, 0, stream, a); break;
    case 80: hipLaunchKernelGGL((attn_fwd_kernel<80>), grid, block, 0, stream, a); break;
    default: hipLaunchKernelGGL((attn_fwd_kernel<96>), grid, block, 0, stream, a); break;
  }
  return mpv_check_launch("mpv_attn_fwd");
}

extern "C" int mpv_attn_bwd(const mpv_attn_desc* d, const void* dO, void* dq, void* dk, void* dv, float* delta,
                            hipStream_t stream) {
  int rc = check_desc(d, "mpv_attn_bwd");
  if (rc) return rc;
  MPV_REQUIRE(dO && dq && dk && dv && delta && d->lse, MPV_E_ARG, "mpv_attn_bwd: null pointer");
  AttnArgs a = {};
  fill_args(a, d);
  a.dO = (const bf16*)dO;
  a.dq = (bf16*)dq;
  a.dk = (bf16*)dk;
  a.dv = (bf16*)dv;
  a.delta = delta;
  // delta = rowsum(dO * O) is produced by the dQ kernel (row_delta) and read by the dK/dV kernel launched after it
  if (d->sk <= RES_MAX_ROWS && d->sq <= RES_MAX_ROWS) {
    res_attr_once();
    const int nw = waves_for(d->sq);
    const int gy = (d->batch + 7) / 8 * 8 * d->heads;
    const int nwk = waves_for(d->sk);
    dim3 gq((d->sq + 32 * nw - 1) / (32 * nw), gy), gk((d->sk + 32 * nwk - 1) / (32 * nwk), gy);
    const size_t lq = res_lds_bytes(d->sk, false), lk = res_lds_bytes(d->sq, true) + (d->head_dim > 64 ? res_region(d->sk, ROWB) : 0);
    const int hdc = d->head_dim <= 64 ? 64 : d->head_dim <= 80 ? 80 : 96;
    const bool small64 = hdc == 64 && d->sk <= 160 && d->sq <= 160 && nw <= 5 && nwk <= 5;     // two workgroups per CU (see attn_fwd_res_kernel)
    const bool vit7 = hdc == 96 && !d->causal && d->dropout_p == 0.f && (d->sk + 31) / 32 == 7;
    // dQ first: it also stores delta = rowsum(dO * O), which the dK/dV kernel reads
    if (pair_ok(d)) {      // paired-block dQ role (attention_pair.inc); the dK/dV role stays with the one-shot kernel below
      const int nb = (d->sk + 31) / 32;
      const dim3 pg(1, gy), pb(64 * ((nb + 1) / 2));
      if (nb <= 5) hipLaunchKernelGGL((attn_bwd_dq_pair64_kernel<192, 3>), pg, pb, pair_lds(d->sk), stream, a);
      else hipLaunchKernelGGL((attn_bwd_dq_pair64_kernel<256, 3>), pg, pb, pair_lds(d->sk), stream, a);
    } else {
      switch (hdc) {
        case 64:
          if (small64) hipLaunchKernelGGL((attn_bwd_dq_res_kernel<64, 0, 320, 3>), gq, dim3(64 * nw), lq, stream, a);
          else hipLaunchKernelGGL((attn_bwd_dq_res_kernel<64>), gq, dim3(64 * nw), lq, stream, a);
          break;
        case 80: hipLaunchKernelGGL((attn_bwd_dq_res_kernel<80>), gq, dim3(64 * nw), lq, stream, a); break;
        default:
          if (vit7 && duo_mode() && d->sq == d->sk && d->head_dim == 96) {      // two 4-wave items per CU (attention_duo.inc)
            static bool attr = false;
            if (!attr) {
              allow_lds(attn_bwd_dq_duo96_kernel<256>, 80 * 1024);
              attr = true;
            }
            const size_t dl = (((size_t)d->sk * 208 + 1023) / 1024 + ((size_t)d->sk * 192 + 1023) / 1024) * 1024;
            hipLaunchKernelGGL((attn_bwd_dq_duo96_kernel<256>), dim3(1, gy), dim3(256), dl, stream, a);
          } else if (vit7) hipLaunchKernelGGL((attn_bwd_dq_res_kernel<96, 7>), gq, dim3(64 * nw), lq, stream, a);
          else hipLaunchKernelGGL((attn_bwd_dq_res_kernel<96>), gq, dim3(64 * nw), lq, stream, a);
          break;
      }
    }
    // the dK/dV role paired too where it measured faster: above 5 blocks (S = 208: 4 waves and two items per CU instead of one
    // 7-wave workgroup, backward 120.8 -> 99.4 us per layer); at S = 160 (three items per CU instead of two 5-wave ones) it is a
    // tie, 76.4 vs 75.7 us, and the one-shot kernel stays (MPV_ATTN_PAIR=2 forces the paired one) -- profiles/r03_c19_dkv_pair_ab.log
    if (pair_ok(d) && (pair_mode() >= 2 || (d->sk + 31) / 32 >= 6)) {
      const int nb = (d->sk + 31) / 32;
      const size_t pl = pair_lds(d->sk) + ((size_t)(2 * nb * 32 * 4) + 1023) / 1024 * 1024;
      const dim3 pg(1, gy), pb(64 * ((nb + 1) / 2));
      if (nb <= 5) hipLaunchKernelGGL((attn_bwd_dkv_pair64_kernel<192, 3>), pg, pb, pl, stream, a);
      else hipLaunchKernelGGL((attn_bwd_dkv_pair64_kernel<256, 3>), pg, pb, pl, stream, a);
    } else {
      switch (hdc) {
        case 64:
          if (small64) hipLaunchKernelGGL((attn_bwd_dkv_res_kernel<64, 320, 3>), gk, dim3(64 * nwk), lk, stream, a);
          else hipLaunchKernelGGL((attn_bwd_dkv_res_kernel<64, 512>), gk, dim3(64 * nwk), lk, stream, a);
          break;
        case 80: hipLaunchKernelGGL((attn_bwd_dkv_res_kernel<80, 512>), gk, dim3(64 * nwk), lk, stream, a); break;
        default:
          if (vit7 && duo_mode() >= 2 && d->sq == d->sk && d->head_dim == 96) {      // two 4-wave items per CU (attention_duo.inc)
            static bool attr = false;
            if (!attr) {
              allow_lds(attn_bwd_dkv_duo96_kernel<256>, 80 * 1024);
              attr = true;
            }
            const int qr = (d->sq + 31) / 32 * 32;
            const size_t dl = ((size_t)(2 * qr * 4 + 1023) / 1024 + 2 * (((size_t)d->sq * 192 + 1023) / 1024)) * 1024;
            hipLaunchKernelGGL((attn_bwd_dkv_duo96_kernel<256>), dim3(1, gy), dim3(256), dl, stream, a);
          } else hipLaunchKernelGGL((attn_bwd_dkv_res_kernel<96, 512>), gk, dim3(64 * nwk), lk, stream, a);
          break;
      }
    }
    return mpv_check_launch("mpv_attn_bwd");
  }
  const int nwq = waves_for(d->sq), nwk = 4;   // dK/dV keeps 4 waves (its accumulators need > 256 VGPRs at 8)
  dim3 gq((d->sq + 32 * nwq - 1) / (32 * nwq), d->batch * d->heads), gk((d->sk + 32 * nwk - 1) / (32 * nwk), d->batch * d->heads);
  dim3 bq(64 * nwq), bk(64 * nwk);
  switch (d->head_dim <= 64 ? 64 : d->head_dim <= 80 ? 80 : 96) {
    case 64:
      hipLaunchKernelGGL((attn_bwd_dq_kernel<64>), gq, bq, 0, stream, a);
      hipLaunchKernelGGL((attn_bwd_dkv_kernel<64>), gk, bk, 0, stream, a);
      break;
    case 80:
      hipLaunchKernelGGL((attn_bwd_dq_kernel<80>), gq, bq, 0, stream, a);
      hipLaunchKernelGGL((attn_bwd_dkv_kernel<80>), gk, bk, 0, stream, a);
      break;
    default:
      hipLaunchKernelGGL((attn_bwd_dq_kernel<96>), gq, bq, 0, stream, a);
      hipLaunchKernelGGL((attn_bwd_dkv_kernel<96>), gk, bk, 0, stream, a);
      break;
  }
  return mpv_check_launch("mpv_attn_bwd");
}

// ---- the same problem with its rows in LDS as bf16 (round 4; measured stand-alone in round 3: tools/probe/temporal_bf16_probe.hip,
// profiles/r03_c26_temporal_bf16_probe.log).  The rows ARE bf16 values (q * scale is rounded to bf16 by the reference itself, :179),
// so bf16 rows at the conflict-free 208-byte pitch hold the same numbers in half the LDS bytes -- LDS is what limits this kernel's
// occupancy (a wave's fp32 rows are 10 / 13 KiB at 8 frames, 20 / 28 KiB at 16: 7 / 5 waves per CU there) -- and the two dot-product
// phases (S = q k^T, dP = dO v^T) run on v_dot2c_f32_bf16 (two MACs per instruction, fp32 accumulate, no widening); the p-weighted row
// sums widen their operand on the fly.  Outputs differ from the fp32-row kernel by the summation order only (2-4e-3 of the largest
// element).  8 frames: forward -8 %, backward -4 %; 16 frames: -19 % / -34 %.  Compile-time instances only.
__device__ __forceinline__ void tdot8(bf16x8 a, bf16x8 b, float& c0, float& c1) {
  c0 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(a, a, 0, 1), __builtin_shufflevector(b, b, 0, 1), c0, false);
  c1 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(a, a, 2, 3), __builtin_shufflevector(b, b, 2, 3), c1, false);
  c0 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(a, a, 4, 5), __builtin_shufflevector(b, b, 4, 5), c0, false);
  c1 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(a, a, 6, 7), __builtin_shufflevector(b, b, 6, 7), c1, false);
}

template <bool BWD, int T, int HD>
constexpr int temporal_b16_wave_lds() {
  return ((BWD ? 4 : 3) * T * (HD + 8) * 2 + (BWD ? 2 : 1) * T * (T + 1) * 4 + 15) & ~15;
}

template <bool BWD, int T, int HD>
__global__ __launch_bounds__(256) void temporal_attn_b16_kernel(const TempArgs p) {
  extern __shared__ __attribute__((aligned(16))) float tsm[];
  constexpr int LDB = HD + 8;                          // bf16 row pitch: 208 bytes at head_dim 96
  constexpr int H4 = HD / 4, H8 = HD / 8;
  constexpr int ROWS = (BWD ? 4 : 3) * T * LDB * 2;    // bytes of the bf16 images
  constexpr int PER_WAVE = temporal_b16_wave_lds<BWD, T, HD>();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwv = (int)(blockDim.x >> 6);
  const int D = p.heads * HD;
  char* base = (char*)tsm + wave * PER_WAVE;
  bf16* qs = (bf16*)base;
  bf16* ks = qs + T * LDB;
  bf16* vs = ks + T * LDB;
  bf16* dos = vs + T * LDB;                            // BWD only
  float* ps = (float*)(base + ROWS);                   // [T][T+1] probabilities
  float* dss = ps + T * (T + 1);                       // BWD only: dS
  const long long nprob = (long long)p.n_outer * p.n_inner * p.heads;
  const long long pstride = (long long)gridDim.x * nwv;
  constexpr int NX = (T * H4 + 63) / 64;
  bf16x4 rq[NX], rk[NX], rv[NX], rd[BWD ? NX : 1];
  auto row_of = [&](long long pr, int& h) {
    h = (int)(pr % p.heads);
    const long long seq = pr / p.heads;
    const long long o = seq / p.n_inner, i = seq % p.n_inner;
    return o * p.outer_stride + p.inner_offset + i;
  };
  auto request = [&](long long pr) {
    int h;
    const long long row0 = row_of(pr, h);
#pragma unroll
    for (int j = 0; j < NX; ++j) {
      const int x = lane + 64 * j;
      if (x < T * H4) {
        const int t = x / H4, c4 = x - t * H4;
        const bf16* src = p.qkv + (row0 + t * p.t_stride) * (3LL * D) + h * HD + c4 * 4;
        rq[j] = *(const bf16x4*)src;
        rk[j] = *(const bf16x4*)(src + D);
        rv[j] = *(const bf16x4*)(src + 2 * D);
        if constexpr (BWD) rd[j] = *(const bf16x4*)(p.dout + (row0 + t * p.t_stride) * (long long)D + h * HD + c4 * 4);
      }
    }
  };
  long long pr0 = (long long)blockIdx.x * nwv + wave;
  if (pr0 < nprob) request(pr0);
  for (long long pr = pr0; pr < nprob; pr += pstride) {
    int h;
    const long long row0 = row_of(pr, h);
#pragma unroll
    for (int j = 0; j < NX; ++j) {
      const int x = lane + 64 * j;
      if (x < T * H4) {
        const int t = x / H4, c4 = x - t * H4;
        *(bf16x4*)(qs + t * LDB + c4 * 4) = cvt4(cvt4(rq[j]) * p.scale);      // q * scale rounds to bf16 (reference :179)
        *(bf16x4*)(ks + t * LDB + c4 * 4) = rk[j];
        *(bf16x4*)(vs + t * LDB + c4 * 4) = rv[j];
        if constexpr (BWD) *(bf16x4*)(dos + t * LDB + c4 * 4) = rd[j];
      }
    }
    if (pr + pstride < nprob) request(pr + pstride);   // the next problem's operands travel while this one is computed
    WAVE_SYNC();
#pragma unroll
    for (int x = lane; x < T * T; x += 64) {
      const int a = x / T, bb = x - a * T;
      float s0 = 0.f, s1 = 0.f, d0 = 0.f, d1 = 0.f;
#pragma unroll
      for (int c8 = 0; c8 < H8; ++c8) {
        tdot8(*(const bf16x8*)(qs + a * LDB + c8 * 8), *(const bf16x8*)(ks + bb * LDB + c8 * 8), s0, s1);
        if constexpr (BWD) tdot8(*(const bf16x8*)(dos + a * LDB + c8 * 8), *(const bf16x8*)(vs + bb * LDB + c8 * 8), d0, d1);
      }
      ps[a * (T + 1) + bb] = s0 + s1;
      if constexpr (BWD) dss[a * (T + 1) + bb] = d0 + d1;                  // dP = dO V^T
    }
    WAVE_SYNC();
    if (lane < T) {
      float mx = -INFINITY;
#pragma unroll
      for (int j = 0; j < T; ++j) mx = fmaxf(mx, ps[lane * (T + 1) + j]);
      float sum = 0.f;
#pragma unroll
      for (int j = 0; j < T; ++j) sum += __expf(ps[lane * (T + 1) + j] - mx);
      const float inv = 1.0f / sum;
      float dl = 0.f;
#pragma unroll
      for (int j = 0; j < T; ++j) {
        const float pv = __expf(ps[lane * (T + 1) + j] - mx) * inv;
        ps[lane * (T + 1) + j] = BWD ? pv : bf2f(f2bf(pv));              // forward: probabilities cast to bf16 (:201)
        if constexpr (BWD) dl += pv * dss[lane * (T + 1) + j];
      }
      if constexpr (BWD)                                                   // dS = P * (dP - rowsum(P*dP))
#pragma unroll
        for (int j = 0; j < T; ++j) dss[lane * (T + 1) + j] = ps[lane * (T + 1) + j] * (dss[lane * (T + 1) + j] - dl);
    }
    WAVE_SYNC();
#pragma unroll
    for (int x = lane; x < T * H4; x += 64) {
      const int a = x / H4, c4 = x - a * H4;
      if constexpr (!BWD) {
        f32x4 o4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < T; ++j) o4 += cvt4(*(const bf16x4*)(vs + j * LDB + c4 * 4)) * ps[a * (T + 1) + j];
        *(bf16x4*)(p.out + (row0 + a * p.t_stride) * (long long)D + h * HD + c4 * 4) = cvt4(o4);
      } else {
        f32x4 dq = {0.f, 0.f, 0.f, 0.f}, dk = dq, dv = dq;
#pragma unroll
        for (int j = 0; j < T; ++j) {
          dq += cvt4(*(const bf16x4*)(ks + j * LDB + c4 * 4)) * dss[a * (T + 1) + j];
          dk += cvt4(*(const bf16x4*)(qs + j * LDB + c4 * 4)) * dss[j * (T + 1) + a];
          dv += cvt4(*(const bf16x4*)(dos + j * LDB + c4 * 4)) * ps[j * (T + 1) + a];
        }
        bf16* dst = p.dqkv + (row0 + a * p.t_stride) * (3LL * D) + h * HD + c4 * 4;
        *(bf16x4*)dst = cvt4(dq * p.scale);
        *(bf16x4*)(dst + D) = cvt4(dk);
        *(bf16x4*)(dst + 2 * D) = cvt4(dv);
      }
    }
    WAVE_SYNC();
  }
}

static int temporal_common(TempArgs& t, int n_outer, int64_t outer_stride, int n_inner, int64_t inner_offset,
                           int64_t t_stride, int T, int heads, int head_dim, float scale, const char* who) {
  MPV_REQUIRE(T >= 1 && T <= 16, MPV_E_SHAPE, "%s: T=%d must be in [1,16]", who, T);
  MPV_REQUIRE(head_dim % 4 == 0 && head_dim <= 96, MPV_E_SHAPE, "%s: head_dim=%d must be a multiple of 4 and <= 96", who, head_dim);
  MPV_REQUIRE(n_outer > 0 && n_inner > 0 && heads > 0, MPV_E_SHAPE, "%s: empty problem", who);
  t.n_outer = n_outer;
  t.n_inner = n_inner;
  t.outer_stride = outer_stride;
  t.inner_offset = inner_offset;
  t.t_stride = t_stride;
  t.T = T;
  t.heads = heads;
  t.hd = head_dim;
  t.scale = scale;
  return MPV_OK;
}

// Waves (one problem each) per workgroup: 4 unless fewer waves per workgroup put more waves on a CU (LDS is what limits the
// occupancy of this kernel: 160 KiB per CU, a wave's rows are 10 KiB at 8 frames -- 4 x 4 waves either way -- but 20 / 28 KiB at
// 16 frames, where 4-wave workgroups would leave one workgroup = 4 waves per CU and 1- / 2-wave ones fit 7 / 5).
static int temporal_waves_per_workgroup(size_t wave_lds) {
  const size_t cu = 160 * 1024;
  int best = 4, best_waves = (int)(cu / (4 * wave_lds)) * 4;
  for (int n = 3; n >= 1; --n) {
    const int waves = (int)(cu / (n * wave_lds)) * n;
    if (waves > best_waves) best = n, best_waves = waves;
  }
  return best;
}
static void temporal_set_attributes() {
  static bool done = false;
  if (done) return;
  const int lim = 150 * 1024;
  (void)hipFuncSetAttribute((const void*)temporal_attn_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lim);
  (void)hipFuncSetAttribute((const void*)temporal_attn_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lim);
  (void)hipFuncSetAttribute((const void*)temporal_attn_kernel<false, 8, 96>, hipFuncAttributeMaxDynamicSharedMemorySize, lim);
  (void)hipFuncSetAttribute((const void*)temporal_attn_kernel<true, 8, 96>, hipFuncAttributeMaxDynamicSharedMemorySize, lim);
  (void)hipFuncSetAttribute((const void*)temporal_attn_kernel<false, 16, 96>, hipFuncAttributeMaxDynamicSharedMemorySize, lim);
  (void)hipFuncSetAttribute((const void*)temporal_attn_kernel<true, 16, 96>, hipFuncAttributeMaxDynamicSharedMemorySize, lim);
  (void)hipFuncSetAttribute((const void*)temporal_attn_kernel<false, 4, 96>, hipFuncAttributeMaxDynamicSharedMemorySize, lim);
  (void)hipFuncSetAttribute((const void*)temporal_attn_kernel<true, 4, 96>, hipFuncAttributeMaxDynamicSharedMemorySize, lim);
  done = true;
}

// bf16-row instances (head_dim 96; 4 / 8 / 16 frames): the default; MPV_TEMPORAL_ROWS=fp32 keeps the fp32-row kernel (same-box A/B).
static bool temporal_rows_bf16() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("MPV_TEMPORAL_ROWS");
    v = (e && !strcmp(e, "fp32")) ? 0 : 1;
  }
  return v == 1;
}
template <bool BWD, int T>
static void temporal_b16_launch(const TempArgs& t, hipStream_t stream) {
  constexpr int NWV = 2;                               // measured best (or tied) at 8 and at 16 frames in both directions
  constexpr size_t lds = (size_t)NWV * temporal_b16_wave_lds<BWD, T, 96>();
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)temporal_attn_b16_kernel<BWD, T, 96>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    attr = true;
  }
  const long long nprob = (long long)t.n_outer * t.n_inner * t.heads;
  const int grid = (int)((nprob + NWV - 1) / NWV < 16384 ? (nprob + NWV - 1) / NWV : 16384);
  hipLaunchKernelGGL((temporal_attn_b16_kernel<BWD, T, 96>), dim3(grid), dim3(64 * NWV), lds, stream, t);
}
template <bool BWD>
static bool temporal_b16_dispatch(const TempArgs& t, hipStream_t stream) {
  if (t.hd != 96 || !temporal_rows_bf16()) return false;
  if (t.T == 8) temporal_b16_launch<BWD, 8>(t, stream);
  else if (t.T == 16) temporal_b16_launch<BWD, 16>(t, stream);
  else if (t.T == 4) temporal_b16_launch<BWD, 4>(t, stream);
  else return false;
  return true;
}

extern "C" int mpv_temporal_attn_fwd(const void* qkv, void* out, int n_outer, int64_t outer_stride, int n_inner,
                                     int64_t inner_offset, int64_t t_stride, int T, int heads, int head_dim,
                                     float scale, hipStream_t stream) {
  MPV_REQUIRE(qkv && out, MPV_E_ARG, "mpv_temporal_attn_fwd: null pointer");
  TempArgs t = {};
  int rc = temporal_common(t, n_outer, outer_stride, n_inner, inner_offset, t_stride, T, heads, head_dim, scale,
                           "mpv_temporal_attn_fwd");
  if (rc) return rc;
  t.qkv = (const bf16*)qkv;
  t.out = (bf16*)out;
  if (temporal_b16_dispatch<false>(t, stream)) return mpv_check_launch("mpv_temporal_attn_fwd");
  const size_t wave_lds = sizeof(float) * (size_t)(((3 * T * (head_dim + 4) + T * (T + 1)) + 3) & ~3);
  const int nwv = temporal_waves_per_workgroup(wave_lds);
  const size_t lds = nwv * wave_lds;
  const long long nprob = (long long)n_outer * n_inner * heads;
  const int grid = (int)((nprob + nwv - 1) / nwv < 16384 ? (nprob + nwv - 1) / nwv : 16384);
  temporal_set_attributes();
  if (T == 8 && head_dim == 96) hipLaunchKernelGGL((temporal_attn_kernel<false, 8, 96>), dim3(grid), dim3(64 * nwv), lds, stream, t);
  else if (T == 16 && head_dim == 96) hipLaunchKernelGGL((temporal_attn_kernel<false, 16, 96>), dim3(grid), dim3(64 * nwv), lds, stream, t);
  else if (T == 4 && head_dim == 96) hipLaunchKernelGGL((temporal_attn_kernel<false, 4, 96>), dim3(grid), dim3(64 * nwv), lds, stream, t);
  else hipLaunchKernelGGL((temporal_attn_kernel<false>), dim3(grid), dim3(64 * nwv), lds, stream, t);
  return mpv_check_launch("mpv_temporal_attn_fwd");
}

extern "C" int mpv_temporal_attn_bwd(const void* qkv, const void* dout, void* dqkv, int n_outer, int64_t outer_stride,
                                     int n_inner, int64_t inner_offset, int64_t t_stride, int T, int heads,
                                     int head_dim, float scale, hipStream_t stream) {
  MPV_REQUIRE(qkv && dout && dqkv, MPV_E_ARG, "mpv_temporal_attn_bwd: null pointer");
  TempArgs t = {};
  int rc = temporal_common(t, n_outer, outer_stride, n_inner, inner_offset, t_stride, T, heads, head_dim, scale,
                           "mpv_temporal_attn_bwd");
  if (rc) return rc;
  t.qkv = (const bf16*)qkv;
  t.dout = (const bf16*)dout;
  t.dqkv = (bf16*)dqkv;
  if (temporal_b16_dispatch<true>(t, stream)) return mpv_check_launch("mpv_temporal_attn_bwd");
  const size_t wave_lds = sizeof(float) * (size_t)(((4 * T * (head_dim + 4) + 2 * T * (T + 1)) + 3) & ~3);
  const int nwv = temporal_waves_per_workgroup(wave_lds);
  const size_t lds = nwv * wave_lds;
  const long long nprob = (long long)n_outer * n_inner * heads;
  const int grid = (int)((nprob + nwv - 1) / nwv < 16384 ? (nprob + nwv - 1) / nwv : 16384);
  temporal_set_attributes();
  if (T == 8 && head_dim == 96) hipLaunchKernelGGL((temporal_attn_kernel<true, 8, 96>), dim3(grid), dim3(64 * nwv), lds, stream, t);
  else if (T == 16 && head_dim == 96) hipLaunchKernelGGL((temporal_attn_kernel<true, 16, 96>), dim3(grid), dim3(64 * nwv), lds, stream, t);
  else if (T == 4 && head_dim == 96) hipLaunchKernelGGL((temporal_attn_kernel<true, 4, 96>), dim3(grid), dim3(64 * nwv), lds, stream, t);
  else hipLaunchKernelGGL((temporal_attn_kernel<true>), dim3(grid), dim3(64 * nwv), lds, stream, t);
  return mpv_check_launch("mpv_temporal_attn_bwd");
}

#ifdef MPV_ATTN_TIMING
extern "C" int mpv_attn_read_timeline(long long* host_out /* [4096*8] */) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(mpv_attn_dbg), sizeof(long long) * 4096 * 8) == hipSuccess ? 0 : -4;
}
#endif
