// norm.hip -- fused LayerNorm forward/backward for gfx950.  HBM-bound: one wave owns a row,
// the row lives in registers (8-byte bf16x4 loads, lanes interleaved over 256-element
// stripes so every wave-level load is a contiguous 512-byte burst), statistics in fp32 via
// wave shuffles, affine fused, bf16 out.  Backward fuses the residual-gradient add, an
// optional dropout-masked copy (for the bias_dropout_add that precedes the LN in the GPT
// layer) and per-block partial dgamma/dbeta sums (finalised by a second tiny kernel).
#include <cstdlib>
#include "mpv_common.h"
#include "../../include/mpv.h"

namespace {

struct LnFwdArgs {
  const bf16* x;
  const bf16* gamma;
  const bf16* beta;
  bf16* y;
  float* mean;
  float* rstd;
  long long rows;
  int cols;
  long long ldx, ldy;
  float eps;
  RowMap xmap, ymap;
};

template <int MAXC>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const LnFwdArgs p) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int nchunk = p.cols >> 2;
  const float inv_n = 1.0f / (float)p.cols;
  for (long long r = (long long)blockIdx.x * 4 + wave; r < p.rows; r += (long long)gridDim.x * 4) {
    const bf16* xr = p.x + map_row(p.xmap, r) * p.ldx;
    f32x4 v[MAXC];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int c = lane + 64 * i;
      if (c < nchunk) {
        v[i] = cvt4(*(const bf16x4*)(xr + c * 4));
        s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
      } else {
        v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    const float mu = wave_sum(s) * inv_n;
    float s2 = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int c = lane + 64 * i;
      if (c < nchunk) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float d = v[i][e] - mu;
          s2 += d * d;
        }
      }
    }
    const float rs = rsqrtf(wave_sum(s2) * inv_n + p.eps);
    bf16* yr = p.y + map_row(p.ymap, r) * p.ldy;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int c = lane + 64 * i;
      if (c < nchunk) {
        const f32x4 g = cvt4(*(const bf16x4*)(p.gamma + c * 4));
        const f32x4 b = cvt4(*(const bf16x4*)(p.beta + c * 4));
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mu) * rs * g[e] + b[e];
        *(bf16x4*)(yr + c * 4) = cvt4(o);
      }
    }
    if (lane == 0) {
      if (p.mean) p.mean[r] = mu;
      if (p.rstd) p.rstd[r] = rs;
    }
  }
}

struct LnBwdArgs {
  const bf16* dy;
  const bf16* x;
  const bf16* gamma;
  const float* mean;
  const float* rstd;
  const bf16* dres;
  bf16* dx;
  bf16* dx_drop;
  float drop_scale;
  uint32_t drop_thr;
  uint64_t seed, offset;
  float* part;  // [gridDim.x][2][cols] fp32 (dgamma, dbeta) or NULL
  long long rows;
  int cols;
  long long ldx, ldy;
  RowMap xmap, ymap;
};

template <int MAXC, bool DPARAM>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const LnBwdArgs p) {
  const uint64_t seed_r = p.drop_thr ? mpv_resolve_seed(p.seed) : 0;      // (bit 63 set: the seed lives in device memory, mpv_common.h)
  __shared__ float red[DPARAM ? 2 * MAXC * 256 : 1];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int nchunk = p.cols >> 2;
  const float inv_n = 1.0f / (float)p.cols;
  f32x4 gacc[DPARAM ? MAXC : 1], bacc[DPARAM ? MAXC : 1];
  if constexpr (DPARAM) {
#pragma unroll
    for (int i = 0; i < MAXC; ++i) gacc[i] = bacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  for (long long r = (long long)blockIdx.x * 4 + wave; r < p.rows; r += (long long)gridDim.x * 4) {
    const long long xrow = map_row(p.xmap, r);
    const bf16* xr = p.x + xrow * p.ldx;
    const bf16* dyr = p.dy + map_row(p.ymap, r) * p.ldy;
    const float mu = p.mean[r], rs = p.rstd[r];
    f32x4 xh[MAXC], g[MAXC];
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int c = lane + 64 * i;
      if (c < nchunk) {
        const f32x4 xv = cvt4(*(const bf16x4*)(xr + c * 4));
        const f32x4 dv = cvt4(*(const bf16x4*)(dyr + c * 4));
        const f32x4 gm = cvt4(*(const bf16x4*)(p.gamma + c * 4));
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          xh[i][e] = (xv[e] - mu) * rs;
          g[i][e] = dv[e] * gm[e];
          c1 += g[i][e];
          c2 += g[i][e] * xh[i][e];
          if constexpr (DPARAM) {
            gacc[i][e] += dv[e] * xh[i][e];
            bacc[i][e] += dv[e];
          }
        }
      } else {
        xh[i] = g[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    c1 = wave_sum(c1) * inv_n;
    c2 = wave_sum(c2) * inv_n;
    bf16* dxr = p.dx + xrow * p.ldx;
    const bf16* drr = p.dres ? p.dres + xrow * p.ldx : nullptr;
    bf16* ddr = p.dx_drop ? p.dx_drop + xrow * p.ldx : nullptr;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int c = lane + 64 * i;
      if (c < nchunk) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = rs * (g[i][e] - c1 - xh[i][e] * c2);
        if (drr) o += cvt4(*(const bf16x4*)(drr + c * 4));
        const bf16x4 ob = cvt4(o);
        *(bf16x4*)(dxr + c * 4) = ob;
        if (ddr) {
          const f32x4 of = cvt4(ob);
          const uint64_t base = p.offset + (uint64_t)r * (uint64_t)p.cols + (uint64_t)(c * 4);
          const f32x4 od = p.drop_thr ? mpv_dropout_vec<f32x4, 4>(of, seed_r, base, p.drop_thr, p.drop_scale) : of * p.drop_scale;
          *(bf16x4*)(ddr + c * 4) = cvt4(od);
        }
      }
    }
  }
  if constexpr (DPARAM) {
    // block reduce of the 4 waves' partial sums (one wave at a time into one LDS image)
#pragma unroll 1
    for (int w = 0; w < 4; ++w) {
      if (wave == w) {
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
          const int c = lane + 64 * i;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (w == 0) {
              red[c * 4 + e] = gacc[i][e];
              red[MAXC * 256 + c * 4 + e] = bacc[i][e];
            } else {
              red[c * 4 + e] += gacc[i][e];
              red[MAXC * 256 + c * 4 + e] += bacc[i][e];
            }
          }
        }
      }
      __syncthreads();
    }
    float* out = p.part + (long long)blockIdx.x * 2 * p.cols;
    for (int c = threadIdx.x; c < p.cols; c += 256) {
      out[c] = red[c];
      out[p.cols + c] = red[MAXC * 256 + c];
    }
  }
}


// ---- 16-byte variants (cols % 8 == 0: every shape of the path).  Same arithmetic and reduction order per element pair as
// the 8-byte kernels above; a lane owns 8 consecutive columns per 512-column stripe, so every wave-level access is a
// contiguous 1 KiB burst (the 8-byte form reaches ~3.4-4.4 TB/s, a 16-byte streaming kernel ~5 TB/s on this chip).
template <int MAXC>
__global__ __launch_bounds__(256) void ln_fwd8_kernel(const LnFwdArgs p) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // wave-uniform: row pointers live in SGPRs
  const int nchunk = p.cols >> 3;
  const float inv_n = 1.0f / (float)p.cols;
  // gamma / beta stay in registers as bf16 (they were re-read from cache after the reductions of every row), and the next
  // row of the wave is requested before the current one is reduced: its HBM latency runs under the two wave reductions
  // and the stores instead of after them.
  bf16x8 gmb[MAXC], btb[MAXC], nx[MAXC];
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int c = lane + 64 * i;
    gmb[i] = c < nchunk ? *(const bf16x8*)(p.gamma + c * 8) : bf16x8{};
    btb[i] = c < nchunk ? *(const bf16x8*)(p.beta + c * 8) : bf16x8{};
  }
  const long long stride = (long long)gridDim.x * 4;
  long long r = (long long)blockIdx.x * 4 + wave;
  if (r < p.rows) {
    const bf16* xr = p.x + map_row(p.xmap, r) * p.ldx;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) nx[i] = lane + 64 * i < nchunk ? *(const bf16x8*)(xr + (lane + 64 * i) * 8) : bf16x8{};
  }
  for (; r < p.rows; r += stride) {
    bf16x8 cur[MAXC];      // the row as raw bf16 (4 VGPRs per chunk); converted again for each sweep instead of 8 fp32 VGPRs kept live
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      cur[i] = nx[i];
      const f32x8 v = cvt8(cur[i]);          // chunks past the row are zeros: they add nothing to the sums
#pragma unroll
      for (int e = 0; e < 8; ++e) s += v[e];
    }
    if (r + stride < p.rows) {
      const bf16* xn = p.x + map_row(p.xmap, r + stride) * p.ldx;
#pragma unroll
      for (int i = 0; i < MAXC; ++i) nx[i] = lane + 64 * i < nchunk ? *(const bf16x8*)(xn + (lane + 64 * i) * 8) : bf16x8{};
    }
    const float mu = wave_sum(s) * inv_n;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) asm volatile("" : "+v"(cur[i]));
    float s2 = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int c = lane + 64 * i;
      if (c < nchunk) {
        const f32x8 v = cvt8(cur[i]);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float d = v[e] - mu;
          s2 += d * d;
        }
      }
    }
    const float rs = rsqrtf(wave_sum(s2) * inv_n + p.eps);
#pragma unroll
    for (int i = 0; i < MAXC; ++i) asm volatile("" : "+v"(cur[i]));
    bf16* yr = p.y + map_row(p.ymap, r) * p.ldy;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int c = lane + 64 * i;
      if (c < nchunk) {
        const f32x8 v = cvt8(cur[i]);
        const f32x8 g = cvt8(gmb[i]);
        const f32x8 b = cvt8(btb[i]);
        f32x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (v[e] - mu) * rs * g[e] + b[e];
        *(bf16x8*)(yr + c * 8) = cvt8(o);
      }
    }
    if (lane == 0) {
      if (p.mean) p.mean[r] = mu;
      if (p.rstd) p.rstd[r] = rs;
    }
  }
}

// No parameter gradients (the frozen decoder's LayerNorms, one row per wave at 5120 x 2048): the residual gradient is read after
// the reductions, interleaved with the stores -- measured 19.9 us against 23.0 us for the single-round-trip form below on that
// shape (5.3 TB/s: reads and writes alternating suit the HBM better than a read burst followed by a write burst).
// XF32: x is the decoder's fp32 residual stream (mpv_ln_stream_bwd); dy / dres / dx stay bf16.
template <int MAXC, bool XF32 = false>
__global__ __launch_bounds__(256, MAXC <= 4 ? 3 : MAXC <= 5 ? 2 : 1) void ln_bwd8_plain_kernel(const LnBwdArgs p) {
  // Round 4: every access of a row is a branch-free buffer access over a descriptor of exactly the row (chunks past the row read
  // zeros / store nothing; an absent residual gradient or dropout output is a descriptor of zero bytes).  With `c < nchunk ? load : 0`
  // in a branch the compiler could not count on a request having been issued and waited with vmcnt(0) behind every one of them:
  // the residual-gradient chunks came in one at a time, each behind the previous chunk's store acknowledgement.
  const uint64_t seed_r = p.drop_thr ? mpv_resolve_seed(p.seed) : 0;      // (bit 63 set: the seed lives in device memory, mpv_common.h)
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nchunk = p.cols >> 3;
  const float inv_n = 1.0f / (float)p.cols;
  union Ld { i32x4 i; bf16x8 b; f32x4 f; };
  const uint32_t rbytes = (uint32_t)p.cols * 2u;
  const __amdgpu_buffer_rsrc_t gs = make_rsrc(p.gamma, rbytes);
  for (long long r = (long long)blockIdx.x * 4 + wave; r < p.rows; r += (long long)gridDim.x * 4) {
    const long long xrow = map_row(p.xmap, r);
    const float mu = p.mean[r], rs = p.rstd[r];
    const __amdgpu_buffer_rsrc_t xs = XF32 ? make_rsrc((const float*)p.x + xrow * p.ldx, rbytes * 2u) : make_rsrc(p.x + xrow * p.ldx, rbytes);
    const __amdgpu_buffer_rsrc_t ds = make_rsrc(p.dy + map_row(p.ymap, r) * p.ldy, rbytes);
    f32x8 xh[MAXC], g[MAXC];
    Ld xa[MAXC], xb[MAXC], dd[MAXC], gg[MAXC];
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const uint32_t c = (uint32_t)(lane + 64 * i);
      if constexpr (XF32) {
        xa[i].i = __builtin_amdgcn_raw_buffer_load_b128(xs, c * 32u, 0, 0);
        xb[i].i = __builtin_amdgcn_raw_buffer_load_b128(xs, c * 32u + 16u, 0, 0);
      } else {
        xa[i].i = __builtin_amdgcn_raw_buffer_load_b128(xs, c * 16u, 0, 0);
      }
      dd[i].i = __builtin_amdgcn_raw_buffer_load_b128(ds, c * 16u, 0, 0);
      gg[i].i = __builtin_amdgcn_raw_buffer_load_b128(gs, c * 16u, 0, 0);
    }
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const bool ok = lane + 64 * i < nchunk;
      f32x8 xv;
      if constexpr (XF32) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          xv[e] = xa[i].f[e];
          xv[4 + e] = xb[i].f[e];
        }
      } else {
        xv = cvt8(xa[i].b);
      }
      const f32x8 dv = cvt8(dd[i].b), gm = cvt8(gg[i].b);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        xh[i][e] = ok ? (xv[e] - mu) * rs : 0.f;        // (past the row x reads 0: x-hat would be -mu * rs)
        g[i][e] = dv[e] * gm[e];
        c1 += g[i][e];
        c2 += g[i][e] * xh[i][e];
      }
    }
    c1 = wave_sum(c1) * inv_n;
    c2 = wave_sum(c2) * inv_n;
    // the residual gradient is read behind the reductions (interleaved reads and writes measured faster on the decoder's shape than
    // one read burst: 19.9 vs 23.0 us at 5120 x 2048) -- all chunks of the row requested together
    const __amdgpu_buffer_rsrc_t rr = make_rsrc(p.dres ? p.dres + xrow * p.ldx : p.dy, p.dres ? rbytes : 0u);
    const __amdgpu_buffer_rsrc_t os = make_rsrc(p.dx + xrow * p.ldx, rbytes);
    const __amdgpu_buffer_rsrc_t qs = make_rsrc(p.dx_drop ? p.dx_drop + xrow * p.ldx : p.dx, p.dx_drop ? rbytes : 0u);
    Ld rv[MAXC];
#pragma unroll
    for (int i = 0; i < MAXC; ++i) rv[i].i = __builtin_amdgcn_raw_buffer_load_b128(rr, (uint32_t)(lane + 64 * i) * 16u, 0, 0);
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int c = lane + 64 * i;
      f32x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = rs * (g[i][e] - c1 - xh[i][e] * c2);
      o += cvt8(rv[i].b);
      Ld ob;
      ob.b = cvt8(o);
      __builtin_amdgcn_raw_buffer_store_b128(ob.i, os, (uint32_t)c * 16u, 0, 0);
      if (p.dx_drop) {
        const f32x8 of = cvt8(ob.b);
        const uint64_t base = p.offset + (uint64_t)r * (uint64_t)p.cols + (uint64_t)(c * 8);
        const f32x8 od = p.drop_thr ? mpv_dropout_vec<f32x8, 8>(of, seed_r, base, p.drop_thr, p.drop_scale) : of * p.drop_scale;
        Ld q;
        q.b = cvt8(od);
        __builtin_amdgcn_raw_buffer_store_b128(q.i, qs, (uint32_t)c * 16u, 0, 0);
      }
    }
  }
}


// waves per SIMD the register allocation is held to: 4 (128 VGPRs) where that fits without scratch; 3 (168) with the next row's
// prefetch (PF, below)
template <int MAXC, bool DPARAM, bool PF = false>
__global__ __launch_bounds__(256, PF ? 3 : MAXC <= 2 ? 4 : MAXC <= 3 ? 2 : 1) void ln_bwd8_kernel(const LnBwdArgs p) {
  const uint64_t seed_r = p.drop_thr ? mpv_resolve_seed(p.seed) : 0;      // (bit 63 set: the seed lives in device memory, mpv_common.h)
  __shared__ float red[DPARAM ? 2 * MAXC * 512 : 1];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // wave-uniform: row pointers and statistics live in SGPRs
  const int nchunk = p.cols >> 3;
  const float inv_n = 1.0f / (float)p.cols;
  f32x8 gacc[DPARAM ? MAXC : 1], bacc[DPARAM ? MAXC : 1];
  if constexpr (DPARAM) {
#pragma unroll
    for (int i = 0; i < MAXC; ++i)
#pragma unroll
      for (int e = 0; e < 8; ++e) gacc[i][e] = bacc[i][e] = 0.f;
  }
  // gamma stays in registers as bf16 for the whole kernel (it was re-read from cache for every row)
  bf16x8 gmb[MAXC];
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int c = lane + 64 * i;
    gmb[i] = c < nchunk ? *(const bf16x8*)(p.gamma + c * 8) : bf16x8{};
  }
  // A row is one round trip to memory: x, dy and the residual gradient are all requested before anything is reduced (the
  // residual gradient used to be read after the two wave reductions -- a second, serial HBM latency per row), and they are
  // held as raw bf16 (12 VGPRs per chunk instead of 16 of fp32 x-hat / g), x-hat and g being recomputed for the output
  // sweep: identical arithmetic, fewer registers, more resident waves.
  if constexpr (PF) {
    // PF (round 4): the NEXT row of the wave is requested before the current one is reduced (two register sets, the loop
    // unrolled by two) -- between a row's stores and the next row's data a wave had nothing in flight.  For the compiler to
    // wait for the CURRENT row only (`s_waitcnt vmcnt(n)` counts the younger requests, and a request inside a branch it cannot
    // count on) every load and store of a row is branch-free: buffer accesses over a descriptor of exactly the row, so chunks
    // past the row read zeros and store nothing, an absent residual gradient is a descriptor of zero bytes, and past the
    // wave's last row the prefetch re-requests the current one (cache hits).  No dropout-masked second output in this form (the
    // launcher keeps such calls on the one-row form: the mask hash does not fit beside two row sets at 3 waves per SIMD).
    union Ld { i32x4 i; bf16x8 b; };
    const uint32_t rbytes = (uint32_t)p.cols * 2u;
    // (the row's statistics are vector loads too -- requested FIRST, so that they are older than the row's data)
    auto load_row = [&](long long r, bf16x8 (&xb)[MAXC], bf16x8 (&db)[MAXC], bf16x8 (&rb)[MAXC], float& mu, float& rs) {
      mu = p.mean[r];
      rs = p.rstd[r];
      const long long xrow = map_row(p.xmap, r);
      const __amdgpu_buffer_rsrc_t xs = make_rsrc(p.x + xrow * p.ldx, rbytes);
      const __amdgpu_buffer_rsrc_t ds = make_rsrc(p.dy + map_row(p.ymap, r) * p.ldy, rbytes);
      const __amdgpu_buffer_rsrc_t rs_ = make_rsrc(p.dres ? p.dres + xrow * p.ldx : p.x, p.dres ? rbytes : 0u);
#pragma unroll
      for (int i = 0; i < MAXC; ++i) {
        const uint32_t off = (uint32_t)(lane + 64 * i) * 16u;
        Ld a, b, c;
        a.i = __builtin_amdgcn_raw_buffer_load_b128(xs, off, 0, 0);
        b.i = __builtin_amdgcn_raw_buffer_load_b128(ds, off, 0, 0);
        c.i = __builtin_amdgcn_raw_buffer_load_b128(rs_, off, 0, 0);
        xb[i] = a.b;
        db[i] = b.b;
        rb[i] = c.b;
      }
    };
    auto do_row = [&](long long r, bf16x8 (&xb)[MAXC], bf16x8 (&db)[MAXC], bf16x8 (&rb)[MAXC], const float mu, const float rs) {
      const long long xrow = map_row(p.xmap, r);
      float c1 = 0.f, c2 = 0.f;
#pragma unroll
      for (int i = 0; i < MAXC; ++i) {
        // (chunks past the row hold x = dy = 0: g = 0 and dy * x-hat = 0 -- they add nothing to the sums although their x-hat = -mu * rs is not 0)
        const f32x8 xv = cvt8(xb[i]), dv = cvt8(db[i]), gm = cvt8(gmb[i]);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float xh = (xv[e] - mu) * rs;
          const float g = dv[e] * gm[e];
          c1 += g;
          c2 += g * xh;
          if constexpr (DPARAM) {
            gacc[i][e] += dv[e] * xh;
            bacc[i][e] += dv[e];
          }
        }
      }
      c1 = wave_sum(c1) * inv_n;
      c2 = wave_sum(c2) * inv_n;
      const __amdgpu_buffer_rsrc_t os = make_rsrc(p.dx + xrow * p.ldx, rbytes);
#pragma unroll
      for (int i = 0; i < MAXC; ++i) {      // opaque: without this the fp32 conversions of the first sweep are kept live across the reductions
        asm volatile("" : "+v"(xb[i]), "+v"(db[i]));
      }
#pragma unroll
      for (int i = 0; i < MAXC; ++i) {
        const int c = lane + 64 * i;
        const f32x8 xv = cvt8(xb[i]), dv = cvt8(db[i]), gm = cvt8(gmb[i]);
        f32x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float xh = (xv[e] - mu) * rs;
          const float g = dv[e] * gm[e];
          o[e] = rs * (g - c1 - xh * c2);
        }
        o += cvt8(rb[i]);                    // zeros without a residual gradient
        Ld ob;
        ob.b = cvt8(o);
        __builtin_amdgcn_raw_buffer_store_b128(ob.i, os, (uint32_t)c * 16u, 0, 0);
      }
    };
    const long long stride = (long long)gridDim.x * 4;
    const long long r0 = (long long)blockIdx.x * 4 + wave;
    const int nrow = r0 < p.rows ? (int)((p.rows - r0 + stride - 1) / stride) : 0;      // rows of this wave (scalar trip count)
    if (nrow > 0) {
      bf16x8 xa[MAXC], da[MAXC], ra[MAXC], xb[MAXC], db[MAXC], rb[MAXC];
      float mua, rsa, mub, rsb;
      load_row(r0, xa, da, ra, mua, rsa);
      for (int k = 0;; k += 2) {
        const long long r = r0 + (long long)k * stride;
        load_row(k + 1 < nrow ? r + stride : r, xb, db, rb, mub, rsb);
        do_row(r, xa, da, ra, mua, rsa);
        if (k + 1 >= nrow) break;
        load_row(k + 2 < nrow ? r + 2 * stride : r + stride, xa, da, ra, mua, rsa);
        do_row(r + stride, xb, db, rb, mub, rsb);
        if (k + 2 >= nrow) break;
      }
    }
  } else
  for (long long r = (long long)blockIdx.x * 4 + wave; r < p.rows; r += (long long)gridDim.x * 4) {
    const long long xrow = map_row(p.xmap, r);
    const bf16* xr = p.x + xrow * p.ldx;
    const bf16* dyr = p.dy + map_row(p.ymap, r) * p.ldy;
    const bf16* drr = p.dres ? p.dres + xrow * p.ldx : nullptr;
    bf16x8 xb[MAXC], db[MAXC], rb[MAXC];
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int c = lane + 64 * i;
      const bool ok = c < nchunk;
      xb[i] = ok ? *(const bf16x8*)(xr + c * 8) : bf16x8{};
      db[i] = ok ? *(const bf16x8*)(dyr + c * 8) : bf16x8{};
      rb[i] = ok && drr ? *(const bf16x8*)(drr + c * 8) : bf16x8{};
    }
    const float mu = p.mean[r], rs = p.rstd[r];
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int c = lane + 64 * i;
      if (c < nchunk) {
        const f32x8 xv = cvt8(xb[i]), dv = cvt8(db[i]), gm = cvt8(gmb[i]);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float xh = (xv[e] - mu) * rs;
          const float g = dv[e] * gm[e];
          c1 += g;
          c2 += g * xh;
          if constexpr (DPARAM) {
            gacc[i][e] += dv[e] * xh;
            bacc[i][e] += dv[e];
          }
        }
      }
    }
    c1 = wave_sum(c1) * inv_n;
    c2 = wave_sum(c2) * inv_n;
    bf16* dxr = p.dx + xrow * p.ldx;
    bf16* ddr = p.dx_drop ? p.dx_drop + xrow * p.ldx : nullptr;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {      // opaque: without this the fp32 conversions of the first sweep are kept live across the reductions
      asm volatile("" : "+v"(xb[i]), "+v"(db[i]));
    }
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int c = lane + 64 * i;
      if (c < nchunk) {
        const f32x8 xv = cvt8(xb[i]), dv = cvt8(db[i]), gm = cvt8(gmb[i]);
        f32x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float xh = (xv[e] - mu) * rs;
          const float g = dv[e] * gm[e];
          o[e] = rs * (g - c1 - xh * c2);
        }
        if (drr) o += cvt8(rb[i]);
        const bf16x8 ob = cvt8(o);
        *(bf16x8*)(dxr + c * 8) = ob;
        if (ddr) {
          const f32x8 of = cvt8(ob);
          const uint64_t base = p.offset + (uint64_t)r * (uint64_t)p.cols + (uint64_t)(c * 8);
          const f32x8 od = p.drop_thr ? mpv_dropout_vec<f32x8, 8>(of, seed_r, base, p.drop_thr, p.drop_scale) : of * p.drop_scale;
          *(bf16x8*)(ddr + c * 8) = cvt8(od);
        }
      }
    }
  }
  if constexpr (DPARAM) {
#pragma unroll 1
    for (int w = 0; w < 4; ++w) {
      if (wave == w) {
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
          const int c = lane + 64 * i;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            if (w == 0) {
              red[c * 8 + e] = gacc[i][e];
              red[MAXC * 512 + c * 8 + e] = bacc[i][e];
            } else {
              red[c * 8 + e] += gacc[i][e];
              red[MAXC * 512 + c * 8 + e] += bacc[i][e];
            }
          }
        }
      }
      __syncthreads();
    }
    float* out = p.part + (long long)blockIdx.x * 2 * p.cols;
    for (int c = threadIdx.x; c < p.cols; c += 256) {
      out[c] = red[c];
      out[p.cols + c] = red[MAXC * 512 + c];
    }
  }
}

// Two-level deterministic reduction of the per-workgroup partials [nblk][2][cols]:
// level 1: grid (cols/64, nblk/64): each workgroup (64 columns x 4 lanes) folds 64 partial rows;
// level 2 (final): grid (cols/64): folds the <= 32 level-1 rows and writes bf16 (optionally accumulating).
__global__ __launch_bounds__(256) void ln_dparam_reduce(const float* __restrict__ part, int nblk, int cols, float* __restrict__ out1,
                                                        bf16* dgamma, bf16* dbeta, int accumulate, int final_level) {
  __shared__ float red[2][4][64];
  const int cl = threadIdx.x & 63, pl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const int i0 = final_level ? 0 : blockIdx.y * 64, i1 = final_level ? nblk : min(nblk, i0 + 64);
  float a = 0.f, b = 0.f;
  if (c < cols)
    for (int i = i0 + pl; i < i1; i += 4) {
      a += part[(long long)i * 2 * cols + c];
      b += part[(long long)i * 2 * cols + cols + c];
    }
  red[0][pl][cl] = a;
  red[1][pl][cl] = b;
  __syncthreads();
  if (pl == 0 && c < cols) {
    a = (red[0][0][cl] + red[0][1][cl]) + (red[0][2][cl] + red[0][3][cl]);
    b = (red[1][0][cl] + red[1][1][cl]) + (red[1][2][cl] + red[1][3][cl]);
    if (!final_level) {
      out1[(long long)blockIdx.y * 2 * cols + c] = a;
      out1[(long long)blockIdx.y * 2 * cols + cols + c] = b;
    } else {
      if (accumulate) {
        a += bf2f(dgamma[c]);
        b += bf2f(dbeta[c]);
      }
      dgamma[c] = f2bf(a);
      dbeta[c] = f2bf(b);
    }
  }
}

// ---- the decoder's residual stream in fp32 -----------------------------------------------------------------------------
// h' = h + a (a = a sublayer's bf16 output: bias and dropout already applied by its GEMM), y = LN(h') in ONE pass: the add that
// the GEMM's residual epilogue used to do in bf16 happens here in fp32, and the sum is kept in fp32 for the next sublayer.
// 48 bf16 roundings of the 24-layer stream were what put the logits 1.2e-2 from the fp32 function (tools/parity_bisect.py:
// every sublayer in bf16 but this stream in fp32 -> 0.87e-2; the reference's own bf16 run: 1.25e-2).
// h_in is fp32, or bf16 for the first LayerNorm behind the embedding (INBF); a == NULL: plain LN of h_in, no h_out.
struct LnStreamArgs {
  const void* h_in;
  const bf16* add;
  float* h_out;
  const bf16 *gamma, *beta;
  bf16* y;
  float *mean, *rstd;
  long long rows;
  int cols;
  long long ldh, lda, ldy;
  float eps;
  RowMap hmap, amap, ymap;
  // dropout of the ADDED tensor (round 4: the bias-dropout of a decoder sublayer moves out of its GEMM's epilogue into the LayerNorm
  // that adds it into the stream -- an HBM-bound kernel with idle VALU slots): element index offset + (row of `add`) * cols + column,
  // the index the GEMM epilogue used, and the dropped value is rounded to bf16 as that epilogue stored it: bit-identical results
  uint32_t drop_thr;
  float drop_scale;
  uint64_t seed, offset;
};
__device__ __forceinline__ bf16x8 ln_stream_drop(const LnStreamArgs& p, bf16x8 av, uint64_t seed_r, long long arow, int col) {
  const f32x8 a = mpv_dropout_vec<f32x8, 8>(cvt8(av), seed_r, p.offset + (uint64_t)arow * (uint64_t)p.cols + (uint64_t)col, p.drop_thr, p.drop_scale);
  return cvt8(a);
}
template <int MAXC, bool INBF>
__global__ __launch_bounds__(256) void ln_stream_fwd_kernel(const LnStreamArgs p) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nchunk = p.cols >> 3;
  const float inv_n = 1.0f / (float)p.cols;
  bf16x8 gmb[MAXC], btb[MAXC];
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int c = lane + 64 * i;
    gmb[i] = c < nchunk ? *(const bf16x8*)(p.gamma + c * 8) : bf16x8{};
    btb[i] = c < nchunk ? *(const bf16x8*)(p.beta + c * 8) : bf16x8{};
  }
  for (long long r = (long long)blockIdx.x * 4 + wave; r < p.rows; r += (long long)gridDim.x * 4) {
    const long long hrow = map_row(p.hmap, r);
    f32x8 v[MAXC];
    bf16x8 av[MAXC];
    const long long arow = p.add ? map_row(p.amap, r) : 0;
    const bf16* ar = p.add ? p.add + arow * p.lda : nullptr;
    const uint64_t seed_r = p.drop_thr ? mpv_resolve_seed(p.seed) : 0;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {      // everything of the row requested before anything is reduced
      const int c = lane + 64 * i;
      const bool ok = c < nchunk;
      if constexpr (INBF) v[i] = ok ? cvt8(*(const bf16x8*)((const bf16*)p.h_in + hrow * p.ldh + c * 8)) : f32x8{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      else v[i] = ok ? *(const f32x8*)((const float*)p.h_in + hrow * p.ldh + c * 8) : f32x8{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      av[i] = ok && ar ? *(const bf16x8*)(ar + c * 8) : bf16x8{};
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      if (ar) {
        if (p.drop_thr && lane + 64 * i < nchunk) av[i] = ln_stream_drop(p, av[i], seed_r, arow, (lane + 64 * i) * 8);
        v[i] += cvt8(av[i]);
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) s += v[i][e];
    }
    if (ar) {
      float* hr = p.h_out + hrow * p.ldh;
#pragma unroll
      for (int i = 0; i < MAXC; ++i)
        if (lane + 64 * i < nchunk) *(f32x8*)(hr + (lane + 64 * i) * 8) = v[i];
    }
    const float mu = wave_sum(s) * inv_n;
    float s2 = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i)
      if (lane + 64 * i < nchunk)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float d = v[i][e] - mu;
          s2 += d * d;
        }
    const float rs = rsqrtf(wave_sum(s2) * inv_n + p.eps);
    bf16* yr = p.y + map_row(p.ymap, r) * p.ldy;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int c = lane + 64 * i;
      if (c < nchunk) {
        const f32x8 g = cvt8(gmb[i]), b = cvt8(btb[i]);
        f32x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (v[i][e] - mu) * rs * g[e] + b[e];
        *(bf16x8*)(yr + c * 8) = cvt8(o);
      }
    }
    if (lane == 0) {
      if (p.mean) p.mean[r] = mu;
      if (p.rstd) p.rstd[r] = rs;
    }
  }
}
// One WORKGROUP per row, one 16-byte chunk (8 columns) per lane: the decoder's streams are few rows (5120 at config B) of many
// columns, and a wave-per-row kernel holds a whole fp32 row in registers (161 VGPRs at 2048 columns: 3 waves per SIMD, two
// rounds of workgroups).  Here a lane holds 8 fp32 values; the two row reductions go through LDS (one float per wave).
// Workgroup barrier that orders LDS traffic only.  __syncthreads() is also a release fence for global memory: with stores in
// flight the compiler puts `s_waitcnt vmcnt(0)` in front of it, i.e. the row reductions of the kernel below waited for the
// write acknowledgements of the fp32 stream (and a finished workgroup kept its slot until its last stores were acknowledged).
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
template <bool INBF>
__global__ __launch_bounds__(1024) void ln_stream_fwd_wg_kernel(const LnStreamArgs p) {
  __shared__ float red[2][16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
  const int nchunk = p.cols >> 3;
  const bool ok = tid < nchunk;
  const float inv_n = 1.0f / (float)p.cols;
  for (long long r = blockIdx.x; r < p.rows; r += gridDim.x) {
    const long long hrow = map_row(p.hmap, r);
    f32x8 v = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    bf16x8 av = {}, gm = {}, bt = {};
    if (ok) {
      // gamma / beta first: behind the (conditional) dropout of `add` the compiler requested them only after the row had arrived --
      // a second, serial round trip per row
      gm = *(const bf16x8*)(p.gamma + tid * 8);
      bt = *(const bf16x8*)(p.beta + tid * 8);
      if constexpr (INBF) v = cvt8(*(const bf16x8*)((const bf16*)p.h_in + hrow * p.ldh + tid * 8));
      else v = *(const f32x8*)((const float*)p.h_in + hrow * p.ldh + tid * 8);
      if (p.add) {
        const long long arow = map_row(p.amap, r);
        av = *(const bf16x8*)(p.add + arow * p.lda + tid * 8);
        if (p.drop_thr) av = ln_stream_drop(p, av, mpv_resolve_seed(p.seed), arow, tid * 8);
      }
    }
    if (p.add) v += cvt8(av);
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) s += v[e];
    s = wave_sum(s);
    if (lane == 0) red[0][wave] = s;
    lds_barrier();
    float tot = 0.f;
    for (int w = 0; w < nwaves; ++w) tot += red[0][w];
    const float mu = tot * inv_n;
    float s2 = 0.f;
    if (ok)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = v[e] - mu;
        s2 += d * d;
      }
    s2 = wave_sum(s2);
    if (lane == 0) red[1][wave] = s2;
    lds_barrier();
    float tot2 = 0.f;
    for (int w = 0; w < nwaves; ++w) tot2 += red[1][w];
    const float rs = rsqrtf(tot2 * inv_n + p.eps);
    if (ok) {
      if (p.add) *(f32x8*)(p.h_out + hrow * p.ldh + tid * 8) = v;      // (behind the reductions: no store is pending at a barrier)
      const f32x8 g = cvt8(gm), b = cvt8(bt);
      f32x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (v[e] - mu) * rs * g[e] + b[e];
      *(bf16x8*)(p.y + map_row(p.ymap, r) * p.ldy + tid * 8) = cvt8(o);
    }
    if (tid == 0) {
      if (p.mean) p.mean[r] = mu;
      if (p.rstd) p.rstd[r] = rs;
    }
    if (r + gridDim.x < p.rows) lds_barrier();      // red[] is reused by the next row of this workgroup
  }
}

template <int MAXC>
void launch_stream_fwd(const LnStreamArgs& a, bool inbf, int grid, hipStream_t s) {
  if (inbf) hipLaunchKernelGGL((ln_stream_fwd_kernel<MAXC, true>), dim3(grid), dim3(256), 0, s, a);
  else hipLaunchKernelGGL((ln_stream_fwd_kernel<MAXC, false>), dim3(grid), dim3(256), 0, s, a);
}

constexpr int LN_BWD_MAX_BLOCKS = 1024;   // measured at 50432 x 768 with dgamma/dbeta: 2048 blocks 82 us, 1024 blocks 73 us, 512 blocks 95 us
constexpr int LN_L1_ROWS = (LN_BWD_MAX_BLOCKS + 63) / 64;
// The parameter-gradient backward of rows <= 1024 columns prefetches each wave's next row (ln_bwd8_kernel<.., PF>): 3 waves per SIMD,
// so launches with parameter gradients are capped at 768 workgroups (3 per CU) -- 50432 x 768: 58.0 -> 54.0 us per launch
// (profiles/r04_c20_ln_bwd_prefetch.md).  MPV_LN_BWD_PF=0 (measurement knob, read once) restores the one-row form on 1024 workgroups.
static bool ln_bwd_pf() {
  static const bool v = [] { const char* e = getenv("MPV_LN_BWD_PF"); return !e || atoi(e) != 0; }();
  return v;
}
static int ln_bwd_blocks(int64_t rows) {
  const int cap = ln_bwd_pf() ? 768 : LN_BWD_MAX_BLOCKS;
  return (int)((rows + 3) / 4 < cap ? (rows + 3) / 4 : cap);
}

template <int MAXC>
void launch_fwd(const LnFwdArgs& a, int grid, hipStream_t s) {
  hipLaunchKernelGGL((ln_fwd_kernel<MAXC>), dim3(grid), dim3(256), 0, s, a);
}
template <int MAXC>
void launch_fwd8(const LnFwdArgs& a, int grid, hipStream_t s) {
  hipLaunchKernelGGL((ln_fwd8_kernel<MAXC>), dim3(grid), dim3(256), 0, s, a);
}
template <int MAXC>
void launch_bwd8(const LnBwdArgs& a, bool dparam, int grid, hipStream_t s) {
  if (dparam && MAXC <= 2 && ln_bwd_pf() && !a.dx_drop)
    hipLaunchKernelGGL((ln_bwd8_kernel<(MAXC <= 2 ? MAXC : 2), true, true>), dim3(grid), dim3(256), 0, s, a);
  else if (dparam)
    hipLaunchKernelGGL((ln_bwd8_kernel<MAXC, true>), dim3(grid), dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL((ln_bwd8_plain_kernel<MAXC>), dim3(grid), dim3(256), 0, s, a);
}
template <int MAXC>
void launch_bwd(const LnBwdArgs& a, bool dparam, int grid, hipStream_t s) {
  if (dparam)
    hipLaunchKernelGGL((ln_bwd_kernel<MAXC, true>), dim3(grid), dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL((ln_bwd_kernel<MAXC, false>), dim3(grid), dim3(256), 0, s, a);
}

}  // namespace

extern "C" int mpv_layernorm_fwd(const void* x, const void* gamma, const void* beta, void* y, float* mean, float* rstd,
                                 int64_t rows, int64_t cols, int64_t ldx, int64_t ldy, float eps, int x_group,
                                 int x_stride, int x_offset, int y_group, int y_stride, int y_offset,
                                 hipStream_t stream) {
  MPV_REQUIRE(x && gamma && beta && y, MPV_E_ARG, "mpv_layernorm_fwd: null pointer");
  MPV_REQUIRE(rows >= 0 && cols > 0 && cols % 4 == 0 && cols <= 4096, MPV_E_SHAPE,
              "mpv_layernorm_fwd: cols=%lld must be a multiple of 4 and <= 4096", (long long)cols);
  MPV_REQUIRE(ldx % 4 == 0 && ldy % 4 == 0, MPV_E_ALIGN, "mpv_layernorm_fwd: leading dims must be multiples of 4");
  if (rows == 0) return MPV_OK;
  LnFwdArgs a = {(const bf16*)x, (const bf16*)gamma, (const bf16*)beta, (bf16*)y, mean, rstd, rows, (int)cols, ldx, ldy, eps,
                 RowMap{x_group, x_stride, x_offset}, RowMap{y_group, y_stride, y_offset}};
  const int grid = (int)((rows + 3) / 4 < 4096 ? (rows + 3) / 4 : 4096);
  const int nc = (int)((cols / 4 + 63) / 64);
  if (cols % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0 && cols <= 4096) {
    const int n8 = (int)((cols / 8 + 63) / 64);
    if (n8 <= 2) launch_fwd8<2>(a, grid, stream);
    else if (n8 <= 3) launch_fwd8<3>(a, grid, stream);
    else if (n8 <= 4) launch_fwd8<4>(a, grid, stream);
    else if (n8 <= 5) launch_fwd8<5>(a, grid, stream);
    else launch_fwd8<8>(a, grid, stream);
    return mpv_check_launch("mpv_layernorm_fwd");
  }
  if (nc <= 3) launch_fwd<3>(a, grid, stream);
  else if (nc <= 6) launch_fwd<6>(a, grid, stream);
  else if (nc <= 8) launch_fwd<8>(a, grid, stream);
  else if (nc <= 10) launch_fwd<10>(a, grid, stream);
  else launch_fwd<16>(a, grid, stream);
  return mpv_check_launch("mpv_layernorm_fwd");
}

extern "C" size_t mpv_layernorm_bwd_workspace_size(int64_t cols) {
  return (size_t)(LN_BWD_MAX_BLOCKS + LN_L1_ROWS) * 2 * (size_t)cols * sizeof(float);
}

// rows of per-workgroup partials [rows][2][cols] a deferred-mode call (accumulate_dparams == MPV_LN_DPARAM_DEFER) leaves at
// the start of its workspace for mpv_layernorm_dparam_finish
extern "C" int mpv_layernorm_bwd_partial_rows(int64_t rows) { return ln_bwd_blocks(rows); }

extern "C" int mpv_layernorm_bwd(const void* dy, const void* x, const void* gamma, const float* mean, const float* rstd,
                                 const void* dres, void* dx, void* dx_drop, float drop_p, uint64_t seed,
                                 uint64_t offset, void* dgamma, void* dbeta, int accumulate_dparams, int64_t rows,
                                 int64_t cols, int64_t ldx, int64_t ldy, int x_group, int x_stride, int x_offset,
                                 int y_group, int y_stride, int y_offset, void* workspace, size_t workspace_bytes,
                                 hipStream_t stream) {
  MPV_REQUIRE(dy && x && gamma && mean && rstd && dx, MPV_E_ARG, "mpv_layernorm_bwd: null pointer");
  MPV_REQUIRE(rows >= 0 && cols > 0 && cols % 4 == 0 && cols <= 4096, MPV_E_SHAPE,
              "mpv_layernorm_bwd: cols=%lld must be a multiple of 4 and <= 4096", (long long)cols);
  MPV_REQUIRE((dgamma == nullptr) == (dbeta == nullptr), MPV_E_ARG, "mpv_layernorm_bwd: dgamma/dbeta must come together");
  MPV_REQUIRE(drop_p >= 0.f && drop_p < 1.f, MPV_E_ARG, "mpv_layernorm_bwd: bad dropout_p");
  if (rows == 0) return MPV_OK;
  const bool dparam = dgamma != nullptr;
  // the 768-workgroup cap belongs to the prefetching parameter-gradient kernel (and to the partial rows mpv_layernorm_bwd_partial_rows
  // promises a deferred finish); a launch without parameter gradients keeps the 1024 that was measured best for it
#ifdef MPV_AB_LN_CAP768      // (same-box A/B build: round 4's grid for every launch)
  int grid = ln_bwd_blocks(rows);
#else
  int grid = dparam ? ln_bwd_blocks(rows) : (int)((rows + 3) / 4 < LN_BWD_MAX_BLOCKS ? (rows + 3) / 4 : LN_BWD_MAX_BLOCKS);
#endif
  if (dparam)
    MPV_REQUIRE(workspace && workspace_bytes >= (size_t)(grid + LN_L1_ROWS) * 2 * cols * sizeof(float), MPV_E_ARG,
                "mpv_layernorm_bwd: workspace too small (need %zu bytes)", (size_t)(grid + LN_L1_ROWS) * 2 * cols * sizeof(float));
  LnBwdArgs a = {};
  a.dy = (const bf16*)dy;
  a.x = (const bf16*)x;
  a.gamma = (const bf16*)gamma;
  a.mean = mean;
  a.rstd = rstd;
  a.dres = (const bf16*)dres;
  a.dx = (bf16*)dx;
  a.dx_drop = (bf16*)dx_drop;
  a.drop_thr = drop_p > 0.f ? mpv_drop_threshold(drop_p) : 0;
  a.drop_scale = 1.0f / (1.0f - drop_p);
  a.seed = seed;
  a.offset = offset;
  a.part = dparam ? (float*)workspace : nullptr;
  a.rows = rows;
  a.cols = (int)cols;
  a.ldx = ldx;
  a.ldy = ldy;
  a.xmap = RowMap{x_group, x_stride, x_offset};
  a.ymap = RowMap{y_group, y_stride, y_offset};
  const int nc = (int)((cols / 4 + 63) / 64);
  if (cols % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0) {
    const int n8 = (int)((cols / 8 + 63) / 64);
    if (n8 <= 2) launch_bwd8<2>(a, dparam, grid, stream);
    else if (n8 <= 3) launch_bwd8<3>(a, dparam, grid, stream);
    else if (n8 <= 4) launch_bwd8<4>(a, dparam, grid, stream);
    else if (n8 <= 5) launch_bwd8<5>(a, dparam, grid, stream);
    else launch_bwd8<8>(a, dparam, grid, stream);
  } else if (nc <= 3) launch_bwd<3>(a, dparam, grid, stream);
  else if (nc <= 6) launch_bwd<6>(a, dparam, grid, stream);
  else if (nc <= 8) launch_bwd<8>(a, dparam, grid, stream);
  else if (nc <= 10) launch_bwd<10>(a, dparam, grid, stream);
  else launch_bwd<16>(a, dparam, grid, stream);
  if (dparam && accumulate_dparams != MPV_LN_DPARAM_DEFER) {
    float* l0 = (float*)workspace;
    float* l1 = l0 + (size_t)grid * 2 * cols;
    const unsigned gx = (unsigned)((cols + 63) / 64);
    const int n1 = (grid + 63) / 64;
    if (grid > 64) {
      hipLaunchKernelGGL(ln_dparam_reduce, dim3(gx, n1), dim3(256), 0, stream, (const float*)l0, grid, (int)cols, l1, (bf16*)nullptr,
                         (bf16*)nullptr, 0, 0);
      hipLaunchKernelGGL(ln_dparam_reduce, dim3(gx), dim3(256), 0, stream, (const float*)l1, n1, (int)cols, (float*)nullptr,
                         (bf16*)dgamma, (bf16*)dbeta, accumulate_dparams, 1);
    } else {
      hipLaunchKernelGGL(ln_dparam_reduce, dim3(gx), dim3(256), 0, stream, (const float*)l0, grid, (int)cols, (float*)nullptr,
                         (bf16*)dgamma, (bf16*)dbeta, accumulate_dparams, 1);
    }
  }
  return mpv_check_launch("mpv_layernorm_bwd");
}

extern "C" int mpv_ln_stream_fwd_drop(const void* h_in, int h_in_bf16, const void* add, float* h_out, const void* gamma, const void* beta, void* y,
                                      float* mean, float* rstd, int64_t rows, int64_t cols, int64_t ldh, int64_t lda, int64_t ldy, float eps,
                                      int h_group, int h_stride, int h_offset, int a_group, int a_stride, int a_offset, int y_group, int y_stride,
                                      int y_offset, float add_dropout_p, uint64_t seed, uint64_t offset, hipStream_t stream) {
  MPV_REQUIRE(h_in && gamma && beta && y, MPV_E_ARG, "mpv_ln_stream_fwd_drop: null pointer");
  MPV_REQUIRE((add == nullptr) == (h_out == nullptr), MPV_E_ARG, "mpv_ln_stream_fwd_drop: add and h_out come together");
  MPV_REQUIRE(rows >= 0 && cols > 0 && cols % 8 == 0 && cols <= 4096 && ldh % 8 == 0 && lda % 8 == 0 && ldy % 8 == 0, MPV_E_SHAPE,
              "mpv_ln_stream_fwd_drop: cols=%lld and the leading dims must be multiples of 8, cols <= 4096", (long long)cols);
  MPV_REQUIRE((((uintptr_t)h_in | (uintptr_t)h_out) & 31) == 0 || h_in_bf16, MPV_E_ALIGN, "mpv_ln_stream_fwd_drop: the fp32 stream must be 32-byte aligned");
  MPV_REQUIRE(add_dropout_p >= 0.f && add_dropout_p < 1.f && (add_dropout_p == 0.f || add), MPV_E_ARG, "mpv_ln_stream_fwd_drop: bad add_dropout_p");
  if (rows == 0) return MPV_OK;
  LnStreamArgs a = {h_in, (const bf16*)add, h_out, (const bf16*)gamma, (const bf16*)beta, (bf16*)y, mean, rstd, rows, (int)cols, ldh, lda, ldy, eps,
                    RowMap{h_group, h_stride, h_offset}, RowMap{a_group, a_stride, a_offset}, RowMap{y_group, y_stride, y_offset},
                    add_dropout_p > 0.f ? mpv_drop_threshold(add_dropout_p) : 0u, 1.0f / (1.0f - add_dropout_p), seed, offset};
  if (cols >= 1024) {      // wide rows: a workgroup per row
    const int threads = (int)((cols / 8 + 63) / 64 * 64);
    const int g = (int)(rows < 65536 ? rows : 65536);
    if (h_in_bf16) hipLaunchKernelGGL((ln_stream_fwd_wg_kernel<true>), dim3(g), dim3(threads), 0, stream, a);
    else hipLaunchKernelGGL((ln_stream_fwd_wg_kernel<false>), dim3(g), dim3(threads), 0, stream, a);
    return mpv_check_launch("mpv_ln_stream_fwd_drop");
  }
  const int grid = (int)((rows + 3) / 4 < 4096 ? (rows + 3) / 4 : 4096);
  const int n8 = (int)((cols / 8 + 63) / 64);
  if (n8 <= 2) launch_stream_fwd<2>(a, h_in_bf16 != 0, grid, stream);
  else if (n8 <= 4) launch_stream_fwd<4>(a, h_in_bf16 != 0, grid, stream);
  else if (n8 <= 5) launch_stream_fwd<5>(a, h_in_bf16 != 0, grid, stream);
  else launch_stream_fwd<8>(a, h_in_bf16 != 0, grid, stream);
  return mpv_check_launch("mpv_ln_stream_fwd_drop");
}

extern "C" int mpv_ln_stream_fwd(const void* h_in, int h_in_bf16, const void* add, float* h_out, const void* gamma, const void* beta, void* y,
                                 float* mean, float* rstd, int64_t rows, int64_t cols, int64_t ldh, int64_t lda, int64_t ldy, float eps,
                                 int h_group, int h_stride, int h_offset, int a_group, int a_stride, int a_offset, int y_group, int y_stride,
                                 int y_offset, hipStream_t stream) {
  return mpv_ln_stream_fwd_drop(h_in, h_in_bf16, add, h_out, gamma, beta, y, mean, rstd, rows, cols, ldh, lda, ldy, eps, h_group, h_stride, h_offset,
                                a_group, a_stride, a_offset, y_group, y_stride, y_offset, 0.f, 0, 0, stream);
}

extern "C" int mpv_ln_stream_bwd(const void* dy, const float* x, const void* gamma, const float* mean, const float* rstd, const void* dres,
                                 void* dx, void* dx_drop, float drop_p, uint64_t seed, uint64_t offset, int64_t rows, int64_t cols,
                                 int64_t ldx, int64_t ldy, int x_group, int x_stride, int x_offset, int y_group, int y_stride, int y_offset,
                                 hipStream_t stream) {
  MPV_REQUIRE(dy && x && gamma && mean && rstd && dx, MPV_E_ARG, "mpv_ln_stream_bwd: null pointer");
  MPV_REQUIRE(rows >= 0 && cols > 0 && cols % 8 == 0 && cols <= 4096 && ldx % 8 == 0 && ldy % 8 == 0, MPV_E_SHAPE,
              "mpv_ln_stream_bwd: cols=%lld and the leading dims must be multiples of 8, cols <= 4096", (long long)cols);
  MPV_REQUIRE(drop_p >= 0.f && drop_p < 1.f, MPV_E_ARG, "mpv_ln_stream_bwd: bad dropout_p");
  if (rows == 0) return MPV_OK;
  LnBwdArgs a = {};
  a.dy = (const bf16*)dy;
  a.x = (const bf16*)x;      // read as fp32 by the XF32 instances
  a.gamma = (const bf16*)gamma;
  a.mean = mean;
  a.rstd = rstd;
  a.dres = (const bf16*)dres;
  a.dx = (bf16*)dx;
  a.dx_drop = (bf16*)dx_drop;
  a.drop_thr = drop_p > 0.f ? mpv_drop_threshold(drop_p) : 0;
  a.drop_scale = 1.0f / (1.0f - drop_p);
  a.seed = seed;
  a.offset = offset;
  a.rows = rows;
  a.cols = (int)cols;
  a.ldx = ldx;
  a.ldy = ldy;
  a.xmap = RowMap{x_group, x_stride, x_offset};
  a.ymap = RowMap{y_group, y_stride, y_offset};
  const int grid = (int)((rows + 3) / 4 < LN_BWD_MAX_BLOCKS ? (rows + 3) / 4 : LN_BWD_MAX_BLOCKS);
  const int n8 = (int)((cols / 8 + 63) / 64);
  if (n8 <= 2) hipLaunchKernelGGL((ln_bwd8_plain_kernel<2, true>), dim3(grid), dim3(256), 0, stream, a);
  else if (n8 <= 4) hipLaunchKernelGGL((ln_bwd8_plain_kernel<4, true>), dim3(grid), dim3(256), 0, stream, a);
  else if (n8 <= 5) hipLaunchKernelGGL((ln_bwd8_plain_kernel<5, true>), dim3(grid), dim3(256), 0, stream, a);
  else hipLaunchKernelGGL((ln_bwd8_plain_kernel<8, true>), dim3(grid), dim3(256), 0, stream, a);
  return mpv_check_launch("mpv_ln_stream_bwd");
}
