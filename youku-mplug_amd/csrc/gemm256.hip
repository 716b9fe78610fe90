// gemm256.hip -- 256x256x64 eight-phase bf16 MFMA GEMM for gfx950 (CDNA4), hand-written.
//
//   C[M,N] = epilogue( sum_k Aop(m,k) * Bop(n,k) )        (same operand conventions as gemm.hip)
//
// Replaces the same reference call sites as gemm.hip (F.linear / mpu.Column/RowParallelLinear and their autograd
// backward: models/vision_transformer.py:104,108,175,205,250; models/modeling_distributed_gpt3.py:562,573,843,852,1348).
//
// Structure (one 512-thread workgroup per CU, 8 waves = 2 (M) x 4 (N), each wave a 128x64 output = 32 accumulator
// tiles of v_mfma_f32_16x16x32_bf16):
//
//  * LDS holds a ring of 8 "units" of 16 KiB = two K-tiles (BK = 64) of 4 units each.  A unit is the part of a K-tile
//    that one phase of the schedule reads:   U0 = A rows {0..63, 128..191}   (first 64-row half of both wave rows)
//                                             U1 = B rows {64w..64w+31}       (first 32-column half of every wave column)
//                                             U2 = B rows {64w+32..64w+63}    U3 = A rows {64..127, 192..255}
//    so that a unit is dead as soon as its phase has read it and can be refilled two phases later.
//  * Every unit is filled by LDS-DMA (buffer_load_dwordx4 ... lds, 2 per thread per unit).  The DMA writes LDS
//    lane-linearly, so the bank-conflict-avoiding XOR swizzles are applied to the per-lane GLOBAL source address and
//    again on the LDS read.  k-contiguous operands: [128 rows][64 k] image, 16-byte chunk index ^ ((row >> 1) & 7),
//    read with ds_read_b128.  Reduction-slow operands (dgrad / wgrad): [64 k][128 cols] image, 32-byte unit index
//    ^ f8(k), read with ds_read_b64_tr_b16 (the transposing LDS read) -- no transposed copies of anything.
//  * The K loop runs 4 phases per K-tile.  A phase is {ds_reads of this phase's unit(s); issue the DMA of the unit
//    six phases ahead; s_waitcnt vmcnt(6); s_barrier; 16 MFMAs (one 64x32 quadrant x K=64); s_barrier}.  The two wave
//    rows run the same code offset by one barrier, so on every SIMD one wave is in its MFMA segment while the other
//    one reads LDS and issues DMA.  vmcnt is never drained inside the loop: three units (6 DMA instructions per lane)
//    stay in flight across the barriers; a unit is read two phases after the wait that retires it.  The prologue issues
//    six units and releases phase 0 as soon as the first two are in (round 4: -0.16 ms per step, profiles/r04_c18_*).
//  * MFMA operands are swapped (D = Bfrag x Afrag) so a lane owns 4 consecutive output columns of one row; the
//    epilogue rounds (acc * alpha + bias) to bf16 in registers, stages the whole 256x256 tile in LDS and finishes it
//    row-contiguously with 16-byte loads/stores (activation (+pre-activation copy), activation-backward multiply,
//    dropout, residual, accumulate, C row map).  fp32 outputs (split-K partials) are staged in four 64-row passes.
//  * wgrad: the fused bias gradient (column sums of the A operand) is one extra MFMA per A fragment against a
//    constant ones operand, spread over the four wave columns.
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "mpv_common.h"
#include "../../include/mpv.h"
#include "gemm_args.h"

namespace {

constexpr int TM = 256, TN = 256, TK = 64;
constexpr int UNIT = 16384;                    // bytes per ring unit
constexpr int CPITCH = 264;                    // bf16 staging pitch (elements): 528-byte rows keep 16-byte alignment
constexpr int FPITCH = 260;                    // fp32 staging pitch (floats) of a 64-row pass
constexpr int STAGE_BYTES = TM * CPITCH * 2;   // 135168 >= 8 * UNIT: ring, then the staged output tile
constexpr int BIAS_LDS = STAGE_BYTES;          // behind it: the tile's 256 bias values (one 1 KiB DMA instruction: 512 bytes + zero fill)
constexpr int SMEM_BYTES = STAGE_BYTES + 1024;

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
template <int V>
using IC = std::integral_constant<int, V>;

#define SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
// 16-byte output stores of the finish pass.  NON-TEMPORAL (round 5): an output tile is written once and not read again by this launch;
// as plain stores its 128 KiB per tile pushed the A / B panels the resident workgroups share out of the XCD's 4 MiB L2 (same-box
// alternation of whole libraries: GEMM family 60.5 -> 58.8 ms per step, step 76.0 -> 75.1 ms, identical results;
// profiles/r05_c14_gemm_store_flavours_ab.log).  -DMPV_AB_PLAIN_C / _2ND / _EXT, -DMPV_AB_NT_F32, -DMPV_AB_NT_DMA_A / _B: A/B builds (profiles/r05_c15_*, r05_c16_*).
#ifdef MPV_AB_PLAIN_C
#define MPV_ST_C(ptr, val) (*(bf16x8*)(ptr) = (val))
#else
#define MPV_ST_C(ptr, val) __builtin_nontemporal_store((val), (bf16x8*)(ptr))
#endif
#ifdef MPV_AB_PLAIN_2ND
#define MPV_ST_2ND(ptr, val) (*(bf16x8*)(ptr) = (val))
#else
#define MPV_ST_2ND(ptr, val) __builtin_nontemporal_store((val), (bf16x8*)(ptr))
#endif

__device__ __forceinline__ bf16x8 lds_read_tr8(const char* p) {
  // two transposing reads: k rows +0..3 and +4..7 (1 KiB apart in the [k][256 B] image)
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 1024));
  union {
    struct { s16x4 a, b; } s;
    bf16x8 v;
  } u;
  u.s.a = lo;
  u.s.b = hi;
  return u.v;
}


enum EpKind { EP_PLAIN, EP_ERF_PRE, EP_TANH_PRE, EP_BWD_ERF, EP_BWD_TANH, EP_RES, EP_DROP_RES, EP_DROP, EP_GENERIC, EP_MUL };

template <int ACT>
__device__ __forceinline__ f32x8 apply_act(f32x8 v) {
  if constexpr (ACT == MPV_ACT_GELU_ERF || ACT == MPV_ACT_GELU_TANH) {
#pragma unroll
    for (int e = 0; e < 8; e += 2) {       // two lanes of the register pair per instruction (v_pk_fma_f32)
      const f32x2 r = mpv_gelu_t<ACT == MPV_ACT_GELU_ERF ? 1 : 2>(f32x2{v[e], v[e + 1]});
      v[e] = r[0];
      v[e + 1] = r[1];
    }
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
  }
  return v;
}
// act'(z) itself (the forward launch that parks it for the dgrad: mpv.h preact_deriv)
template <int ACT>
__device__ __forceinline__ f32x8 act_deriv(f32x8 z) {
  f32x8 d;
#pragma unroll
  for (int e = 0; e < 8; e += 2) {
    const f32x2 r = mpv_gelu_grad_mul_t<ACT == MPV_ACT_GELU_ERF ? 1 : 2>(f32x2{1.0f, 1.0f}, f32x2{z[e], z[e + 1]});
    d[e] = r[0];
    d[e + 1] = r[1];
  }
  return d;
}
template <int ACT>
__device__ __forceinline__ f32x8 apply_act_grad(f32x8 v, f32x8 z) {
  if constexpr (ACT == MPV_ACT_GELU_ERF || ACT == MPV_ACT_GELU_TANH) {
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
      const f32x2 r = mpv_gelu_grad_mul_t<ACT == MPV_ACT_GELU_ERF ? 1 : 2>(f32x2{v[e], v[e + 1]}, f32x2{z[e], z[e + 1]});
      v[e] = r[0];
      v[e + 1] = r[1];
    }
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = z[e] > 0.f ? v[e] : 0.f;
  }
  return v;
}

// second half of the epilogue: the staged bf16 tile (acc * alpha + bias, pitch CPITCH) -> global, 8 columns per thread.
// Everything the finish reads from global memory (GELU' pre-activation, residual) is fetched for all 16 chunks of the
// thread right after the main loop, before the accumulators are staged: interleaved with the stores each load sat behind `s_waitcnt vmcnt(0)` (stores
// count in vmcnt, and the compiler cannot prove C does not alias z / the residual), 16 serial round trips = 14 us per
// tile at K = 768; up front they cost one latency, hidden behind the staging pass and its barrier.
// NIT = 16-byte chunks per thread = tile rows / 16 (16 for the 256-row tile, 12 / 10 for the 192 / 160-row variants)
template <int KIND, int NIT>
struct EpExt {
  static constexpr bool EXT = KIND == EP_BWD_ERF || KIND == EP_BWD_TANH || KIND == EP_RES || KIND == EP_DROP_RES || KIND == EP_MUL;
  bf16x8 v[EXT ? NIT : 1];
};
template <int KIND, int NIT>
__device__ __forceinline__ void prefetch_rows(const GemmArgs& p, EpExt<KIND, NIT>& x, int tid, int m0, int n0) {
  if constexpr (EpExt<KIND, NIT>::EXT) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int c = tid + 512 * it;
      const int m = m0 + (c >> 5), n = n0 + (c & 31) * 8;
      x.v[it] = bf16x8{};
      if (m < p.M && n < p.N) {
#ifndef MPV_AB_PLAIN_EXT      // read once, by this thread only: non-temporal (step 74.3 -> 74.0 ms on top of the non-temporal stores)
        if constexpr (KIND == EP_BWD_ERF || KIND == EP_BWD_TANH || KIND == EP_MUL) x.v[it] = __builtin_nontemporal_load((const bf16x8*)(p.actz + (long long)m * p.ldz + n));
        else x.v[it] = __builtin_nontemporal_load((const bf16x8*)(p.residual + map_row(p.cmap, m) * p.ldr + n));
#else
        if constexpr (KIND == EP_BWD_ERF || KIND == EP_BWD_TANH || KIND == EP_MUL) x.v[it] = *(const bf16x8*)(p.actz + (long long)m * p.ldz + n);
        else x.v[it] = *(const bf16x8*)(p.residual + map_row(p.cmap, m) * p.ldr + n);
#endif
      }
    }
  }
}
template <int KIND, int NIT>
__device__ __forceinline__ void finish_rows(const GemmArgs& p, const EpExt<KIND, NIT>& x, const bf16* cb, int tid, int m0, int n0) {
  constexpr bool EXT = EpExt<KIND, NIT>::EXT;
  constexpr int JW = NIT % 4 == 0 ? 4 : 2, HN = NIT / JW;      // rolled outer loop x unrolled inner loop (code size)
  static_assert(HN >= 3 && HN <= 5, "nested selects below");
  const bf16x8* ext = x.v;
  const uint64_t seed_r = p.drop_thr ? mpv_resolve_seed(p.seed) : 0;      // (bit 63 set: the seed lives in device memory, mpv_common.h)
#pragma unroll 1
  for (int h = 0; h < HN; ++h)
#pragma unroll
  for (int j = 0; j < JW; ++j) {
    const int c = tid + 512 * (h * JW + j);
    bf16x8 ex = bf16x8{};
    if constexpr (EXT) {     // nested register selects keep the loop rolled (written as one nested ?: on purpose: a chain of
                             // separate selects gets folded back into an indexed load from a scratch copy of the array)
      if constexpr (HN == 3) ex = h == 0 ? ext[j] : h == 1 ? ext[JW + j] : ext[2 * JW + j];
      else if constexpr (HN == 4) ex = h == 0 ? ext[j] : h == 1 ? ext[JW + j] : h == 2 ? ext[2 * JW + j] : ext[3 * JW + j];
      else ex = h == 0 ? ext[j] : h == 1 ? ext[JW + j] : h == 2 ? ext[2 * JW + j] : h == 3 ? ext[3 * JW + j] : ext[4 * JW + j];
    }
    const int row = c >> 5, col = (c & 31) * 8;
    const int m = m0 + row, n = n0 + col;
    if (m < p.M && n < p.N) {
      const bf16x8 zb = *(const bf16x8*)(cb + row * CPITCH + col);
      const long long crow = map_row(p.cmap, m);
      bf16* cp = (bf16*)p.C + crow * p.ldc + n;
      if constexpr (KIND == EP_PLAIN) {
        if (p.keep_c) *(bf16x8*)cp = zb;      // (mpv.h keep_output: a small tile a latency-bound consumer reads next stays cacheable)
        else MPV_ST_C(cp, zb);
      } else if constexpr (KIND == EP_ERF_PRE || KIND == EP_TANH_PRE) {
        constexpr int A = KIND == EP_ERF_PRE ? MPV_ACT_GELU_ERF : MPV_ACT_GELU_TANH;
        // preact_deriv: the second output is act'(z) for the dgrad to multiply by (one wave-uniform branch per 8-element chunk);
        // this epilogue is bound by its two stores, the extra polynomial rides in its idle VALU slots
        if (p.preact_deriv) {
          if constexpr (A == MPV_ACT_GELU_TANH) {      // both from one polynomial (mpv_gelu_tanh_both_t)
            const f32x8 zf = cvt8(zb);
            f32x8 gv, dv;
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
              f32x2 g2, d2;
              mpv_gelu_tanh_both_t(f32x2{zf[e], zf[e + 1]}, g2, d2);
              gv[e] = g2[0]; gv[e + 1] = g2[1];
              dv[e] = d2[0]; dv[e + 1] = d2[1];
            }
            MPV_ST_2ND((p.preact + crow * p.ldc + n), cvt8(dv));
            MPV_ST_C(cp, cvt8(gv));
          } else {
            MPV_ST_2ND((p.preact + crow * p.ldc + n), cvt8(act_deriv<A>(cvt8(zb))));
            MPV_ST_C(cp, cvt8(apply_act<A>(cvt8(zb))));
          }
        } else {
          MPV_ST_2ND((p.preact + crow * p.ldc + n), zb);
          MPV_ST_C(cp, cvt8(apply_act<A>(cvt8(zb))));
        }
      } else if constexpr (KIND == EP_BWD_ERF || KIND == EP_BWD_TANH) {
        MPV_ST_C(cp, cvt8(apply_act_grad<KIND == EP_BWD_ERF ? MPV_ACT_GELU_ERF : MPV_ACT_GELU_TANH>(cvt8(zb), cvt8(ex))));
      } else if constexpr (KIND == EP_MUL) {          // act_bwd == MPV_ACT_DERIV: ex holds act'(z)
        MPV_ST_C(cp, cvt8(cvt8(zb) * cvt8(ex)));
      } else if constexpr (KIND == EP_RES) {
        if (p.tap_out && m % p.tap_group == 0) MPV_ST_2ND((p.tap_out + (long long)(m / p.tap_group) * p.N + n), zb);
        MPV_ST_C(cp, cvt8(cvt8(zb) + cvt8(ex)));
      } else if constexpr (KIND == EP_DROP_RES) {
        const uint64_t base = p.drop_offset + (uint64_t)m * (uint64_t)p.N + (uint64_t)n;
        const f32x8 v = mpv_dropout_vec<f32x8, 8>(cvt8(zb), seed_r, base, p.drop_thr, p.drop_scale);
        MPV_ST_C(cp, cvt8(v + cvt8(ex)));
      } else if constexpr (KIND == EP_DROP) {       // dropout(acc + bias): the decoder's sublayer outputs (the residual add is the next LayerNorm's, in fp32)
        const uint64_t base = p.drop_offset + (uint64_t)m * (uint64_t)p.N + (uint64_t)n;
        MPV_ST_C(cp, cvt8(mpv_dropout_vec<f32x8, 8>(cvt8(zb), seed_r, base, p.drop_thr, p.drop_scale)));
      } else {
        f32x8 v = cvt8(zb);
        if (p.tap_out && m % p.tap_group == 0) MPV_ST_2ND((p.tap_out + (long long)(m / p.tap_group) * p.N + n), zb);
        if (p.act) {
          if (p.preact) {
            if (p.preact_deriv) MPV_ST_2ND((p.preact + crow * p.ldc + n), cvt8(p.act == MPV_ACT_GELU_ERF ? act_deriv<MPV_ACT_GELU_ERF>(v) : act_deriv<MPV_ACT_GELU_TANH>(v)));
            else MPV_ST_2ND((p.preact + crow * p.ldc + n), zb);
          }
          v = p.act == MPV_ACT_GELU_ERF ? apply_act<MPV_ACT_GELU_ERF>(v) : p.act == MPV_ACT_GELU_TANH ? apply_act<MPV_ACT_GELU_TANH>(v) : apply_act<MPV_ACT_RELU>(v);
        }
        if (p.act_bwd) {
          const f32x8 z = cvt8(*(const bf16x8*)(p.actz + (long long)m * p.ldz + n));
          v = p.act_bwd == MPV_ACT_GELU_ERF ? apply_act_grad<MPV_ACT_GELU_ERF>(v, z)
              : p.act_bwd == MPV_ACT_GELU_TANH ? apply_act_grad<MPV_ACT_GELU_TANH>(v, z)
              : p.act_bwd == MPV_ACT_DERIV ? v * z : apply_act_grad<MPV_ACT_RELU>(v, z);
        }
        if (p.drop_thr) {
          const uint64_t base = p.drop_offset + (uint64_t)m * (uint64_t)p.N + (uint64_t)n;
          v = mpv_dropout_vec<f32x8, 8>(v, seed_r, base, p.drop_thr, p.drop_scale);
        }
        if (p.residual) v += cvt8(*(const bf16x8*)(p.residual + crow * p.ldr + n));
        if (p.accumulate) v += cvt8(*(const bf16x8*)cp);
        MPV_ST_C(cp, cvt8(v));
      }
    }
  }
}

// LDS-DMA of 16 bytes per lane (1 KiB per wave) to LDS byte address `lds_addr` (wave-uniform) + 16 * lane, from
// rsrc base + soff + voff.  Inline asm on purpose: hipcc orders every LDS read it cannot disambiguate behind a
// pending buffer_load..lds it knows about with s_waitcnt vmcnt(0) (seen in the .s for ds_read_b64_tr_b16), which would
// drain the ring every phase; the waits for these DMAs are the counted s_waitcnt vmcnt(N) of the schedule below.
__device__ __forceinline__ void dma16(i32x4 rsrc, uint32_t lds_addr, uint32_t voff, uint32_t soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :
               : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff)
               : "memory");
}
__device__ __forceinline__ void dma16_nt(i32x4 rsrc, uint32_t lds_addr, uint32_t voff, uint32_t soff) {      // (A/B builds: -DMPV_AB_NT_DMA_A / _B)
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen nt lds"
               :
               : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff)
               : "memory");
}
__device__ __forceinline__ i32x4 raw_rsrc(const void* ptr, uint32_t bytes) {
  const uint64_t a = (uint64_t)ptr;
  return i32x4{(int)(uint32_t)a, (int)((uint32_t)(a >> 32) & 0xffffu), (int)bytes, 0x00020000};
}

// MB1 = 16-row blocks in the SECOND 64-row half of a wave row (4: the 256-row tile).  MB1 = 2 / 1 give 192 / 160-row tiles
// on the same ring, DMA schedule and barriers: a wave row then owns 64 + 16*MB1 tile rows, the unused rows of the second
// A unit are out-of-range DMA lanes (zero fill, no traffic) and their LDS reads / MFMAs are not issued.  For problems
// whose 256-row tiling leaves CUs idle (M = 5120, N = 2048: 160 tiles on 256 CUs; 160-row tiles: exactly 256).
template <bool TA, bool TB, bool KMAP, int MB1 = 4>
__global__ __launch_bounds__(512, 2) void gemm256_kernel(const GemmArgs p) {
  static_assert(MB1 == 4 || !TA, "short tiles exist for a k-contiguous A operand (forward / dgrad forms) only");
  constexpr int WROWS = 64 + 16 * MB1;          // tile rows of one wave row
  constexpr int TME = 2 * WROWS;                // tile rows
  constexpr int NIT = TME / 16;
  __shared__ __attribute__((aligned(1024))) char smem[SMEM_BYTES];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int l15 = lane & 15, lg = lane >> 4;

  // ---- workgroup -> (split, tile): XCD-aware bijective remap (workgroup b runs on XCD b % 8; each XCD walks a contiguous
  // range), then GM m-tiles per n-tile inside the range so the 32 workgroups resident on an XCD share panels in its L2.
  const int nwg = p.nwg * p.splits;
  const int bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
  const int lin = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int split = lin / p.nwg;
  const int pid = lin - split * p.nwg;
  const int GM = p.gm;
  const int gsz = GM * p.tiles_n;
  const int grp = pid / gsz, rem = pid - grp * gsz;
  const int gm = min(GM, p.tiles_m - grp * GM);
  const int tile_n = rem / gm, tile_m = grp * GM + (rem - tile_n * gm);
  const int m0 = p.m_base + tile_m * TME, n0 = tile_n * TN;

  const int kbeg = split * p.k_per_split;
  const int kend = min(p.K, kbeg + p.k_per_split);
  // whole K-tiles for the k-contiguous forms (launcher); the weight-gradient form (both operands reduction-slow: a K row is a memory row) may
  // end in a partial tile -- its rows past K lie behind the operands' buffer descriptors and read as zeros (round 6: the abstractor's K / V
  // weight gradient reduces over B * (1 + T * N) = 50208 rows, not a multiple of 64, and used to fall back to the 128 x 128 kernel)
  const int nk = (kend - kbeg + TK - 1) / TK;

  const i32x4 ra = raw_rsrc(p.A, p.a_bytes);
  const i32x4 rb = raw_rsrc(p.B, p.b_bytes);
  const uint32_t smem_base = (uint32_t)(uintptr_t)(lds_void*)smem;

  // ---- per-lane DMA source offsets (bytes) of the 2 instructions a thread issues per unit
  constexpr uint32_t OOB = 0x80000000u;
  uint32_t vo[4][2];
  constexpr bool kmapped = KMAP;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    if constexpr (!TA) {   // [128 rows][64 k] image: instruction (wave, j) covers unit rows 16*wave + 8*j + lane/8
      const int ur = wave * 16 + j * 8 + (lane >> 3);
      const int kc = (lane & 7) ^ (j * 4 + (lane >> 4));          // = chunk slot ^ ((ur >> 1) & 7)
      const int mA = m0 + (ur >> 6) * WROWS + (ur & 63);
      vo[0][j] = mA < p.M ? (uint32_t)((map_row(p.amap, mA) * p.lda + kc * 8) * 2) : OOB;
      vo[3][j] = (ur & 63) < 16 * MB1 && mA + 64 < p.M ? (uint32_t)((map_row(p.amap, mA + 64) * p.lda + kc * 8) * 2) : OOB;
    } else {               // [64 k][128 cols] image: instruction (wave, j) covers k rows 8*wave + 4*j + lane/16
      const int f8 = (lane >> 4) | ((wave & 1) << 2);
      const int c = ((((lane & 15) >> 1) ^ f8) << 4) + (lane & 1) * 8;     // unit column of this lane's 8 elements
      const int mA = m0 + (c >> 6) * 128 + (c & 63);
      const uint32_t kr = kmapped ? 0u : (uint32_t)((wave * 8 + j * 4 + (lane >> 4)) * p.lda * 2);
      vo[0][j] = mA < p.M ? kr + (uint32_t)(mA * 2) : OOB;
      vo[3][j] = mA + 64 < p.M ? kr + (uint32_t)((mA + 64) * 2) : OOB;
    }
    if constexpr (!TB) {
      const int ur = wave * 16 + j * 8 + (lane >> 3);
      const int kc = (lane & 7) ^ (j * 4 + (lane >> 4));
      const int nB = n0 + (ur >> 5) * 64 + (ur & 31);
      vo[1][j] = nB < p.N ? (uint32_t)(((long long)nB * p.ldb + kc * 8) * 2) : OOB;
      vo[2][j] = nB + 32 < p.N ? (uint32_t)(((long long)(nB + 32) * p.ldb + kc * 8) * 2) : OOB;
    } else {
      const int f8 = (lane >> 4) | ((wave & 1) << 2);
      const int c = ((((lane & 15) >> 1) ^ f8) << 4) + (lane & 1) * 8;
      const int nB = n0 + (c >> 5) * 64 + (c & 31);
      const uint32_t kr = kmapped ? 0u : (uint32_t)((wave * 8 + j * 4 + (lane >> 4)) * p.ldb * 2);
      vo[1][j] = nB < p.N ? kr + (uint32_t)(nB * 2) : OOB;
      vo[2][j] = nB + 32 < p.N ? kr + (uint32_t)((nB + 32) * 2) : OOB;
    }
  }
  // mapped reduction rows (temporal-branch wgrad): physical row of this lane's k row of the K-tile being issued
  // Round 6: map_ktile is called for K-tiles 0, 1, 2, ... in order, so only K-tile 0 pays the division of map_row; every later tile
  // advances the lane's (remainder, physical row) pair by TK rows -- two adds and a compare per row instead of a 32-bit divide (~30 VALU
  // instructions each, 60 per K-tile and lane beside 64 MFMAs: the kmapped weight gradients ran 5-11 % behind their unmapped twins).
  uint32_t prow[2] = {0u, 0u}, krem[2] = {0u, 0u};
  auto map_first = [&]() {          // K-tile 0
    if constexpr (kmapped) {
      const uint32_t g = (uint32_t)p.kmap.group;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const uint32_t r = (uint32_t)(kbeg + wave * 8 + j * 4 + (lane >> 4));
        const uint32_t q = r / g;
        krem[j] = r - q * g;
        prow[j] = q * (uint32_t)p.kmap.stride + krem[j] + (uint32_t)p.kmap.offset;
      }
    }
  };
  auto map_next = [&]() {           // the K-tile behind the one mapped last
    if constexpr (kmapped) {
      const uint32_t g = (uint32_t)p.kmap.group;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        krem[j] += TK;
        prow[j] += TK;
        while (krem[j] >= g) {          // (at most once when the group is at least a K-tile long: 196 token rows per frame)
          krem[j] -= g;
          prow[j] += (uint32_t)p.kmap.stride - g;
        }
      }
    }
  };

  // issue the two DMA instructions of unit X (0..3) of K-tile kt into ring slot `slot`
  auto issue_unit = [&](auto X, int kt, int slot) {
    constexpr int x = decltype(X)::value;
    constexpr bool isA = (x == 0 || x == 3);
    constexpr bool T = isA ? TA : TB;
    const bool live = kt < nk;                                     // wave-uniform
    const uint32_t dead = live ? 0u : OOB;                         // units past the end: every lane out of range -> zero fill, no traffic
    const i32x4 r = isA ? ra : rb;
    const long long ld = isA ? p.lda : p.ldb;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const uint32_t dst = smem_base + (uint32_t)(slot * UNIT + (wave * 2 + j) * 1024);
#if defined(MPV_AB_NT_DMA_A) && defined(MPV_AB_NT_DMA_B)
      constexpr bool NT = true;
#elif defined(MPV_AB_NT_DMA_A)
      constexpr bool NT = isA;
#elif defined(MPV_AB_NT_DMA_B)
      constexpr bool NT = !isA;
#elif defined(MPV_AB_NT_DMA_ACT)      // the ACTIVATION operand(s): A always; B too unless it is a weight (forward / dgrad forms: B = W)
      constexpr bool NT = isA || (TA && TB);
#else
      constexpr bool NT = false;
#endif
      auto dma = [&](i32x4 rr, uint32_t d, uint32_t vv, uint32_t ss) {
        if constexpr (NT) dma16_nt(rr, d, vv, ss);
        else dma16(rr, d, vv, ss);
      };
      if constexpr (!T) {
        dma(r, dst, vo[x][j] | dead, live ? (uint32_t)((kbeg + kt * TK) * 2) : 0u);
      } else if constexpr (!kmapped) {
        dma(r, dst, vo[x][j] | dead, live ? (uint32_t)((long long)(kbeg + kt * TK) * ld * 2) : 0u);
      } else {
        dma(r, dst, (vo[x][j] == OOB ? OOB : vo[x][j] + (uint32_t)(prow[j] * ld * 2)) | dead, 0u);
      }
    }
  };

  // ---- per-lane LDS read offsets
  // k-contiguous image: fragment (blk, kk) of a wave = base + blk * 2048 (16 rows) with the k-half folded into the base
  // (one base VGPR per ring buffer: the DS immediate offset is 16 bits, and the second buffer starts at 64 KiB)
  const int sw = (lane >> 1) & 7;
  int a_off[2][2], b_off[2][2];       // !T: [buf][kk]
  int a_tr[2][4], b_tr[2][2];         // T: [buf][blk] (kk adds 8192, the +4 k rows add 1024)
#pragma unroll
  for (int bf = 0; bf < 2; ++bf) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      a_off[bf][kk] = bf * 4 * UNIT + (wr * 64 + l15) * 128 + (((kk * 4 + lg) ^ sw) << 4);
      b_off[bf][kk] = bf * 4 * UNIT + (wc * 32 + l15) * 128 + (((kk * 4 + lg) ^ sw) << 4);
      asm volatile("" : "+v"(a_off[bf][kk]), "+v"(b_off[bf][kk]));
    }
    const int f8 = (l15 >> 2) | ((lg & 1) << 2);
    const int krow = lg * 8 + (l15 >> 2);
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
      a_tr[bf][mb] = bf * 4 * UNIT + krow * 256 + ((((wr * 4 + mb) ^ f8) & 7) << 5) + (lane & 3) * 8;
      asm volatile("" : "+v"(a_tr[bf][mb]));
    }
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      b_tr[bf][nb] = bf * 4 * UNIT + krow * 256 + ((((wc * 2 + nb) ^ f8) & 7) << 5) + (lane & 3) * 8;
      asm volatile("" : "+v"(b_tr[bf][nb]));
    }
  }

  bf16x8 fa[4][2], fb[2][2][2];     // A fragments of the current 64-row half; B fragments of both 32-column halves
  auto read_a = [&](auto BUF, auto X) {     // unit X (0 or 3) of ring buffer BUF
    constexpr int bf = decltype(BUF)::value, x = decltype(X)::value;
#pragma unroll
    for (int mb = 0; mb < (x == 3 ? MB1 : 4); ++mb)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        if constexpr (!TA) fa[mb][kk] = *(const bf16x8*)(smem + a_off[bf][kk] + x * UNIT + mb * 2048);
        else fa[mb][kk] = lds_read_tr8(smem + a_tr[bf][mb] + x * UNIT + kk * 8192);
      }
  };
  auto read_b = [&](auto BUF, auto NH) {    // unit 1 + NH of ring buffer BUF
    constexpr int bf = decltype(BUF)::value, nh = decltype(NH)::value;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        if constexpr (!TB) fb[nh][nb][kk] = *(const bf16x8*)(smem + b_off[bf][kk] + (1 + nh) * UNIT + nb * 2048);
        else fb[nh][nb][kk] = lds_read_tr8(smem + b_tr[bf][nb] + (1 + nh) * UNIT + kk * 8192);
      }
  };

  f32x4 acc[2][2][4][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int d = 0; d < 2; ++d) acc[a][b][c][d] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fused bias gradient (wgrad, n-tile 0 only): wave column wc owns A row block mb == wc of both halves
  const bool do_colsum = TA && p.colsum_part != nullptr && n0 == 0;
  f32x4 cs_acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  bf16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (bf16)1.0f;

  // 4 MFMAs: row block mb of half mh against the 32-column half nh (both 16-column blocks, both k-steps)
  auto mma_blk = [&](auto MH, auto MB, auto NH) {
    constexpr int mh = decltype(MH)::value, mb = decltype(MB)::value, nh = decltype(NH)::value;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
        acc[mh][nh][mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[nh][nb][kk], fa[mb][kk], acc[mh][nh][mb][nb], 0, 0, 0);
  };
  auto mma = [&](auto MH, auto NH) {
    mma_blk(MH, IC<0>{}, NH);
    mma_blk(MH, IC<1>{}, NH);
    mma_blk(MH, IC<2>{}, NH);
    mma_blk(MH, IC<3>{}, NH);
  };
  // MFMA segment of phase j.  256-row tile: one 64 x 32 quadrant per phase, order (0,0) (0,1) (1,1) (1,0).  Short tiles:
  // the second half has only MB1 row blocks, so whole quadrants would give 16/16/4/4 (8/8) MFMAs over phases that cost the
  // same barriers and LDS latency; fragments stay in registers once read (A half 0: fa[0..3] from phase 0, fa[0..MB1-1]
  // replaced by half 1 in phase 2; B nh0 from phase 0, nh1 from phase 1), so row block 3 of half 0 is deferred to phases
  // 2 and 3: 12/12/8/8 (MB1 = 1) and 12/12/12/12 (MB1 = 2).
  auto mma_phase = [&](auto J) {
    constexpr int j = decltype(J)::value;
    if constexpr (MB1 == 4) {
      if constexpr (j == 0) mma(IC<0>{}, IC<0>{});
      else if constexpr (j == 1) mma(IC<0>{}, IC<1>{});
      else if constexpr (j == 2) mma(IC<1>{}, IC<1>{});
      else mma(IC<1>{}, IC<0>{});
    } else {
      if constexpr (j == 0) {
        mma_blk(IC<0>{}, IC<0>{}, IC<0>{});
        mma_blk(IC<0>{}, IC<1>{}, IC<0>{});
        mma_blk(IC<0>{}, IC<2>{}, IC<0>{});
      } else if constexpr (j == 1) {
        mma_blk(IC<0>{}, IC<0>{}, IC<1>{});
        mma_blk(IC<0>{}, IC<1>{}, IC<1>{});
        mma_blk(IC<0>{}, IC<2>{}, IC<1>{});
      } else if constexpr (j == 2) {
        mma_blk(IC<0>{}, IC<3>{}, IC<1>{});
        mma_blk(IC<1>{}, IC<0>{}, IC<1>{});
        if constexpr (MB1 == 2) mma_blk(IC<1>{}, IC<1>{}, IC<1>{});
      } else {
        mma_blk(IC<0>{}, IC<3>{}, IC<0>{});
        mma_blk(IC<1>{}, IC<0>{}, IC<0>{});
        if constexpr (MB1 == 2) mma_blk(IC<1>{}, IC<1>{}, IC<0>{});
      }
    }
  };
  auto colsum_mma = [&](auto MH) {
    constexpr int mh = decltype(MH)::value;
    if constexpr (TA) {
      if (do_colsum) {
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
          if (mb == wc) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) cs_acc[mh] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, fa[mb][kk], cs_acc[mh], 0, 0, 0);
          }
      }
    }
  };

  // one phase J (0..3) of K-tile t living in ring buffer BUF (0/1)
  auto phase = [&](auto J, auto BUF, int t) {
    constexpr int j = decltype(J)::value, buf = decltype(BUF)::value;
    // ---- load segment
    if constexpr (j == 0) {
      read_b(BUF, IC<0>{});
      SCHED_FENCE();
      read_a(BUF, IC<0>{});
    } else if constexpr (j == 1) {
      read_b(BUF, IC<1>{});
    } else if constexpr (j == 2) {
      read_a(BUF, IC<3>{});
    }
    SCHED_FENCE();
    // unit (4t + j + 6): K-tile t+1 for j < 2 (other buffer), t+2 for j >= 2 (this buffer); unit-in-buffer (j + 2) & 3
    if constexpr (j == 2) map_next();      // K-tile t + 2
    issue_unit(IC<((j + 2) & 3)>{}, t + (j < 2 ? 1 : 2), (j < 2 ? (buf ^ 1) : buf) * 4 + ((j + 2) & 3));
    __builtin_amdgcn_s_waitcnt(0x0F76);   // vmcnt(6): everything but the three newest units of this wave has landed
    SCHED_FENCE();
    __builtin_amdgcn_s_barrier();
    SCHED_FENCE();
    // ---- MFMA segment: quadrant order (0,0) (0,1) (1,1) (1,0)
    __builtin_amdgcn_s_setprio(1);
    mma_phase(J);
    if constexpr (j == 0) colsum_mma(IC<0>{});
    else if constexpr (j == 2) colsum_mma(IC<1>{});
    __builtin_amdgcn_s_setprio(0);
    SCHED_FENCE();
    __builtin_amdgcn_s_barrier();
    SCHED_FENCE();
  };
  auto ktile = [&](auto BUF, int t) {
    phase(IC<0>{}, BUF, t);
    phase(IC<1>{}, BUF, t);
    phase(IC<2>{}, BUF, t);
    phase(IC<3>{}, BUF, t);
  };

  // ---- the tile's bias slice goes to LDS by one DMA instruction of wave 0, ahead of everything else (so it is the oldest request:
  // every counted wait below retires it, and the prologue barrier publishes it).  The epilogue used to fetch it from global memory
  // where it needs it -- eight dependent L2 round trips per tile, each behind its own s_waitcnt vmcnt(0), inside the staging pass.
  const bool lds_bias = p.bias != nullptr && !p.out_f32;
  if (lds_bias && wave == 0) {
    const int nb0 = n0 + lane * 8;
    dma16(raw_rsrc(p.bias, (uint32_t)p.N * 2u), smem_base + BIAS_LDS, (lane < 32 && nb0 < p.N) ? (uint32_t)(nb0 * 2) : OOB, 0u);
  }

  // ---- prologue: units 0..5 (K-tile 0 and U0, U1 of K-tile 1)
  map_first();
  issue_unit(IC<0>{}, 0, 0);
  issue_unit(IC<1>{}, 0, 1);
  issue_unit(IC<2>{}, 0, 2);
  issue_unit(IC<3>{}, 0, 3);
  map_next();       // K-tile 1
  issue_unit(IC<0>{}, 1, 4);
  issue_unit(IC<1>{}, 1, 5);
  // phase 0 reads U0 and U1 only, so it starts once THEY have landed (vmcnt(8): everything but the four newest units of this
  // wave); U2 / U3 of K-tile 0 are retired by phase 0's own vmcnt(6) + barrier, one full barrier pair before phase 1 / 2 read
  // them -- also for wave row 0, whose phase-0 MFMA barrier is wave row 1's phase-0 wait barrier (the stagger below)
  __builtin_amdgcn_s_waitcnt(0x0F78);
  SCHED_FENCE();
  __builtin_amdgcn_s_barrier();           // ... and everybody else's
  SCHED_FENCE();
  if (wr == 1) __builtin_amdgcn_s_barrier();   // stagger: wave row 1 runs one barrier behind wave row 0
  SCHED_FENCE();

  int t = 0;
  for (; t + 2 <= nk; t += 2) {
    ktile(IC<0>{}, t);
    ktile(IC<1>{}, t + 1);
  }
  if (t < nk) ktile(IC<0>{}, t);
  SCHED_FENCE();
  __builtin_amdgcn_s_waitcnt(0x0F70);     // vmcnt(0): the zero-fill DMAs of the dead units are done before LDS is reused
  if (wr == 0) __builtin_amdgcn_s_barrier();   // un-stagger
  SCHED_FENCE();
  __syncthreads();

  // ------------------------------------------------------------------------------------------------ epilogue
  const float alpha = p.alpha * (p.alpha_dev ? *p.alpha_dev : 1.0f);
  if (alpha != 1.0f) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int c = 0; c < (a == 1 ? MB1 : 4); ++c)
#pragma unroll
          for (int d = 0; d < 2; ++d) acc[a][b][c][d] *= alpha;
  }

  if constexpr (TA) {
    if (do_colsum) {
      // D = ones x Afrag: every D row i holds the same column sums; lane (l15, lg) reg r = D[4*lg + r][m = l15]
      if (lg == 0) {
#pragma unroll
        for (int mh = 0; mh < 2; ++mh) {
          const int m = m0 + wr * 128 + mh * 64 + wc * 16 + l15;
          if (m < p.M) p.colsum_part[(long long)split * p.M + m] = cs_acc[mh][0];
        }
      }
    }
  }

  if (p.out_f32) {
    float* cs = (float*)smem;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int wrq = q >> 1;
      if (wr == wrq) {
#pragma unroll
        for (int nh = 0; nh < 2; ++nh)
#pragma unroll
          for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
              *(f32x4*)(cs + (mb * 16 + l15) * FPITCH + wc * 64 + nh * 32 + nb * 16 + lg * 4) = (q & 1) ? acc[1][nh][mb][nb] : acc[0][nh][mb][nb];
            }
      }
      __syncthreads();
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int c = tid + 512 * it;
        const int row = c >> 6, col = (c & 63) * 4;
        const int m = m0 + wrq * 128 + (q & 1) * 64 + row, n = n0 + col;
        if (m < p.M && n < p.N) {
          f32x4 v = *(const f32x4*)(cs + row * FPITCH + col);
          float* cp = (float*)p.C + (long long)split * p.M * p.N + map_row(p.cmap, m) * p.ldc + n;
          if (p.accumulate) v += *(const f32x4*)cp;
#ifdef MPV_AB_NT_F32
          __builtin_nontemporal_store(v, (f32x4*)cp);
#else
          *(f32x4*)cp = v;      // (split-K partials are read back by the reduce kernel right behind this launch: they stay cacheable)
#endif
        }
      }
      __syncthreads();
    }
    return;
  }

  // Row-contiguous finish, specialised at compile time for the epilogue combinations the step uses (a tile is 128
  // elements per thread with the matrix pipe idle: per-element runtime switches on act / act_bwd cost more than the
  // activation itself); anything else takes the generic instance with the switches hoisted to one per 8-element chunk.
  auto epilogue = [&](auto kind) {
    constexpr int KIND = decltype(kind)::value;
    EpExt<KIND, NIT> ext;
    prefetch_rows<KIND, NIT>(p, ext, tid, m0, n0);      // global reads of the finish go out first
    bf16* cb = (bf16*)smem;
#pragma unroll
    for (int mh = 0; mh < 2; ++mh)
#pragma unroll
      for (int nh = 0; nh < 2; ++nh)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
          const int col = wc * 64 + nh * 32 + nb * 16 + lg * 4;
          f32x4 bv = {0.f, 0.f, 0.f, 0.f};       // (past column N: the DMA's zero fill)
          if (lds_bias) bv = cvt4(*(const bf16x4*)(smem + BIAS_LDS + col * 2));
#pragma unroll
          for (int mb = 0; mb < (mh == 1 ? MB1 : 4); ++mb) {
            const int row = wr * WROWS + mh * 64 + mb * 16 + l15;
            bf16x4 zb = cvt4(acc[mh][nh][mb][nb] + bv);
            if constexpr (KIND == EP_PLAIN) {      // mpv.h colscale: columns below colscale_cols are rounded, scaled and rounded again
              if (n0 + col < p.colscale_cols) zb = cvt4(cvt4(zb) * p.colscale);
            }
            *(bf16x4*)(cb + row * CPITCH + col) = zb;
          }
        }
    __syncthreads();     // the staged tile is complete
    finish_rows<KIND, NIT>(p, ext, cb, tid, m0, n0);
  };
  const int cfg = (p.act ? 1 : 0) | (p.act_bwd ? 2 : 0) | (p.drop_thr ? 4 : 0) | (p.residual ? 8 : 0) | (p.accumulate ? 16 : 0) |
                  (p.preact ? 32 : 0) | (p.tap_out && !(p.residual && !p.act && !p.act_bwd && !p.drop_thr && !p.accumulate && !p.preact) ? 64 : 0);
  if (cfg == 0) epilogue(IC<EP_PLAIN>{});
  else if (cfg == (1 | 32) && p.act == MPV_ACT_GELU_ERF) epilogue(IC<EP_ERF_PRE>{});
  else if (cfg == (1 | 32) && p.act == MPV_ACT_GELU_TANH) epilogue(IC<EP_TANH_PRE>{});
  else if (cfg == 2 && p.act_bwd == MPV_ACT_GELU_ERF) epilogue(IC<EP_BWD_ERF>{});
  else if (cfg == 2 && p.act_bwd == MPV_ACT_GELU_TANH) epilogue(IC<EP_BWD_TANH>{});
  else if (cfg == 2 && p.act_bwd == MPV_ACT_DERIV) epilogue(IC<EP_MUL>{});
  else if (cfg == 8) epilogue(IC<EP_RES>{});
  else if (cfg == (4 | 8)) epilogue(IC<EP_DROP_RES>{});
  else if (cfg == 4) epilogue(IC<EP_DROP>{});
  else epilogue(IC<EP_GENERIC>{});
}

struct Band { int rows, m_tiles; };
// Row-band plan of one product (see mpv_gemm256_try_launch): fills plan[0..n) with (tile rows, m-tiles) in row order.
int plan_bands(int M, int tn, int K, float epi_tiles, int ncu, Band plan[3]) {
  // MPV_BAND_*: measurement overrides of the model constants (percent / rounds), read once
  static const float rel192 = [] { const char* e = getenv("MPV_BAND_REL192"); return e ? atoi(e) * 0.01f : 0.80f; }();
  static const float rel160 = [] { const char* e = getenv("MPV_BAND_REL160"); return e ? atoi(e) * 0.01f : 0.73f; }();
  static const float thr = [] { const char* e = getenv("MPV_BAND_THR"); return e ? atoi(e) * 0.01f : 0.97f; }();
  static const int maxr = [] { const char* e = getenv("MPV_BAND_MAXR"); return e ? atoi(e) : 4; }();
  const float nk = (float)(K / TK);
  const float epi = 4.9f * epi_tiles;
  auto tile_us = [&](int r) { return 2.2f + 1.31f * nk * (r == 256 ? 1.0f : r == 192 ? rel192 : rel160) + epi * (float)r * (1.0f / 256.0f); };
  auto band_us = [&](int r, int mt) { return mt > 0 ? (float)(((long long)mt * tn + ncu - 1) / ncu) * tile_us(r) + 2.5f : 0.f; };
  const int all256 = (M + 255) / 256;
  int nband = 1;
  plan[0] = Band{256, all256};
  const float single256 = band_us(256, all256);
  float best = single256;
  // measured (tools/gemm_ab.py, 256 CUs): launches of a few rounds gain 2-5 % (N = 768 ViT shapes at 2.31 rounds, the GPT
  // N = 8192 shapes at 2.5); at ~9 rounds (N = 3072) the workgroups have drifted apart, the last round is no longer
  // paid in full and cutting the launch only adds tails (-2..-4 %): single launch from 5 rounds on
  if (((long long)all256 * tn + ncu - 1) / ncu > maxr) return 1;
  // candidates: n256 m-tiles of 256 rows filling whole rounds, then n192 m-tiles of 192 rows likewise, the rest at 160 / 192 / 256
  for (int r1 = 0;; ++r1) {
    int n256 = (int)((long long)r1 * ncu / tn);
    const bool last1 = n256 >= M / 256;
    if (last1) n256 = M / 256;                              // whole tiles only: what is left goes to the next band
    const int m1 = M - n256 * 256;
    const float c1 = band_us(256, n256);
    for (int r2 = 0;; ++r2) {
      int n192 = (int)((long long)r2 * ncu / tn);
      const bool last2 = n192 >= m1 / 192;
      if (last2) n192 = m1 / 192;
      const int m2 = m1 - n192 * 192;
      const float c2 = c1 + band_us(192, n192);
      for (int rr : {160, 192, 256}) {
        const int nr = (m2 + rr - 1) / rr;
        const float c = c2 + band_us(rr, nr);
        if (c < best - 0.01f) {
          best = c;
          nband = 0;
          if (n256) plan[nband++] = Band{256, n256};
          if (n192) plan[nband++] = Band{192, n192};
          if (nr) plan[nband++] = Band{rr, nr};
        }
      }
      if (last2) break;
    }
    if (last1) break;
  }
  if (best > thr * single256) {                           // not worth the extra launches
    nband = 1;
    plan[0] = Band{256, all256};
  }
  for (int i = 0; i + 1 < nband;) {                         // adjacent bands of equal tile rows are one launch
    if (plan[i].rows == plan[i + 1].rows) {
      plan[i].m_tiles += plan[i + 1].m_tiles;
      for (int j = i + 1; j + 1 < nband; ++j) plan[j] = plan[j + 1];
      --nband;
    } else {
      ++i;
    }
  }
  return nband;
}

}  // namespace

// test hook: the row-band plan of an M x N x K product on `ncu` compute units; out[2 * i] = tile rows, out[2 * i + 1] = m-tiles
extern "C" int mpv_gemm_plan_bands(int64_t M, int64_t N, int64_t K, int ncu, int preact, int ext_rows, int* out6) {
  MPV_REQUIRE(M > 0 && N > 0 && K > 0 && ncu > 0 && out6, MPV_E_ARG, "mpv_gemm_plan_bands: bad argument");
  Band plan[3] = {};
  const int n = plan_bands((int)M, (int)((N + TN - 1) / TN), (int)K, 1.0f + (preact ? 1.0f : 0.f) + (ext_rows ? 0.5f : 0.f), ncu, plan);
  for (int i = 0; i < 3; ++i) {
    out6[2 * i] = i < n ? plan[i].rows : 0;
    out6[2 * i + 1] = i < n ? plan[i].m_tiles : 0;
  }
  return n;
}

// Takes the problem if the 256x256 kernel is expected to beat the 128x128 one on it.  g is fully populated by
// mpv_gemm_bf16 (epilogue, maps, byte extents, split-K fields for wgrad: splits / k_per_split / C = fp32 workspace).
bool mpv_gemm256_try_launch(const GemmArgs& g0, int transA, int transB, hipStream_t stream) {
  GemmArgs g = g0;
  if ((g.K % TK != 0 && !(transA && transB)) || g.k_per_split % TK != 0) return false;
  if (g.tail_g > 1) return false;
  if (g.bias && ((uintptr_t)g.bias & 15) != 0) return false;      // the bias slice of a tile is fetched by 16-byte LDS-DMA
  g.tiles_n = (g.N + TN - 1) / TN;
  // MPV_GEMM_GM / MPV_GEMM_BANDS=0: measurement knobs (XCD walk width; row bands off), read once
  static const int env_gm = [] { const char* e = getenv("MPV_GEMM_GM"); return e ? atoi(e) : 0; }();
  static const bool env_bands = [] { const char* e = getenv("MPV_GEMM_BANDS"); return !e || atoi(e) != 0; }();
  g.gm = g.gm > 0 ? g.gm : env_gm > 0 ? env_gm : 4;
  const bool km = (transA || transB) && g.kmap.group != 0;
  // Tile rows.  A launch runs in rounds of one workgroup per CU, and a partly filled last round costs a whole tile time
  // (M = 50432, N = 768: 591 tiles = 2.31 rounds on 256 CUs, paid as 3).  With a k-contiguous A operand (forward and
  // dgrad forms, bf16 output, no split-K) the rows are therefore cut into up to three BANDS, each its own launch of the
  // same kernel at 256, 192 or 160 tile rows (same ring, barriers and epilogue; m_base / M delimit the band), sized so that
  // the 256- and 192-row bands are whole rounds and the short tiles take the remainder: 1 + 1 + 1 rounds at relative
  // tile times 1.0 / 0.8 / 0.73 instead of 3 x 1.0.  Cost model = measured tile anatomy (tools/probe/gemm256_timeline.py):
  // prologue 2.2 us + 1.31 us per K-tile x {1.0, 0.80, 0.73} + store-bound epilogue 4.9 us x rows / 256 x (tiles written
  // or read besides C), and 2.5 us per extra launch.
  const bool bandable = !transA && !g.out_f32 && g.splits == 1 && !(transB && km);
  Band plan[3] = {{256, (g.M + 255) / 256}, {0, 0}, {0, 0}};
  int nband = 1;
  if (bandable && (g.tile_rows == 192 || g.tile_rows == 160)) {
    plan[0] = Band{g.tile_rows, (g.M + g.tile_rows - 1) / g.tile_rows};
  } else if (bandable && g.tile_rows == 0 && env_bands) {
    static int ncu = 0;
    if (!ncu) {
      int dev = 0;
      hipDeviceProp_t prop;
      ncu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
    }
    nband = plan_bands(g.M, g.tiles_n, g.K, 1.0f + (g.preact ? 1.0f : 0.f) + ((g.residual || g.act_bwd) ? 0.5f : 0.f), ncu, plan);
  }
  const int M_all = g.M;
  int m_base = 0;
  for (int b = 0; b < nband; ++b) {
    const int rows = plan[b].rows;
    g.m_base = m_base;
    g.M = std::min(M_all, m_base + plan[b].m_tiles * rows);
    g.tiles_m = plan[b].m_tiles;
    g.nwg = g.tiles_m * g.tiles_n;
    m_base = g.M;
    const dim3 grid((unsigned)(g.nwg * g.splits)), block(512);
    if (!transA && !transB) {
      if (rows == 160) hipLaunchKernelGGL((gemm256_kernel<false, false, false, 1>), grid, block, 0, stream, g);
      else if (rows == 192) hipLaunchKernelGGL((gemm256_kernel<false, false, false, 2>), grid, block, 0, stream, g);
      else hipLaunchKernelGGL((gemm256_kernel<false, false, false>), grid, block, 0, stream, g);
    } else if (!transA && transB && !km) {
      if (rows == 160) hipLaunchKernelGGL((gemm256_kernel<false, true, false, 1>), grid, block, 0, stream, g);
      else if (rows == 192) hipLaunchKernelGGL((gemm256_kernel<false, true, false, 2>), grid, block, 0, stream, g);
      else hipLaunchKernelGGL((gemm256_kernel<false, true, false>), grid, block, 0, stream, g);
    } else if (!transA && transB)
      hipLaunchKernelGGL((gemm256_kernel<false, true, true>), grid, block, 0, stream, g);
    else if (!km)
      hipLaunchKernelGGL((gemm256_kernel<true, true, false>), grid, block, 0, stream, g);
    else
      hipLaunchKernelGGL((gemm256_kernel<true, true, true>), grid, block, 0, stream, g);
  }
  return true;
}
