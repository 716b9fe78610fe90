// gemm.hip -- bf16 MFMA GEMM family for gfx950 (CDNA4), hand-written.
//
//   C[M,N] = epilogue( sum_k Aop(m,k) * Bop(n,k) )
//
// Operand storage ("T" = the reduction index is the SLOW dimension in memory):
//   TA=0: A stored [M][K] (k contiguous, lda)      TA=1: A stored [K][M] (m contiguous, lda)
//   TB=0: B stored [N][K] (nn.Linear weight, ldb)  TB=1: B stored [K][N] (n contiguous, ldb)
// which covers the three passes of a Linear layer without any transposed copies:
//   forward  Y = X W^T          : TA=0, TB=0   (replaces cuBLAS F.linear, reference
//                                               models/vision_transformer.py:104,108,175,205,250;
//                                               models/modeling_distributed_gpt3.py:562,573,843,852,1348)
//   dgrad    dX = dY W          : TA=0, TB=1   (reduction over N; W is [N][K] = "[red][out]")
//   wgrad    dW = dY^T X        : TA=1, TB=1   (reduction over rows)
//
// Structure: 128x128 block tile, BK=64, 256 threads = 4 waves (2x2), each wave 64x64 =
// 2x2 v_mfma_f32_32x32x16_bf16 tiles.  Global->LDS staging goes through registers with
// buffer loads (out-of-range chunks read as zero, so ragged M/N/K edges need no branches),
// issued for tile k+1 before the MFMAs of tile k and written to the other LDS buffer after
// them (one barrier per K step).  k-contiguous operands sit in LDS as [row][64] with a
// 16-byte-chunk XOR swizzle and are read with ds_read_b128; reduction-slow operands sit as
// [k][128] with a 64-byte-chunk XOR swizzle and are read with ds_read_b64_tr_b16 (the
// gfx950 transposing LDS read), so neither layout bank-conflicts.  The MFMA is issued with
// swapped operands (D = Bfrag x Afrag) so every lane ends up owning 4 consecutive output
// columns of one output row -> 8-byte packed bf16 stores and a cheap fused epilogue
// (bias, erf/tanh GELU (+pre-activation copy), GELU-backward multiply, dropout, residual).
// Workgroup ids are remapped so that each XCD (private L2) walks a contiguous range of tiles.
#include <stdlib.h>

#include "mpv_common.h"
#include "../../include/mpv.h"
#include "gemm_args.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = 128 * 64 * 2;  // 16 KiB per operand tile
constexpr unsigned TAIL_WS_BYTES_DEV = 512u * 128 * 128 * 4;   // partial-tile workspace of the tail split (32 MiB)


// LDS byte address of 16-byte chunk `kc` (0..7) of `row` in a k-contiguous [128][64] tile.
__device__ __forceinline__ int lds_addr_kc(int row, int kc) { return row * 128 + ((kc ^ ((row >> 1) & 7)) << 4); }
// LDS byte address of byte column `cb` (0..255) of k-row `krow` in a [64][128] tile.
__device__ __forceinline__ int lds_addr_tr(int krow, int cb) {
  return krow * 256 + ((((cb >> 6) ^ (krow & 3))) << 6) + (cb & 63);
}

typedef __attribute__((address_space(3))) void lds_void;

// Per-thread addressing of the LDS-DMA staging (buffer_load_dwordx4 ... lds): one wave instruction
// fills 1 KiB of LDS lane-linearly (lane l -> +16*l), so the LDS image is fixed and the XOR
// swizzles are applied to the per-lane GLOBAL source address instead (and again on the read side).
// Wave w issues instructions j=0..3 of each operand tile; instruction (w,j) covers LDS bytes
// [(4w+j)*1024, +1024) = 8 rows of a k-contiguous tile or 4 k-rows of a reduction-slow tile.
template <bool T>
struct Loader {
  uint32_t off[4];   // byte offset of this lane's 16-byte chunk, minus the k0-dependent part
  bool ok[4];

  __device__ __forceinline__ void init_kc(int lane, int wave, int row0, int R, long long ld, RowMap map) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = (wave * 4 + j) * 8 + (lane >> 3);
      const int kc = (lane & 7) ^ ((row >> 1) & 7);
      const long long gr = row0 + row;
      ok[j] = gr < R;
      off[j] = (uint32_t)((map_row(map, ok[j] ? gr : 0) * ld + kc * 8) * 2);
    }
  }
  // register-staged k-contiguous operand (used beside a reduction-slow operand): thread t owns
  // rows (t>>3)+32i, 16-byte chunk t&7
  __device__ __forceinline__ void init_kc_reg(int tid, int row0, int R, long long ld, RowMap map) {
    const int kc = tid & 7;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = (tid >> 3) + 32 * i;
      const long long gr = row0 + row;
      ok[i] = gr < R;
      off[i] = (uint32_t)((map_row(map, ok[i] ? gr : 0) * ld + kc * 8) * 2);
      ldsw[i] = lds_addr_kc(row, kc);
    }
  }
  __device__ __forceinline__ int kc_of(int lane, int wave, int j) const {
    const int row = (wave * 4 + j) * 8 + (lane >> 3);
    return (lane & 7) ^ ((row >> 1) & 7);
  }
  // reduction-slow operands are staged through registers (thread t: k-rows (t>>4)+16i, chunk t&15)
  int ldsw[4];
  __device__ __forceinline__ void init_tr(int tid, int col0, int Cn) {
    const int cc = tid & 15;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int krow = (tid >> 4) + 16 * i;
      ok[i] = (col0 + cc * 8) < Cn;
      off[i] = (uint32_t)((col0 + cc * 8) * 2);
      ldsw[i] = lds_addr_tr(krow, cc * 16);
    }
  }
};

// `batch` independent products of ONE shape in one launch (blockIdx.y = problem; mpv_gemm_bf16_batched): the [D, D] chain-rule products of
// the composed temporal projection -- Wc = Wf Wp of every ViT block at the head of the step, dWf' = dWc Wp^T and dWp = Wf^T dWc of every
// block at the end of the tower's backward -- were 36 launches of 36 workgroups each (768^3: 42-50 TFLOP/s, 0.7 ms per step); as three
// launches of 12 x 36 workgroups they fill the chip once.  The operand pointers of the problems travel by value in the launch
// (GemmArgs.batch > 0 selects them; an ordinary launch passes an empty table and a grid of height 1).
constexpr int GEMM_BATCH_MAX = 16;
struct GemmBatchPtrs {
  const bf16* a[GEMM_BATCH_MAX];
  const bf16* b[GEMM_BATCH_MAX];
  bf16* c[GEMM_BATCH_MAX];
};

template <bool TA, bool TB>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(const GemmArgs p, const GemmBatchPtrs bptr) {
  const uint64_t seed_r = p.drop_thr ? mpv_resolve_seed(p.seed) : 0;      // (bit 63 set: the seed lives in device memory, mpv_common.h)
  __shared__ __attribute__((aligned(1024))) char smem[4 * TILE_BYTES];
  const bf16* const opA = p.batch ? bptr.a[blockIdx.y] : p.A;
  const bf16* const opB = p.batch ? bptr.b[blockIdx.y] : p.B;
  void* const opC = p.batch ? (void*)bptr.c[blockIdx.y] : p.C;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wrow = wave >> 1, wcol = wave & 1;

  // XCD-aware bijective remap: workgroup b runs on XCD b%8; give each XCD a contiguous tile range.
  int bid = blockIdx.x;
  int part = 0, parts = 1, tail_tile = 0;
  if (p.tail_g > 1 && bid >= p.tail_start) {
    const int u = bid - p.tail_start;
    tail_tile = u / p.tail_g;
    part = u - tail_tile * p.tail_g;
    parts = p.tail_g;
    bid = p.tail_start + tail_tile;
  }
  // Split-K launches are 1-D too, split-major: an XCD's contiguous range then lies inside one or two K-splits, so
  // the workgroups sharing its L2 stream the SAME rows of both operands (measured on the wgrad shapes: the
  // (x = tile, y = split) grid spread every XCD over all splits and re-fetched ~2.6x the algorithmic bytes).
  const int nwg = p.nwg * p.splits;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
  const int lin = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int split = lin / p.nwg;
  const int pid = lin - split * p.nwg;
  // L2 blocking inside an XCD's range: walk GM m-tiles per n-tile, so the ~64 workgroups resident on an
  // XCD cover an ~8x8 patch of tiles (8 A panels + 8 B panels, each reused 8x from the 4 MiB L2)
  constexpr int GM = 8;
  const int gsz = GM * p.tiles_n;
  const int grp = pid / gsz, rem = pid - grp * gsz;
  const int gm = min(GM, p.tiles_m - grp * GM);
  const int tile_n = rem / gm, tile_m = grp * GM + (rem - tile_n * gm);
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const int kbeg = parts > 1 ? part * p.tail_steps * BK : split * p.k_per_split;
  const int kend = min(p.K, kbeg + (parts > 1 ? p.tail_steps * BK : p.k_per_split));
  const int nk = kend > kbeg ? (kend - kbeg + BK - 1) / BK : 0;

  const __amdgpu_buffer_rsrc_t ra_src = make_rsrc(opA, p.a_bytes);
  const __amdgpu_buffer_rsrc_t rb_src = make_rsrc(opB, p.b_bytes);

  Loader<TA> la;
  Loader<TB> lb;
  constexpr bool DMA = !TA && !TB;   // forward pass: both operands by LDS-DMA; dgrad/wgrad: register staging
  if constexpr (TA) la.init_tr(tid, m0, p.M);
  else if constexpr (DMA) la.init_kc(lane, wave, m0, p.M, p.lda, p.amap);
  else la.init_kc_reg(tid, m0, p.M, p.lda, p.amap);
  if constexpr (TB) lb.init_tr(tid, n0, p.N);
  else lb.init_kc(lane, wave, n0, p.N, p.ldb, RowMap{0, 0, 0});
  i32x4 sa[DMA ? 1 : 4], sb[DMA ? 1 : 4];

  // k-contiguous operands: LDS-DMA of K-tile k0 into LDS buffer `buf` (4 x 1 KiB per wave per operand);
  // reduction-slow operands: buffer loads into registers (committed to LDS after the MFMAs)
  auto issue = [&](int k0, int buf) __attribute__((always_inline)) {
    char* pa = smem + buf * 2 * TILE_BYTES + wave * 4096;
    char* pb = pa + TILE_BYTES;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if constexpr (TA) {
        const int kr = k0 + (tid >> 4) + 16 * j;
        const uint32_t va = (la.ok[j] && kr < kend) ? (uint32_t)(map_row(p.kmap, kr) * p.lda * 2) + la.off[j] : 0x80000000u;
        sa[j] = __builtin_amdgcn_raw_buffer_load_b128(ra_src, va, 0, 0);
      } else if constexpr (DMA) {
        const uint32_t va = (la.ok[j] && (k0 + la.kc_of(lane, wave, j) * 8) < kend) ? la.off[j] + (uint32_t)(k0 * 2) : 0x80000000u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra_src, (lds_void*)(pa + j * 1024), 16, va, 0, 0, 0);
      } else {
        const uint32_t va = (la.ok[j] && (k0 + (tid & 7) * 8) < kend) ? la.off[j] + (uint32_t)(k0 * 2) : 0x80000000u;
        sa[j] = __builtin_amdgcn_raw_buffer_load_b128(ra_src, va, 0, 0);
      }
      if constexpr (TB) {
        const int kr = k0 + (tid >> 4) + 16 * j;
        const uint32_t vb = (lb.ok[j] && kr < kend) ? (uint32_t)(map_row(p.kmap, kr) * p.ldb * 2) + lb.off[j] : 0x80000000u;
        sb[j] = __builtin_amdgcn_raw_buffer_load_b128(rb_src, vb, 0, 0);
      } else {
        const uint32_t vb = (lb.ok[j] && (k0 + lb.kc_of(lane, wave, j) * 8) < kend) ? lb.off[j] + (uint32_t)(k0 * 2) : 0x80000000u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rb_src, (lds_void*)(pb + j * 1024), 16, vb, 0, 0, 0);
      }
    }
  };
  // fused bias gradient (wgrad, first n-tile only): column sums of the A operand (dY) ride along with its
  // staging registers -- each thread owns 8 columns x 4 k-rows per K step
  f32x8 csum = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const bool do_colsum = TA && p.colsum_part != nullptr && n0 == 0;
  auto commit = [&](int buf) __attribute__((always_inline)) {
    char* pa = smem + buf * 2 * TILE_BYTES;
    char* pb = pa + TILE_BYTES;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if constexpr (!DMA) {
        *(i32x4*)(pa + la.ldsw[j]) = sa[j];
        *(i32x4*)(pb + lb.ldsw[j]) = sb[j];
      }
      if constexpr (TA) {
        if (do_colsum) {
          union { i32x4 i; bf16x8 b; } u;
          u.i = sa[j];
          csum += cvt8(u.b);
        }
      }
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // fragment fetch: 8 bf16 along k for output index (lane&31) of 32-wide sub-tile `t`
  auto frag = [&](const char* base, bool tr, int sub0, int s) __attribute__((always_inline)) -> bf16x8 {
    if (!tr) {
      const int row = sub0 + (lane & 31);
      const int chunk = s * 2 + (lane >> 5);
      return *(const bf16x8*)(base + lds_addr_kc(row, chunk));
    } else {
      const int kr = s * 16 + (lane >> 5) * 8 + ((lane & 15) >> 2);
      const int cb = (sub0 + ((lane >> 4) & 1) * 16 + (lane & 3) * 4) * 2;
      typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + lds_addr_tr(kr, cb)));
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + lds_addr_tr(kr + 4, cb)));
      union {
        struct { s16x4 a, b; } s;
        bf16x8 v;
      } u;
      u.s.a = lo;
      u.s.b = hi;
      return u.v;
    }
  };

  if (nk > 0) {
    issue(kbeg, 0);
    commit(0);
  }
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    // tile kt has landed (every wave drains its own DMA, then the barrier), and every wave is done
    // reading the other buffer (its reads belong to iteration kt-1)
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    __syncthreads();
    if (kt + 1 < nk) issue(kbeg + (kt + 1) * BK, cur ^ 1);
    const char* pa = smem + cur * 2 * TILE_BYTES;
    const char* pb = pa + TILE_BYTES;
    // software-pipelined fragment fetch: the ds_reads of k-sub s+1 are in flight under the MFMAs of k-sub s
    bf16x8 fa[2][2], fb[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      fa[0][t] = frag(pa, TA, wrow * 64 + t * 32, 0);
      fb[0][t] = frag(pb, TB, wcol * 64 + t * 32, 0);
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if (s < 3) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          fa[(s + 1) & 1][t] = frag(pa, TA, wrow * 64 + t * 32, s + 1);
          fb[(s + 1) & 1][t] = frag(pb, TB, wcol * 64 + t * 32, s + 1);
        }
      }
      __builtin_amdgcn_sched_barrier(0);   // keep the next k-sub's reads ahead of this k-sub's MFMAs
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[s & 1][j], fa[s & 1][i], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (TA || TB) {
      if (kt + 1 < nk) commit(cur ^ 1);
    }
  }

  if constexpr (TA) {
    if (do_colsum) {     // 16 threads (tid>>4) share a column chunk (tid&15): reduce through LDS, 128 columns per tile
      __syncthreads();
      float* red = (float*)smem;
#pragma unroll
      for (int e = 0; e < 8; ++e) red[(tid >> 4) * 128 + (tid & 15) * 8 + e] = csum[e];
      __syncthreads();
      if (tid < 128 && m0 + tid < p.M) {
        float a = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) a += red[r * 128 + tid];
        p.colsum_part[(long long)split * p.M + m0 + tid] = a;
      }
      // the epilogue starts with its own __syncthreads() before reusing smem
    }
  }
  // ------------------------------------------------------------------ epilogue
  // The MFMA layout gives each lane 4 columns of 32 different rows: stored directly that is 32
  // partial cache lines per store instruction (measured: ~6.7 us fixed cost per tile, independent of
  // the output dtype).  Instead the fp32 accumulators go through LDS (two 64-row halves, pitch 132
  // floats) and every thread finishes 8 consecutive columns of one row: bias / activation /
  // GELU-backward / dropout / residual loads and the final store are all 16-byte, row-contiguous.
  const float alpha = p.alpha * (p.alpha_dev ? *p.alpha_dev : 1.0f);
  constexpr int CP = 132;
  float* cs = (float*)smem;
  __syncthreads();   // every wave is done with the operand tiles
  // Plain epilogue (bias only, bf16 out: the qkv / projection GEMMs and every plain dgrad): add the bias in the MFMA
  // layout, round to bf16 in registers and stage the WHOLE tile once as bf16 (pitch 136 elements) -- half the LDS
  // traffic and one barrier pair instead of two; the stores stay 16-byte row-contiguous.
  if (parts == 1 && !p.out_f32 && !p.act && !p.act_bwd && !p.residual && !p.drop_thr && !p.accumulate && !p.preact) {
    constexpr int BP = 136;
    bf16* cb = (bf16*)smem;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int col = wcol * 64 + j * 32 + 8 * q + 4 * (lane >> 5);
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * q + e] * alpha;
          if (p.bias && n0 + col < p.N) v += cvt4(*(const bf16x4*)(p.bias + n0 + col));
          bf16x4 zb = cvt4(v);
          if (n0 + col < p.colscale_cols) zb = cvt4(cvt4(zb) * p.colscale);      // (mpv.h colscale: a second rounding, as `q * scale` on a bf16 q)
          *(bf16x4*)(cb + (wrow * 64 + i * 32 + (lane & 31)) * BP + col) = zb;
        }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int c = tid + 256 * it;
      const int row = c >> 4, col = (c & 15) * 8;
      const int m = m0 + row, n = n0 + col;
      if (m < p.M && n < p.N) *(bf16x8*)((bf16*)opC + map_row(p.cmap, m) * p.ldc + n) = *(const bf16x8*)(cb + row * BP + col);
    }
    return;
  }
  auto stage = [&](int half) __attribute__((always_inline)) {
    if (wrow == half) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * q + e];
            *(f32x4*)(cs + (i * 32 + (lane & 31)) * CP + wcol * 64 + j * 32 + 8 * q + 4 * (lane >> 5)) = v;
          }
    }
  };
  // tail split: park this K-part's raw fp32 tile in the workspace; the last part to arrive sums all parts in
  // part order (bitwise reproducible whoever is last) and runs the fused epilogue on the sum
  bool tail_src = false;
  constexpr int COHERENT = 0x11;   // sc0 | sc1
  const __amdgpu_buffer_rsrc_t tail_rsrc = make_rsrc(p.tail_ws, (uint32_t)TAIL_WS_BYTES_DEV);
  if (parts > 1) {
    // partials cross XCDs (private L2s): write them through and read them back with sc0|sc1 instead of
    // fencing (an agent-scope fence writes back / invalidates the whole L2 of the XCD)
    const uint32_t wbase = (uint32_t)(tail_tile * parts + part) << 16;
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
      stage(half);
      __syncthreads();
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int c = tid + 256 * it;
        const int row = c >> 4, col = (c & 15) * 8;
        const uint32_t d = wbase + (uint32_t)((half * 64 + row) * 128 + col) * 4;
        __builtin_amdgcn_raw_buffer_store_b128(*(const i32x4*)(cs + row * CP + col), tail_rsrc, d, 0, COHERENT);
        __builtin_amdgcn_raw_buffer_store_b128(*(const i32x4*)(cs + row * CP + col + 4), tail_rsrc, d + 16, 0, COHERENT);
      }
      __syncthreads();
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): my partial stores are acknowledged
    __syncthreads();
    if (tid == 0) ((int*)smem)[0] = atomicInc(p.tail_cnt + tail_tile, (unsigned)(parts - 1)) == (unsigned)(parts - 1);
    __syncthreads();
    if (!((int*)smem)[0]) return;
    tail_src = true;
  }
#pragma unroll 1
  for (int half = 0; half < 2; ++half) {
    if (!tail_src) {
      stage(half);
      __syncthreads();
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int c = tid + 256 * it;
      const int row = c >> 4, col = (c & 15) * 8;
      const int m = m0 + half * 64 + row, n = n0 + col;
      if (m < p.M && n < p.N) {
        f32x8 v;
        {
          f32x4 lo, hi;
          if (tail_src) {
            const uint32_t sp = ((uint32_t)(tail_tile * parts) << 16) + (uint32_t)((half * 64 + row) * 128 + col) * 4;
            union { i32x4 i; f32x4 f; } u0, u1;
            u0.i = __builtin_amdgcn_raw_buffer_load_b128(tail_rsrc, sp, 0, COHERENT);
            u1.i = __builtin_amdgcn_raw_buffer_load_b128(tail_rsrc, sp + 16, 0, COHERENT);
            lo = u0.f;
            hi = u1.f;
            for (int z = 1; z < parts; ++z) {
              u0.i = __builtin_amdgcn_raw_buffer_load_b128(tail_rsrc, sp + ((uint32_t)z << 16), 0, COHERENT);
              u1.i = __builtin_amdgcn_raw_buffer_load_b128(tail_rsrc, sp + ((uint32_t)z << 16) + 16, 0, COHERENT);
              lo += u0.f;
              hi += u1.f;
            }
          } else {
            lo = *(const f32x4*)(cs + row * CP + col);
            hi = *(const f32x4*)(cs + row * CP + col + 4);
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[e] = lo[e] * alpha;
            v[4 + e] = hi[e] * alpha;
          }
        }
        const long long crow = map_row(p.cmap, m);
        if (p.out_f32) {
          float* cp = (float*)opC + (long long)split * p.M * p.N + crow * p.ldc + n;
          if (p.accumulate) {
            const f32x4 o0 = *(const f32x4*)cp, o1 = *(const f32x4*)(cp + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              v[e] += o0[e];
              v[4 + e] += o1[e];
            }
          }
          *(f32x4*)cp = f32x4{v[0], v[1], v[2], v[3]};
          *(f32x4*)(cp + 4) = f32x4{v[4], v[5], v[6], v[7]};
          continue;
        }
        if (p.bias) v += cvt8(*(const bf16x8*)(p.bias + n));
        if (n < p.colscale_cols) v = cvt8(cvt8(v)) * p.colscale;      // (mpv.h colscale; reached by a tail-split tile: the plain path above handles the rest)
        // the product rounds to bf16 before anything else is applied to it (as the unfused reference ops do, and as the
        // 256x256 kernel's staged tile does): results do not depend on which tile kernel took the problem
        if (p.act_bwd || p.drop_thr || p.residual || p.accumulate || p.tap_out) v = cvt8(cvt8(v));
        if (p.tap_out && m % p.tap_group == 0) *(bf16x8*)(p.tap_out + (long long)(m / p.tap_group) * p.N + n) = cvt8(v);
        if (p.act) {
          const bf16x8 zb = cvt8(v);
          const f32x8 z = cvt8(zb);
          if (p.preact) {
            if (p.preact_deriv) {
              f32x8 d;
              if (p.act == 1) {
#pragma unroll
                for (int e = 0; e < 8; ++e) d[e] = gelu_erf_grad_f(z[e]);
              } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) {     // the tanh flavour from the 256x256 kernel's formula (one polynomial for GELU and GELU'):
                  float gv, dv;                   // a product split between the two tile kernels parks ONE function
                  mpv_gelu_tanh_both_t(z[e], gv, dv);
                  d[e] = dv;
                }
              }
              *(bf16x8*)(p.preact + crow * p.ldc + n) = cvt8(d);
            } else {
              *(bf16x8*)(p.preact + crow * p.ldc + n) = zb;
            }
          }
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = p.act == 1 ? gelu_erf_f(z[e]) : p.act == 2 ? gelu_tanh_f(z[e]) : fmaxf(z[e], 0.f);
        }
        if (p.act_bwd) {
          const f32x8 z = cvt8(*(const bf16x8*)(p.actz + (long long)m * p.ldz + n));
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] *= p.act_bwd == 1 ? gelu_erf_grad_f(z[e]) : p.act_bwd == 2 ? gelu_tanh_grad_f(z[e]) : p.act_bwd == MPV_ACT_DERIV ? z[e] : (z[e] > 0.f ? 1.f : 0.f);
        }
        if (p.drop_thr) {
          const uint64_t base = p.drop_offset + (uint64_t)m * (uint64_t)p.N + (uint64_t)n;
          v = mpv_dropout_vec<f32x8, 8>(v, seed_r, base, p.drop_thr, p.drop_scale);
        }
        if (p.residual) v += cvt8(*(const bf16x8*)(p.residual + crow * p.ldr + n));
        bf16* cp = (bf16*)opC + crow * p.ldc + n;
        if (p.accumulate) v += cvt8(*(const bf16x8*)cp);
        *(bf16x8*)cp = cvt8(v);
      }
    }
    if (!tail_src) __syncthreads();
  }
}

// sum split-K fp32 partials -> bf16 (optionally accumulating into the existing bf16 value); the trailing workgroups of
// the same launch finish the fused bias gradient (column sums of dY over the K splits)
// Epilogue of a split forward product (round 4: the decoder's 4h -> h product at 2.7B dims is 100 tiles on 256 CUs; its plain form was
// split along K, its bias + dropout form could not be -- 162 against 117 us): bias (bf16 [N], N = row length of the [M, N] output) and
// dropout on bf16(sum + bias), element index drop_offset + i, exactly as the unsplit epilogue orders them (gemm256.hip EP_DROP).
struct ReduceEpi {
  const bf16* bias;
  int N;
  uint32_t drop_thr;
  float drop_scale;
  uint64_t seed, drop_offset;
};
__global__ void splitk_reduce_kernel(const float* part, bf16* out, long long MN, int splits, int accumulate, int mn_blocks,
                                     const float* colsum_part, bf16* colsum_out, int M, const ReduceEpi epi) {
  if ((int)blockIdx.x >= mn_blocks) {
    const int m = ((int)blockIdx.x - mn_blocks) * blockDim.x + threadIdx.x;
    if (m >= M) return;
    float a = 0.f;
    for (int z = 0; z < splits; ++z) a += colsum_part[(long long)z * M + m];
    colsum_out[m] = f2bf(a);
    return;
  }
  const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i >= MN) return;
  // four partials requested per round trip (a rolled loop waits for every load before the next address is even formed: 9 - 28
  // serial HBM latencies per thread); the summation order z = 0, 1, 2, ... is unchanged
  f32x4 s = *(const f32x4*)(part + i);
  int z = 1;
  for (; z + 4 <= splits; z += 4) {
    const f32x4 a = *(const f32x4*)(part + (long long)z * MN + i), b = *(const f32x4*)(part + (long long)(z + 1) * MN + i);
    const f32x4 c = *(const f32x4*)(part + (long long)(z + 2) * MN + i), d = *(const f32x4*)(part + (long long)(z + 3) * MN + i);
    s += a;
    s += b;
    s += c;
    s += d;
  }
  for (; z < splits; ++z) s += *(const f32x4*)(part + (long long)z * MN + i);
  if (epi.bias) s += cvt4(*(const bf16x4*)(epi.bias + (int)(i % epi.N)));       // N % 4 == 0: the four elements share a row
  if (epi.drop_thr) {
    s = cvt4(cvt4(s));                                                            // the unsplit epilogue drops bf16(acc + bias)
    s = mpv_dropout_vec<f32x4, 4>(s, mpv_resolve_seed(epi.seed), epi.drop_offset + (uint64_t)i, epi.drop_thr, epi.drop_scale);
  }
  if (accumulate) s += cvt4(*(const bf16x4*)(out + i));
  *(bf16x4*)(out + i) = cvt4(s);
}

__global__ void colsum_finish_kernel(const float* part, bf16* out, int M, int splits) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  float a = 0.f;
  for (int z = 0; z < splits; ++z) a += part[(long long)z * M + m];
  out[m] = f2bf(a);
}


// ---------------------------------------------------------------------------------------------------------------
// Small-M variant (decode steps of KV-cache generation: M = beams <= 16, nn.Linear weights [N][K]).  HBM-bound: the
// weight is streamed exactly once, straight from global memory into MFMA operand registers -- no LDS staging, no
// 128-row tile.  A workgroup owns 16 output columns; its 4 waves split K; each lane loads 32 contiguous bytes of a
// weight row per 64-wide K block (4 lanes cover the row's full 128-byte line) and the matching 32 bytes of an
// activation row (L2-resident), and feeds them to two v_mfma_f32_16x16x32_bf16 (the k order inside an MFMA is free
// as long as both operands agree).  Partial tiles meet in LDS; the epilogue handles bias / activation / residual /
// C row map (the qkv row lands directly in the KV cache).
struct SmallMArgs {
  const bf16* A;
  const bf16* W;
  bf16* C;
  int M, N, K;
  long long lda, ldw, ldc, ldr;
  RowMap cmap;
  const bf16* bias;
  const bf16* residual;
  int act;
};

// NW = waves per workgroup that split K (round 4).  With 4 waves a lane of the K = 8192 product (4h -> h of the decoder MLP: N = 2048,
// so only 128 workgroups) walked 32 K-blocks = four dependent batches of 16 loads; with 16 waves every lane has its whole share in
// flight at once (one batch) and the 128 workgroups put 8 waves on a CU instead of 2.
// COLS = output columns per workgroup: 16, or 8 where N / 16 workgroups would leave CUs idle (N = 2048: 128 workgroups on 256 CUs; the
// weight rows 8..15 of the MFMA operand are then masked lanes -- no traffic, the matrix pipe is nowhere near a limit here).
// Non-temporal weight loads (`global_load_dwordx4 ... nt`) were measured and are not used: 1.396 -> 1.422 ms per decode step
// (profiles/r04_c19_decode_nt_ab.log) -- the five beams' rows of a step re-read nothing, but the NEXT step re-reads what L2 / MALL kept.
template <bool FULL, int NW = 4, int COLS = 16>   // FULL: N % 16 == 0 and K % 64 == 0 -> branch-free main loop (unrolled 8x: 16 weight loads in flight per lane)
__global__ __launch_bounds__(64 * NW) void gemm_small_m_kernel(const SmallMArgs p) {
  __shared__ float part[NW][16][17];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n0 = blockIdx.x * COLS;
  const int r = lane & 15, g = lane >> 4;                       // operand row (n for W, m for A), 16-wide k group
  const int kblocks = (p.K + 63) >> 6;
  const int per = (kblocks + NW - 1) / NW;
  const int kb0 = wave * per, kb1 = min(kblocks, kb0 + per);
  const bool wok = r < COLS && (n0 + r) < p.N, aok = r < p.M;
  const bf16* wrow = p.W + (long long)(wok ? n0 + r : 0) * p.ldw + g * 16;
  const bf16* arow = p.A + (long long)(aok ? r : 0) * p.lda + g * 16;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const i32x4 zero = {0, 0, 0, 0};
  union Frag { i32x4 i; bf16x8 b; };
  if constexpr (FULL) {
    int kb = kb0;
    constexpr int UN = NW > 4 ? 4 : 8;      // (16 waves: 128 VGPRs per wave -- four K-blocks = 8 weight + 8 activation chunks per batch)
    for (; kb + UN <= kb1; kb += UN) {
      Frag w0[UN], w1[UN], a0[UN], a1[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        w0[u].i = (COLS == 16 || wok) ? *(const i32x4*)(wrow + (kb + u) * 64) : zero;
        w1[u].i = (COLS == 16 || wok) ? *(const i32x4*)(wrow + (kb + u) * 64 + 8) : zero;
      }
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        a0[u].i = aok ? *(const i32x4*)(arow + (kb + u) * 64) : zero;
        a1[u].i = aok ? *(const i32x4*)(arow + (kb + u) * 64 + 8) : zero;
      }
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0[u].b, w0[u].b, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[u].b, w1[u].b, acc, 0, 0, 0);
      }
    }
    for (; kb < kb1; ++kb) {
      Frag w0, w1, a0, a1;
      w0.i = (COLS == 16 || wok) ? *(const i32x4*)(wrow + kb * 64) : zero;
      w1.i = (COLS == 16 || wok) ? *(const i32x4*)(wrow + kb * 64 + 8) : zero;
      a0.i = aok ? *(const i32x4*)(arow + kb * 64) : zero;
      a1.i = aok ? *(const i32x4*)(arow + kb * 64 + 8) : zero;
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0.b, w0.b, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1.b, w1.b, acc, 0, 0, 0);
    }
  } else {
    for (int kb = kb0; kb < kb1; ++kb) {
      const int k = kb * 64 + g * 16;
      const bool k0 = k < p.K, k1 = k + 8 < p.K;                // K is a multiple of 8: 16-byte chunks are all-in or all-out
      Frag w0, w1, a0, a1;
      w0.i = (wok && k0) ? *(const i32x4*)(wrow + kb * 64) : zero;
      w1.i = (wok && k1) ? *(const i32x4*)(wrow + kb * 64 + 8) : zero;
      a0.i = (aok && k0) ? *(const i32x4*)(arow + kb * 64) : zero;
      a1.i = (aok && k1) ? *(const i32x4*)(arow + kb * 64 + 8) : zero;
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0.b, w0.b, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1.b, w1.b, acc, 0, 0, 0);
    }
  }
  // D layout: lane -> column n = lane & 15, rows m = 4 * (lane >> 4) + i
#pragma unroll
  for (int i = 0; i < 4; ++i) part[wave][4 * g + i][r] = acc[i];
  __syncthreads();
  const int m = tid >> 4, n = n0 + (tid & 15);
  if (tid < 256 && (tid & 15) < COLS && m < p.M && n < p.N) {
    float v = (part[0][m][tid & 15] + part[1][m][tid & 15]) + (part[2][m][tid & 15] + part[3][m][tid & 15]);
#pragma unroll
    for (int w = 4; w < NW; w += 4)
      v += (part[w][m][tid & 15] + part[w + 1][m][tid & 15]) + (part[w + 2][m][tid & 15] + part[w + 3][m][tid & 15]);
    if (p.bias) v += bf2f(p.bias[n]);
    if (p.act) {
      const float z = bf2f(f2bf(v));
      v = p.act == 1 ? gelu_erf_f(z) : p.act == 2 ? gelu_tanh_f(z) : fmaxf(z, 0.f);
    }
    const long long crow = map_row(p.cmap, m);
    if (p.residual) v += bf2f(p.residual[crow * p.ldr + n]);
    p.C[crow * p.ldc + n] = f2bf(v);
  }
}

}  // namespace

constexpr int SLOTS = 512;                                   // resident workgroups: 256 CUs x 2
constexpr size_t TAIL_WS_BYTES = (size_t)SLOTS * BM * BN * sizeof(float);   // 32 MiB of partial tiles
constexpr size_t TAIL_CNT_BYTES = SLOTS * sizeof(unsigned);                  // + their arrival counters, in the caller's workspace

// Forward / dgrad products with at most half a round of 256-tiles are split along K when the reduction is at least this long
// (MPV_FWD_SPLIT_MINK: measurement knob; 4096 = rounds 4-5: the LM head's dgrad on the loss window, the 2.7B decoder's 4h -> h product)
static int fwd_split_min_k() {
  static const int v = [] {
    const char* e = getenv("MPV_FWD_SPLIT_MINK");
    return e && atoi(e) >= 512 ? atoi(e) : 4096;
  }();
  return v;
}

extern "C" size_t mpv_gemm_workspace_size(int64_t M, int64_t N, int64_t K, int transA, int transB) {
  if (!(transA && transB)) {
    // forward / dgrad products with few output tiles and a long reduction (the LM head's dgrad on the loss window:
    // 1024 x 2048 x 51200) are split along K like a wgrad: fp32 partials of up to 16 splits
    const int64_t t256 = ((M + 255) / 256) * ((N + 255) / 256);
    const size_t part = (t256 * 2 <= 256 && K >= fwd_split_min_k()) ? (size_t)M * N * sizeof(float) * 16 : 0;
    return (part > TAIL_WS_BYTES ? part : TAIL_WS_BYTES) + TAIL_CNT_BYTES;
  }
  // wgrad: split the (long) reduction so that >= ~2 workgroups per CU exist (+ the per-split column sums of the fused
  // bias gradient)
  const size_t w = (size_t)M * N * sizeof(float) * 32 + (size_t)M * sizeof(float) * 64;
  return w > TAIL_WS_BYTES ? w : TAIL_WS_BYTES;
}

// wgrad split: pick the split count whose workgroup count best fills whole rounds of the 512 resident
// slots (256 CUs x 2 workgroups), keeping at least 4 K-steps per split.
static int choose_splitk(int64_t M, int64_t N, int64_t K, size_t ws_bytes) {
  const int64_t tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
  int best = 1;
  double best_eff = 0.0;
  for (int s = 1; s <= 32; ++s) {
    if (s > 1 && (K / s < 4 * BK || (size_t)M * N * sizeof(float) * s + (size_t)s * M * sizeof(float) > ws_bytes)) break;
    const int64_t blocks = tiles * s;
    const double eff = (double)blocks / (double)(((blocks + 511) / 512) * 512);
    if (eff > best_eff + 0.02) {
      best_eff = eff;
      best = s;
    }
  }
  return best;
}

// MPV_GEMM_KERNEL=128|256 pins the tile kernel (measurement aid); anything else = choose per problem
static int gemm_variant() {
  static const int v = [] {
    const char* e = getenv("MPV_GEMM_KERNEL");
    return e ? atoi(e) : 0;
  }();
  return v;
}

// wgrad split for the 256x256 kernel.  Filling whole rounds of the CUs is not enough of a criterion: every split writes
// an fp32 partial of the whole output and the reduce pass reads it back, so 27 tiles x 28 splits (98 % of three rounds)
// lost to 27 x 9 (95 % of one round) by 25 % (tools/probe/wgrad_splits.py: 217.7 vs 177.6 us at M = 2304, N = 768,
// K = 50432).  Modelled time (us), constants fitted to that sweep on MI355X: rounds x (12 for prologue + fp32 epilogue
// + 1.5 per K-tile of a split) + partial bytes read back at 3.5 TB/s + 6 for the reduce launch.
static int choose_splitk256(int64_t tiles, int64_t K, int64_t M, int64_t N, size_t ws_bytes) {
  int best = 1;
  double best_us = 0.0;
  const int64_t nk = K / 64;
  for (int s = 1; s <= 64; ++s) {
    if (s > 1 && (K / s < 8 * 64 || (size_t)M * N * sizeof(float) * s + (size_t)s * M * sizeof(float) > ws_bytes)) break;
    const int64_t blocks = tiles * s;
    const double rounds = (double)((blocks + 255) / 256);
    const double kt = (double)((nk + s - 1) / s);
    const double us = rounds * (12.0 + 1.5 * kt) + (s > 1 ? (double)s * (double)M * (double)N * 4.0 / 3.5e6 + 6.0 : 0.0);
    if (s == 1 || us < best_us) {
      best_us = us;
      best = s;
    }
  }
  return best;
}

extern "C" int mpv_gemm_bf16(const void* A, const void* B, void* C, int64_t M, int64_t N, int64_t K, int64_t lda,
                             int64_t ldb, int64_t ldc, int transA, int transB, const mpv_gemm_epilogue* ep,
                             void* workspace, size_t workspace_bytes, hipStream_t stream) {
  MPV_REQUIRE(A && B && C, MPV_E_ARG, "mpv_gemm_bf16: null operand");
  MPV_REQUIRE(M > 0 && N > 0 && K > 0, MPV_E_SHAPE, "mpv_gemm_bf16: empty problem %lld x %lld x %lld", (long long)M,
              (long long)N, (long long)K);
  MPV_REQUIRE(!(transA && !transB), MPV_E_ARG, "mpv_gemm_bf16: transA=1,transB=0 is not a Linear pass");
  MPV_REQUIRE(N % 8 == 0 && ldc % 8 == 0, MPV_E_ALIGN, "mpv_gemm_bf16: N (%lld) and ldc must be multiples of 8", (long long)N);
  MPV_REQUIRE(lda % 8 == 0 && ldb % 8 == 0, MPV_E_ALIGN, "mpv_gemm_bf16: lda/ldb must be multiples of 8 elements");
  if (!transA) MPV_REQUIRE(K % 8 == 0, MPV_E_ALIGN, "mpv_gemm_bf16: K (%lld) must be a multiple of 8", (long long)K);
  if (transA) MPV_REQUIRE(M % 8 == 0, MPV_E_ALIGN, "mpv_gemm_bf16: M (%lld) must be a multiple of 8 when transA", (long long)M);
  MPV_REQUIRE((((uintptr_t)A | (uintptr_t)B | (uintptr_t)C) & 15) == 0, MPV_E_ALIGN, "mpv_gemm_bf16: operands must be 16-byte aligned");
  // a leading dimension shorter than the row it strides over would make rows overlap (the kernels' buffer descriptors are sized from
  // rows x ld): refused, not clipped
  MPV_REQUIRE(lda >= (transA ? M : K) && ldb >= (transB ? N : K) && ldc >= N, MPV_E_SHAPE,
              "mpv_gemm_bf16: leading dimension shorter than its row (lda=%lld ldb=%lld ldc=%lld for M=%lld N=%lld K=%lld, transA=%d transB=%d)",
              (long long)lda, (long long)ldb, (long long)ldc, (long long)M, (long long)N, (long long)K, transA, transB);

  GemmArgs g = {};
  g.A = (const bf16*)A;
  g.B = (const bf16*)B;
  g.C = C;
  g.M = (int)M;
  g.N = (int)N;
  g.K = (int)K;
  g.lda = lda;
  g.ldb = ldb;
  g.ldc = ldc;
  g.alpha = 1.0f;
  RowMap id = {0, 0, 0};
  g.amap = g.cmap = g.kmap = id;
  if (ep) {
    g.amap = RowMap{ep->a_group, ep->a_stride, ep->a_offset};
    g.cmap = RowMap{ep->c_group, ep->c_stride, ep->c_offset};
    g.kmap = RowMap{ep->k_group, ep->k_stride, ep->k_offset};
    g.bias = (const bf16*)ep->bias;
    g.act = ep->act;
    g.preact = (bf16*)ep->preact_out;
    g.residual = (const bf16*)ep->residual;
    g.ldr = ep->ldr ? ep->ldr : ldc;
    g.actz = (const bf16*)ep->act_bwd_z;
    g.ldz = ep->ldz ? ep->ldz : N;
    g.act_bwd = ep->act_bwd_z ? ep->act_bwd : 0;
    MPV_REQUIRE(!ep->preact_deriv || (ep->preact_out && (ep->act == MPV_ACT_GELU_ERF || ep->act == MPV_ACT_GELU_TANH)), MPV_E_ARG,
                "mpv_gemm_bf16: preact_deriv needs preact_out and a GELU activation");
    g.preact_deriv = ep->preact_deriv ? 1 : 0;
    g.keep_c = ep->keep_output ? 1 : 0;
    if (ep->colscale_cols > 0) {
      MPV_REQUIRE(ep->colscale_cols % 8 == 0 && !ep->act && !ep->preact_out && !ep->residual && !ep->act_bwd_z && !(ep->dropout_p > 0.f) &&
                      !ep->out_f32 && !ep->accumulate && !ep->row_tap_out && !ep->alpha_dev && !(transA && transB),
                  MPV_E_ARG, "mpv_gemm_bf16: colscale takes a bias-only bf16 forward / dgrad product and a multiple of 8 columns");
      g.colscale = ep->colscale;
      g.colscale_cols = ep->colscale_cols;
    }
    if (ep->dropout_p > 0.f) {
      MPV_REQUIRE(ep->dropout_p < 1.f, MPV_E_ARG, "mpv_gemm_bf16: dropout_p must be < 1");
      g.drop_thr = mpv_drop_threshold(ep->dropout_p);
      g.drop_scale = 1.0f / (1.0f - ep->dropout_p);
      g.seed = ep->seed;
      g.drop_offset = ep->offset;
    }
    g.alpha_dev = ep->alpha_dev;
    if (ep->alpha != 0.f) g.alpha = ep->alpha;
    g.out_f32 = ep->out_f32;
    g.accumulate = ep->accumulate;
    if (ep->row_tap_out) {
      MPV_REQUIRE(ep->row_tap_group > 0 && !ep->out_f32 && !(transA && transB), MPV_E_ARG,
                  "mpv_gemm_bf16: row_tap_out needs row_tap_group > 0 and a bf16 forward/dgrad product");
      g.tap_out = (bf16*)ep->row_tap_out;
      g.tap_group = ep->row_tap_group;
    }
  }
  void* colsum_out = ep ? ep->colsum_out : nullptr;
  MPV_REQUIRE(!colsum_out || (transA && transB && !g.out_f32), MPV_E_ARG, "mpv_gemm_bf16: colsum_out is a wgrad (transA=transB=1) option");
  // decode regime: a handful of rows against a k-contiguous weight -> the weight-streaming kernel
  if (!transA && !transB && M <= 16 && !g.out_f32 && !g.accumulate && !g.drop_thr && !g.act_bwd && !g.preact && !g.alpha_dev && !g.tap_out && !g.colscale_cols &&
      g.alpha == 1.0f && g.amap.group == 0 && !colsum_out) {
    SmallMArgs sm = {};
    sm.A = g.A; sm.W = g.B; sm.C = (bf16*)C;
    sm.M = (int)M; sm.N = (int)N; sm.K = (int)K;
    sm.lda = lda; sm.ldw = ldb; sm.ldc = ldc; sm.ldr = g.ldr;
    sm.cmap = g.cmap; sm.bias = g.bias; sm.residual = g.residual; sm.act = g.act;
    if (N % 16 == 0 && K % 64 == 0 && K >= 4096 && N / 16 <= 256)        // long reduction, few workgroups: 16 waves split K, 8 columns each
      hipLaunchKernelGGL((gemm_small_m_kernel<true, 16, 8>), dim3((unsigned)(N / 8)), dim3(1024), 0, stream, sm);
    else if (N % 16 == 0 && K % 64 == 0 && N / 16 <= 384)                  // workgroup counts that leave CUs idle or 2 : 1 unbalanced (N = 2048: 128, N = 6144: 384 on 256 CUs): 8 columns each
      hipLaunchKernelGGL((gemm_small_m_kernel<true, 4, 8>), dim3((unsigned)(N / 8)), dim3(256), 0, stream, sm);
    else if (N % 16 == 0 && K % 64 == 0)
      hipLaunchKernelGGL(gemm_small_m_kernel<true>, dim3((unsigned)(N / 16)), dim3(256), 0, stream, sm);
    else
      hipLaunchKernelGGL(gemm_small_m_kernel<false>, dim3((unsigned)((N + 15) / 16)), dim3(256), 0, stream, sm);
    return mpv_check_launch("mpv_gemm_bf16");
  }
  // byte extents for the buffer descriptors (rows may be gathered through amap/kmap)
  const long long a_rows = transA ? map_row(g.kmap, K - 1) + 1 : map_row(g.amap, M - 1) + 1;
  const long long b_rows = transB ? map_row(g.kmap, K - 1) + 1 : N;
  const long long a_bytes = ((a_rows - 1) * lda + (transA ? M : K)) * 2;
  const long long b_bytes = ((b_rows - 1) * ldb + (transB ? N : K)) * 2;
  MPV_REQUIRE(a_bytes < 0x7FFFFFF0ll && b_bytes < 0x7FFFFFF0ll, MPV_E_SHAPE,
              "mpv_gemm_bf16: operand larger than 2 GiB (%lld / %lld bytes)", a_bytes, b_bytes);
  g.a_bytes = (uint32_t)a_bytes;
  g.b_bytes = (uint32_t)b_bytes;

  const int tiles_m = (int)((M + BM - 1) / BM), tiles_n = (int)((N + BN - 1) / BN);
  g.tiles_n = tiles_n;
  g.tiles_m = tiles_m;
  g.nwg = tiles_m * tiles_n;
  int splitk = 1;
  if (transA && transB && !g.out_f32) splitk = choose_splitk(M, N, K, workspace ? workspace_bytes : 0);
  int kps = (int)K;
  if (splitk > 1) {
    kps = (int)(((K + splitk - 1) / splitk + BK - 1) / BK * BK);
    splitk = (int)((K + kps - 1) / kps);
  }
  g.k_per_split = kps;

  // ---- 256x256 eight-phase kernel (gemm256.hip) for the problems that fill the chip with 256-tiles
  const int variant = (ep && ep->tile_hint) ? ep->tile_hint : gemm_variant();
  if (variant != 128 && (K % 64 == 0 || (transA && transB && K >= 64 && g.kmap.group == 0)) && (M >= 256 || variant == 160 || variant == 192) && N >= 256) {
    const int64_t t256 = ((M + 255) / 256) * ((N + 255) / 256);
    int s256 = 1;
    if (transA && transB && !g.out_f32) {
      s256 = choose_splitk256(t256, K, M, N, workspace ? workspace_bytes : 0);
      if (ep && ep->split_hint > 0 && K / ep->split_hint >= 64 &&
          (size_t)M * N * sizeof(float) * ep->split_hint + (size_t)ep->split_hint * M * sizeof(float) <= (workspace ? workspace_bytes : 0))
        s256 = ep->split_hint;
    } else if (!g.out_f32 && t256 * 2 <= 256 && K >= fwd_split_min_k() && !g.act && !g.residual && !g.act_bwd && !g.tap_out && !g.colscale_cols &&
             !g.preact && g.cmap.group == 0 && ldc == N && N % 4 == 0 && ((uintptr_t)g.bias & 7) == 0) {      // (the reduce reads the bias as bf16x4)
      // few tiles, long reduction, plain / bias / bias + dropout epilogue (applied by the reduce): split-K over the idle CUs (at most
      // the 16 splits the workspace size allows for)
      s256 = choose_splitk256(t256, K, M, N, workspace ? workspace_bytes : 0);
      if (s256 > 16) s256 = 16;
    }
    {   // measured on MI355X (tools/gemm_ab.py): the 256x256 kernel wins on every eligible shape of the path, also when
        // its tiles fill only 5/8 of the CUs (M = 5120, N = 2048: 959 vs 764 TFLOP/s)
      GemmArgs h = g;
      h.tile_rows = (variant == 160 || variant == 192 || variant == 256) ? variant : 0;
      h.gm = (ep && ep->gm_hint > 0) ? ep->gm_hint : 0;
      h.splits = s256;
      h.k_per_split = (int)((K + 63) / 64 * 64);      // (= K for whole K-tiles; a weight-gradient product may end in a partial one)
      if (s256 > 1) {
        h.k_per_split = (int)((((K + 63) / 64 + s256 - 1) / s256) * 64);
        h.splits = (int)((K + h.k_per_split - 1) / h.k_per_split);
      }
      void* user_c256 = C;
      const int user_acc256 = g.accumulate;
      bool ok = true;
      if (colsum_out) {
        const size_t need = (size_t)(h.splits > 1 ? h.splits : 0) * M * N * sizeof(float) + (size_t)h.splits * M * sizeof(float);
        ok = workspace && workspace_bytes >= need;
        h.colsum_part = (float*)((char*)workspace + (size_t)(h.splits > 1 ? h.splits : 0) * M * N * sizeof(float));
      }
      ReduceEpi repi = {};
      if (h.splits > 1) {
        const bool wgrad = transA && transB;
        ok = ok && (wgrad ? (!g.bias && !g.drop_thr) : (N % 4 == 0 && ((uintptr_t)g.bias & 7) == 0)) && !g.act && !g.residual && !g.act_bwd &&
             g.cmap.group == 0 && ldc == N;
        if (!wgrad) {       // forward / dgrad product split along K: bias and dropout move into the reduce
          repi = ReduceEpi{g.bias, (int)N, g.drop_thr, g.drop_scale, g.seed, g.drop_offset};
          h.bias = nullptr;
          h.drop_thr = 0;
        }
        h.C = workspace;
        h.out_f32 = 1;
        h.accumulate = 0;
      }
      if (ok && mpv_gemm256_try_launch(h, transA, transB, stream)) {
        if (h.splits > 1) {
          const long long MN = (long long)M * N;
          const int thr = 256;
          const long long rblocks = (MN / 4 + thr - 1) / thr;
          const int cblocks = colsum_out ? (int)((M + thr - 1) / thr) : 0;
          hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)(rblocks + cblocks)), dim3(thr), 0, stream, (const float*)workspace,
                             (bf16*)user_c256, MN, h.splits, user_acc256, (int)rblocks, (const float*)h.colsum_part, (bf16*)colsum_out, (int)M, repi);
        } else if (colsum_out) {
          hipLaunchKernelGGL(colsum_finish_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, stream, (const float*)h.colsum_part,
                             (bf16*)colsum_out, (int)M, 1);
        }
        return mpv_check_launch("mpv_gemm_bf16");
      }
    }
  }

  // tail split of a mostly empty last round (see GemmArgs); needs the partial-tile workspace
  int grid_x = g.nwg;
  if (splitk == 1 && !g.out_f32 && !colsum_out && workspace && workspace_bytes >= TAIL_WS_BYTES + TAIL_CNT_BYTES) {
    const int R = g.nwg % SLOTS, nk = (int)((K + BK - 1) / BK);
    int tg = R > 0 ? SLOTS / R : 1;
    if (tg > 8) tg = 8;
    while (tg > 1 && nk / tg < 16) --tg;   // short parts do not pay for the partial-tile round trip (measured)
    if (tg > 1) {
      const int steps = (nk + tg - 1) / tg;
      tg = (nk + steps - 1) / steps;
      // arrival counters live behind the partial tiles in the caller's (stream-ordered) workspace and are zeroed by a
      // memset node ahead of the launch: no library-owned device state, nothing allocated here
      unsigned* cnt = tg > 1 ? (unsigned*)((char*)workspace + TAIL_WS_BYTES) : nullptr;
      if (cnt && hipMemsetAsync(cnt, 0, TAIL_CNT_BYTES, stream) != hipSuccess) cnt = nullptr;
      if (cnt) {
        g.tail_start = g.nwg - R;
        g.tail_g = tg;
        g.tail_steps = steps;
        g.tail_ws = (float*)workspace;
        g.tail_cnt = cnt;
        grid_x = g.tail_start + R * tg;
      }
    }
  }
  g.splits = splitk;
  dim3 grid(splitk > 1 ? g.nwg * splitk : grid_x), block(256);
  void* user_c = C;
  const int user_acc = g.accumulate;
  if (colsum_out) {
    const size_t need = (size_t)(splitk > 1 ? splitk : 0) * M * N * sizeof(float) + (size_t)splitk * M * sizeof(float);
    MPV_REQUIRE(workspace && workspace_bytes >= need, MPV_E_ARG, "mpv_gemm_bf16: workspace too small for colsum_out");
    g.colsum_part = (float*)((char*)workspace + (size_t)(splitk > 1 ? splitk : 0) * M * N * sizeof(float));
  }
  if (splitk > 1) {
    MPV_REQUIRE(!g.bias && !g.act && !g.residual && !g.act_bwd && !g.drop_thr && g.cmap.group == 0 && ldc == N, MPV_E_ARG,
                "mpv_gemm_bf16: split-K (wgrad) pass takes no fused epilogue");
    g.C = workspace;
    g.out_f32 = 1;
    g.accumulate = 0;
  }
  if (!transA && !transB)
    hipLaunchKernelGGL((gemm_bf16_kernel<false, false>), grid, block, 0, stream, g, GemmBatchPtrs{});
  else if (!transA && transB)
    hipLaunchKernelGGL((gemm_bf16_kernel<false, true>), grid, block, 0, stream, g, GemmBatchPtrs{});
  else
    hipLaunchKernelGGL((gemm_bf16_kernel<true, true>), grid, block, 0, stream, g, GemmBatchPtrs{});
  if (splitk > 1) {
    const long long MN = (long long)M * N;
    const int thr = 256;
    const long long blocks = (MN / 4 + thr - 1) / thr;
    const int cblocks = colsum_out ? (int)((M + thr - 1) / thr) : 0;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)(blocks + cblocks)), dim3(thr), 0, stream, (const float*)workspace,
                       (bf16*)user_c, MN, splitk, user_acc, (int)blocks, (const float*)g.colsum_part, (bf16*)colsum_out, (int)M, ReduceEpi{});
  } else if (colsum_out) {
    hipLaunchKernelGGL(colsum_finish_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, stream, (const float*)g.colsum_part,
                       (bf16*)colsum_out, (int)M, splitk);
  }
  return mpv_check_launch("mpv_gemm_bf16");
}

// mpv_gemm_bf16_batched (include/mpv.h): `batch` products of one shape, plain epilogue (bf16 out), on the 128x128 kernel with
// blockIdx.y = problem; more than GEMM_BATCH_MAX problems go out as several launches.
extern "C" int mpv_gemm_bf16_batched(const void* const* A, const void* const* B, void* const* C, int batch, int64_t M, int64_t N, int64_t K,
                                     int64_t lda, int64_t ldb, int64_t ldc, int transA, int transB, hipStream_t stream) {
  MPV_REQUIRE(A && B && C && batch > 0, MPV_E_ARG, "mpv_gemm_bf16_batched: null pointer table or empty batch");
  MPV_REQUIRE(M > 0 && N > 0 && K > 0, MPV_E_SHAPE, "mpv_gemm_bf16_batched: empty problem %lld x %lld x %lld", (long long)M, (long long)N, (long long)K);
  MPV_REQUIRE(!(transA && !transB), MPV_E_ARG, "mpv_gemm_bf16_batched: transA=1,transB=0 is not a Linear pass");
  MPV_REQUIRE(N % 8 == 0 && ldc % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && (transA ? M % 8 == 0 : K % 8 == 0), MPV_E_ALIGN,
              "mpv_gemm_bf16_batched: N, K (M when transA) and the leading dimensions must be multiples of 8");
  MPV_REQUIRE(lda >= (transA ? M : K) && ldb >= (transB ? N : K) && ldc >= N, MPV_E_SHAPE, "mpv_gemm_bf16_batched: leading dimension shorter than its row");
  const long long a_bytes = (((transA ? K : M) - 1) * lda + (transA ? M : K)) * 2, b_bytes = (((transB ? K : N) - 1) * ldb + (transB ? N : K)) * 2;
  MPV_REQUIRE(a_bytes < 0x7FFFFFF0ll && b_bytes < 0x7FFFFFF0ll, MPV_E_SHAPE, "mpv_gemm_bf16_batched: operand larger than 2 GiB");
  for (int i = 0; i < batch; ++i) {
    MPV_REQUIRE(A[i] && B[i] && C[i], MPV_E_ARG, "mpv_gemm_bf16_batched: null operand in problem %d", i);
    MPV_REQUIRE((((uintptr_t)A[i] | (uintptr_t)B[i] | (uintptr_t)C[i]) & 15) == 0, MPV_E_ALIGN, "mpv_gemm_bf16_batched: operands must be 16-byte aligned (problem %d)", i);
  }
  GemmArgs g = {};
  g.M = (int)M; g.N = (int)N; g.K = (int)K;
  g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  g.alpha = 1.0f;
  g.amap = g.cmap = g.kmap = RowMap{0, 0, 0};
  g.a_bytes = (uint32_t)a_bytes;
  g.b_bytes = (uint32_t)b_bytes;
  g.tiles_m = (int)((M + BM - 1) / BM);
  g.tiles_n = (int)((N + BN - 1) / BN);
  g.nwg = g.tiles_m * g.tiles_n;
  g.splits = 1;
  g.k_per_split = (int)K;
  g.batch = 1;
  for (int b0 = 0; b0 < batch; b0 += GEMM_BATCH_MAX) {
    const int nb = batch - b0 < GEMM_BATCH_MAX ? batch - b0 : GEMM_BATCH_MAX;
    GemmBatchPtrs bp = {};
    for (int i = 0; i < nb; ++i) {
      bp.a[i] = (const bf16*)A[b0 + i];
      bp.b[i] = (const bf16*)B[b0 + i];
      bp.c[i] = (bf16*)C[b0 + i];
    }
    const dim3 grid((unsigned)g.nwg, (unsigned)nb), block(256);
    if (!transA && !transB)
      hipLaunchKernelGGL((gemm_bf16_kernel<false, false>), grid, block, 0, stream, g, bp);
    else if (!transA && transB)
      hipLaunchKernelGGL((gemm_bf16_kernel<false, true>), grid, block, 0, stream, g, bp);
    else
      hipLaunchKernelGGL((gemm_bf16_kernel<true, true>), grid, block, 0, stream, g, bp);
  }
  return mpv_check_launch("mpv_gemm_bf16_batched");
}
