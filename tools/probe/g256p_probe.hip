// g256p_probe.hip -- a GEMM tile whose epilogue does not own the matrix pipe (NOTEBOOK.md section 10, lead 1).  NOT a product path,
// NOT yet run on a GPU: written and compiled in round 4 (ISA checked on the host: tools/probe/README.md), to be validated and measured
// against csrc/gemm256.hip at the start of the next round (./g256p_probe M N K prints its error against a host reference and its time).
//
// What the production kernel cannot do: at K = 768 a 256 x 256 tile is 2.2 us of prologue + 15.7 us of main loop + >= 4.9 us of a
// store-issue-bound epilogue, and nothing overlaps the first and the last -- 160 KiB of LDS and all 256 registers of both waves of a
// SIMD belong to the one tile (three rounds of attempts: NOTEBOOK section 9).  This kernel changes the register economy instead:
//
//  * FOUR waves (one per SIMD, 512 registers each): a wave owns a 128 x 128 quarter of the tile, its 64 accumulator blocks
//    (v_mfma_f32_16x16x32_bf16) fill the 256 AGPRs, the 256 arch VGPRs hold the operand fragments (A double-buffered, B rolling: 96) and ...
//  * ... HALF of the previous tile, already converted to bf16 and arranged as 16-byte row pieces (64 registers); the other half is
//    parked in 64 KiB of LDS (a wave-private spill area: no barrier).  The workgroup is PERSISTENT: the LDS-DMA ring keeps running
//    across tile boundaries (the next tile's first K-steps arrive while the current tile finishes: no prologue), and the parked
//    tile leaves through two 16-byte stores per K-step under the next tile's MFMAs (store issue: 8 per 1024 clocks per CU against
//    the ~74-clock limit).  The conversion of the finished accumulators happens DURING the tile's last K-step: a column pair is final
//    once its MFMAs of that step are issued and is converted in the shadow of the later columns' MFMAs (only the last pair is exposed).
//  * K-steps of 32 through a ring of THREE slots of (A 256 x 32, B 256 x 32) = 96 KiB; one barrier per K-step (the production
//    kernel: eight per 64).  Image of a unit: [256 rows][64 B], 16-byte chunk c of row r at position c ^ f4((r >> 2) & 3) (applied to
//    the DMA's global source address and to the ds_read_b128 fragment address; tools/probe/sim_g256p_layout.py checks both).
//  * MFMA operands swapped as in the production kernel (a lane owns 4 consecutive columns of a row); two adjacent column blocks are
//    merged into 8 consecutive columns per lane by v_permlane16_swap (odd 16-lane rows of one register <-> even rows of the other).
//  * The bias is added at the conversion (as the C operand of a tile's first MFMAs it would have to sit in AGPRs, which are full).
// NT form only (A [M][K], B [N][K], both k-contiguous), M % 256 == N % 256 == 0, K % 64 == 0, K >= 640.
// Build: hipcc --offload-arch=gfx950 -O3 -o g256p_probe g256p_probe.hip ; run: timeout 30 ./g256p_probe M N K [0|1]  (both variants, or one)
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((address_space(3))) void lds_void;
template <int V>
struct IC { static constexpr int value = V; };

constexpr int TM = 256, TN = 256, SUBK = 32;
constexpr int UNIT = 16384;                       // one operand half of a K-step: 256 rows x 64 B
constexpr int NSLOT = 3;
constexpr int RING = NSLOT * 2 * UNIT;            // 96 KiB
constexpr int PARK = 4 * 16 * 1024;               // 64 KiB: 4 waves x 16 pieces x (64 lanes x 16 B)
constexpr int SMEM = RING + PARK;                 // 160 KiB
constexpr int HEAD_SUB = 17;                      // K-steps 0..16 of a tile are straight-line code (they carry the parked tile out)
constexpr uint32_t OOB = 0x80000000u;

#define SB() __builtin_amdgcn_sched_barrier(0)
// Round-5 attribution switches (-DABL_NODMA / -DABL_NOREAD / -DABL_NOBAR / -DABL_M0X4): the steady-state K loop without its LDS-DMA
// instructions, without its fragment reads, without its workgroup barrier (TIMING ONLY: the results are wrong by construction), and
// with ONE m0 write per operand half (the four pieces of a half addressed through the instruction offset, which moves the LDS
// address and the global address together; the scalar offset takes the difference back out) -- that one stays correct.
// -DABL_OOBDMA: every DMA lane out of range (the instructions are issued, zero fill, NO memory traffic); -DABL_NOWAIT: no counted wait.
// Result (profiles/r05_c2_*, r05_c3_*): 0.79 PF as is; no DMA 1.28-1.33; DMA issued but nothing fetched 1.21-1.23; no wait 0.80;
// no fragment reads 0.75; no barrier 0.80; a FOUR-slot ring (two batches behind the wait; patch not kept) 0.78.  It is neither
// the instructions nor the latency: it is the BYTES -- the walk below hands tile t to workgroup t % G, i.e. to XCD t % 8, so the
// three / nine tiles that share an A panel run on different XCDs and every L2 fetches every panel (~3 TB/s of L2 misses).
#if defined(ABL_NODMA) || defined(ABL_NOREAD) || defined(ABL_NOBAR) || defined(ABL_OOBDMA) || defined(ABL_NOWAIT) || defined(ABL_FULLLINE)
#define ABL_WRONG 1
#else
#define ABL_WRONG 0
#endif

__device__ __forceinline__ void dma16(i32x4 rsrc, uint32_t lds_addr, uint32_t voff, uint32_t soff) {
  // (readfirstlane: both are wave-uniform by construction; this only tells the register allocator)
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :
               : "s"(__builtin_amdgcn_readfirstlane(lds_addr)), "v"(voff), "s"(rsrc), "s"(__builtin_amdgcn_readfirstlane(soff))
               : "memory");
}
__device__ __forceinline__ i32x4 raw_rsrc(const void* ptr, uint32_t bytes) {
  const uint64_t a = (uint64_t)ptr;
  return i32x4{(int)(uint32_t)a, (int)((uint32_t)(a >> 32) & 0xffffu), (int)bytes, 0x00020000};
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
// chunk rotation key of a row group: 0, 2, 3, 1 -- conflict-free for the hardware lane groups of ds_read_b128 (sim_g256p_layout.py)
__host__ __device__ __forceinline__ int f4(int i) { return (((i ^ (i >> 1)) & 1) << 1) | (i >> 1); }
union Piece {
  i32x4 i;
  unsigned u[4];
};
union Pk {
  bf16x4 b;
  unsigned u[2];
};
__device__ __forceinline__ bf16x4 cvt4(f32x4 v) { return bf16x4{(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]}; }
__device__ __forceinline__ f32x4 cvt4(bf16x4 v) { return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]}; }

// 16 bytes = 8 consecutive columns of this lane's row out of two adjacent 16-column accumulator blocks (lane (l15, lg) holds columns
// 4 lg .. 4 lg + 3 of each): after the swaps an even-lg lane holds columns 8 (lg >> 1) .. + 7 of block x, an odd-lg lane those of block y
__device__ __forceinline__ Piece make_piece(f32x4 x, f32x4 y, f32x4 bx, f32x4 by) {
  Pk a, b;
  a.b = cvt4(x + bx);
  b.b = cvt4(y + by);
  const u32x2 r0 = __builtin_amdgcn_permlane16_swap(a.u[0], b.u[0], false, false);
  const u32x2 r1 = __builtin_amdgcn_permlane16_swap(a.u[1], b.u[1], false, false);
  Piece p;
  p.u[0] = r0[0];
  p.u[1] = r1[0];
  p.u[2] = r0[1];
  p.u[3] = r1[1];
  return p;
}

// PARK = 1: the full scheme.  PARK = 0: the same persistent ring and main loop, but a finished tile is converted and stored at once
// (exposed) -- the reference point that separates "does the ring / the way out work" from "does the parked epilogue work", and the
// denominator of what parking buys.
template <int PARK>
__global__ __launch_bounds__(256) void g256p(const bf16* __restrict__ A, const bf16* __restrict__ B, const bf16* __restrict__ bias,
                                             bf16* __restrict__ C, int M, int N, int K, int tiles_n, int ntiles, int GM) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int l15 = lane & 15, lg = lane >> 4;
  const int nk = K / 64, nsub = 2 * nk;
  const int G = gridDim.x;
  const long long lda = K, ldb = K, ldc = N;
  const i32x4 ra = raw_rsrc(A, (uint32_t)((long long)M * K * 2)), rb = raw_rsrc(B, (uint32_t)((long long)N * K * 2));
  const uint32_t smem_base = (uint32_t)(uintptr_t)(lds_void*)smem;

  // ---- issue stream: K-step `it_ks` of tile `it_tile` goes to ring slot `islot`; it runs three K-steps ahead of the consumer and
  // simply continues into the workgroup's next tile
  // XCD-aware walk (round 5; -DABL_FLATWALK restores tile = blockIdx.x + i * G): workgroup b runs on XCD b % 8; every XCD owns a
  // contiguous eighth of the tile ids in GROUPED order (GM m-tiles x all n-tiles, as csrc/gemm256.hip walks them), and the nx
  // workgroups of an XCD take ids j, j + nx, j + 2 nx, ... of that range -- at any moment they work on ~32 neighbouring tiles whose A and
  // B panels their L2 fetches once.  (The flat walk gives the 3 / 9 tiles that share an A panel to 3 / 8 different XCDs.)
  const int tiles_m = ntiles / tiles_n;
#ifdef ABL_FLATWALK
  const int x_start = 0, x_count = ntiles, nx = G, jx = blockIdx.x;
#else
  const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
  const int q8 = ntiles >> 3, r8 = ntiles & 7;
  const int x_start = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
  const int x_count = q8 + (xcd < r8 ? 1 : 0);
  const int nx = (G - xcd + 7) >> 3;
#endif
  auto decode = [&](int w, int& tm, int& tn) __attribute__((always_inline)) {
#ifdef ABL_FLATWALK
    tm = w / tiles_n;
    tn = w - tm * tiles_n;
#else
    const int pid = x_start + w, gsz = GM * tiles_n;
    const int grp = pid / gsz, rem = pid - grp * gsz;
    const int gm = min(GM, tiles_m - grp * GM);
    tn = rem / gm;
    tm = grp * GM + (rem - tn * gm);
#endif
  };
  int it_tile = jx, it_ks = 0, islot = 0;
  // per-lane part of the DMA source offsets (row inside the tile, swizzled 16-byte chunk): constant for the whole kernel; the tile's
  // origin and the K-step travel in the SCALAR offset of the DMA instruction, so moving on to the next tile is SALU work only
  // (one VGPR per operand: the four instructions of a wave cover rows 16 j apart -- a scalar)
#ifdef ABL_FULLLINE
  // (attribution, TIMING ONLY: the same bytes and the same number of DMA instructions per K-step, but every instruction reads 8 rows x
  // one whole 128-byte line instead of 16 rows x half a line -- K-step s fetches rows [128 (s & 1), +128) of the K = 64 column pair
  // s >> 1, so that two K-steps cover each line exactly once)
  const uint32_t vo_lane = (uint32_t)((wave * 32 + (lane >> 3)) * K * 2 + (lane & 7) * 16);
  const uint32_t jstride = (uint32_t)(8 * K * 2);
#else
  const uint32_t vo_lane = (uint32_t)((wave * 64 + (lane >> 2)) * K * 2 + (((lane & 3) ^ f4((lane >> 4) & 3)) * 16));      // = slot ^ f4((r >> 2) & 3)
  const uint32_t jstride = (uint32_t)(16 * K * 2);
#endif
  uint32_t baseA = 0, baseB = 0;
  auto set_tile = [&](int t) __attribute__((always_inline)) {
    int tm, tn;
    decode(t, tm, tn);
    baseA = (uint32_t)((long long)tm * TM * lda * 2);
    baseB = (uint32_t)((long long)tn * TN * ldb * 2);
  };
  if (it_tile < x_count) set_tile(it_tile);
  // One K-step's eight DMA instructions are issued ONE PER COLUMN BLOCK, behind that block's eight MFMAs (a single wave feeds its SIMD:
  // issued in one batch in front of the step they would cost the matrix pipe ~100 idle clocks per K-step).
  uint32_t is_vo = 0, is_sa = 0, is_sb = 0, is_lds = 0;
  auto issue_begin = [&]() __attribute__((always_inline)) {
    const bool live = it_tile < x_count;                             // wave-uniform
#ifdef ABL_FULLLINE
    const uint32_t ko = (uint32_t)((it_ks >> 1) * 128 + (it_ks & 1) * 128 * K * 2);
#else
    const uint32_t ko = (uint32_t)(it_ks * SUBK * 2);
#endif
#ifdef ABL_OOBDMA
    is_vo = vo_lane | OOB;                                           // (attribution: the DMA instructions are issued, nothing is fetched)
#else
    is_vo = vo_lane | (live ? 0u : OOB);
#endif                             // past the last tile: zero fill, no traffic (the counted waits stay uniform)
    is_sa = live ? baseA + ko : 0u;
    is_sb = live ? baseB + ko : 0u;
    is_lds = smem_base + (uint32_t)((islot * 2) * UNIT + wave * 4096);
  };
  auto issue_one = [&](auto I) __attribute__((always_inline)) {                                     // i = 0..3: A rows, 4..7: B rows
    constexpr int i = decltype(I)::value;
#if defined(ABL_NODMA)
    (void)i;
#elif defined(ABL_M0X4)
    constexpr int j = i & 3;
    if constexpr (j == 0) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" : : "s"(__builtin_amdgcn_readfirstlane(is_lds + (uint32_t)(i < 4 ? 0 : UNIT))) : "memory");
    const uint32_t so = (i < 4 ? is_sa : is_sb) + (uint32_t)j * jstride - (uint32_t)(j * 1024);
    if constexpr (j == 0) asm volatile("buffer_load_dwordx4 %0, %1, %2 offen lds" : : "v"(is_vo), "s"(i < 4 ? ra : rb), "s"(__builtin_amdgcn_readfirstlane(so)) : "memory");
    else if constexpr (j == 1) asm volatile("buffer_load_dwordx4 %0, %1, %2 offen offset:1024 lds" : : "v"(is_vo), "s"(i < 4 ? ra : rb), "s"(__builtin_amdgcn_readfirstlane(so)) : "memory");
    else if constexpr (j == 2) asm volatile("buffer_load_dwordx4 %0, %1, %2 offen offset:2048 lds" : : "v"(is_vo), "s"(i < 4 ? ra : rb), "s"(__builtin_amdgcn_readfirstlane(so)) : "memory");
    else asm volatile("buffer_load_dwordx4 %0, %1, %2 offen offset:3072 lds" : : "v"(is_vo), "s"(i < 4 ? ra : rb), "s"(__builtin_amdgcn_readfirstlane(so)) : "memory");
#else
    if constexpr (i < 4) dma16(ra, is_lds + (uint32_t)(i * 1024), is_vo, is_sa + (uint32_t)i * jstride);
    else dma16(rb, is_lds + (uint32_t)(UNIT + (i - 4) * 1024), is_vo, is_sb + (uint32_t)(i - 4) * jstride);
#endif
  };
  auto issue_end = [&]() __attribute__((always_inline)) {
    islot = islot == NSLOT - 1 ? 0 : islot + 1;
    if (it_tile < x_count) {
      if (++it_ks == nsub) {
        it_ks = 0;
        it_tile += nx;
        if (it_tile < x_count) set_tile(it_tile);
      }
    }
  };
  auto issue_next = [&]() __attribute__((always_inline)) {                                          // (prologue only: a whole K-step at once)
    issue_begin();
    issue_one(IC<0>{}); issue_one(IC<1>{}); issue_one(IC<2>{}); issue_one(IC<3>{});
    issue_one(IC<4>{}); issue_one(IC<5>{}); issue_one(IC<6>{}); issue_one(IC<7>{});
    issue_end();
  };

  // ---- consumer: fragments of K-step g live in buffer g & 1 (read one K-step ahead, under the MFMAs of g - 1)
  const int swz = (lg ^ f4((l15 >> 2) & 3)) << 4;
  const int a_lane = (wr * 128 + l15) * 64 + swz, b_lane = UNIT + (wc * 128 + l15) * 64 + swz;
  int rslot = 0;                                                     // ring slot of the NEXT K-step to read
  // A fragments double-buffered (all eight are needed until a K-step's last MFMA), B fragments single-buffered: column block nb of
  // the NEXT K-step is read into the registers of the current one as soon as its eight MFMAs have been issued
  // (column block 7 is double-buffered like A: its refill would otherwise be the last instruction of a K-step, and the step's closing
  // lgkmcnt(0) would wait a whole LDS latency for it; behind column 6's refill come the eight MFMAs of column 7)
  bf16x8 fa[2][8], fb[7], fb7[2];
  const char* nbase = smem;                                          // slot of the K-step being read
  auto next_slot = [&]() __attribute__((always_inline)) {
    nbase = smem + rslot * 2 * UNIT;
    rslot = rslot == NSLOT - 1 ? 0 : rslot + 1;
  };
  auto read_a = [&](auto BUF) __attribute__((always_inline)) {
    constexpr int buf = decltype(BUF)::value;
#pragma unroll
    for (int i = 0; i < 8; ++i) fa[buf][i] = *(const bf16x8*)(nbase + a_lane + i * 1024);
  };
  auto read_a_one = [&](auto BUF, int i) __attribute__((always_inline)) { fa[decltype(BUF)::value][i] = *(const bf16x8*)(nbase + a_lane + i * 1024); };
  auto read_b = [&](int nb) __attribute__((always_inline)) { fb[nb] = *(const bf16x8*)(nbase + b_lane + nb * 1024); };      // nb < 7
  auto read_b7 = [&](auto BUF) __attribute__((always_inline)) { fb7[decltype(BUF)::value] = *(const bf16x8*)(nbase + b_lane + 7 * 1024); };
  f32x4 acc[8][8];

  // ---- the previous tile, on its way out: pieces q = p * 8 + mb (column pair p, row block mb); q < 16 in registers, the rest in the
  // wave's 16 KiB of the LDS park
  Piece park[16];
  // lane-dependent addresses of the way out are recomputed where they are used (a few VALU instructions from a fresh lane id): kept
  // live through the K loop they were spilled, and a scratch reload waits with vmcnt(0) -- for the whole DMA ring
  auto fresh_lane = [&]() __attribute__((always_inline)) {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
  };
  auto park_addr = [&]() __attribute__((always_inline)) -> char*  { return smem + RING + wave * 16384 + fresh_lane() * 16; };
  auto c_lane_off = [&]() __attribute__((always_inline)) {
    const int l = fresh_lane();
    const int r15 = l & 15, g4 = l >> 4;
    return (uint32_t)((((long long)(wr * 128 + r15)) * ldc + wc * 128 + (g4 & 1) * 16 + (g4 >> 1) * 8) * 2);
  };
  const uint32_t tile_bytes = (uint32_t)((255 * ldc + 256) * 2);
  __amdgpu_buffer_rsrc_t c_prev = make_rsrc(C, 0);
  bool prev_valid = false;
  auto store_piece = [&](auto Q, const char* pk, uint32_t cl) __attribute__((always_inline)) {      // one parked piece -> C of the previous tile
    constexpr int q = decltype(Q)::value, p = q >> 3, mb = q & 7;
    Piece v;
    if constexpr (q < 16) v = park[q];
    else v.i = *(const i32x4*)(pk + (q - 16) * 1024);
    __builtin_amdgcn_raw_buffer_store_b128(v.i, c_prev, cl + (uint32_t)(p * 64), (uint32_t)(mb * 16 * ldc * 2), 0);
  };

  // bias of a tile as raw bf16 (16 registers), requested one K-step before the tile ends
  bf16x4 braw[8];
  const __amdgpu_buffer_rsrc_t bias_rs = make_rsrc(bias, (uint32_t)N * 2u);
  // Inline asm on purpose: a load the compiler knows about is waited for with ITS count of younger requests -- it cannot see the
  // LDS-DMA instructions, so `s_waitcnt vmcnt(0)` in front of the first use drained the whole ring once per tile.  These are
  // requested in front of the second-to-last K-step's DMA batch; that step's own vmcnt(8) retires them, one K-step before they are used.
  auto load_bias = [&](int tn) __attribute__((always_inline)) {
    const uint32_t lo = (uint32_t)((wc * 128 + (fresh_lane() >> 4) * 4) * 2), so = (uint32_t)(tn * TN * 2);
    typedef __attribute__((ext_vector_type(2))) unsigned u2;
    u2 t0, t1, t2, t3, t4, t5, t6, t7;
#define BIAS_LD(t, i) asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen offset:" #i : "=v"(t) : "v"(lo), "s"(bias_rs), "s"(__builtin_amdgcn_readfirstlane(so)) : "memory")
    BIAS_LD(t0, 0); BIAS_LD(t1, 32); BIAS_LD(t2, 64); BIAS_LD(t3, 96); BIAS_LD(t4, 128); BIAS_LD(t5, 160); BIAS_LD(t6, 192); BIAS_LD(t7, 224);
#undef BIAS_LD
    braw[0] = __builtin_bit_cast(bf16x4, t0); braw[1] = __builtin_bit_cast(bf16x4, t1); braw[2] = __builtin_bit_cast(bf16x4, t2);
    braw[3] = __builtin_bit_cast(bf16x4, t3); braw[4] = __builtin_bit_cast(bf16x4, t4); braw[5] = __builtin_bit_cast(bf16x4, t5);
    braw[6] = __builtin_bit_cast(bf16x4, t6); braw[7] = __builtin_bit_cast(bf16x4, t7);
  };

  // ---- one K-step: issue three ahead, carry two parked pieces out, read the next step's fragments, 64 MFMAs, counted wait, barrier
  // column block nb of a K-step: eight MFMAs, and in their shadow (one wave feeds its SIMD: whatever is issued in FRONT of a step's
  // first MFMA idles the matrix pipe) one of everything the step owes: the refill of this column's B fragment and of A fragment nb
  // for the next K-step, one DMA instruction, and -- behind columns 1 and 2 -- one parked piece of the previous tile on its way out
  auto mma_col = [&](auto BUF, auto NB, auto PF, auto TRK) __attribute__((always_inline)) {
    constexpr int buf = decltype(BUF)::value, nb = decltype(NB)::value, trk = decltype(TRK)::value;
    constexpr bool pf = decltype(PF)::value != 0;
#pragma unroll
    for (int mb = 0; mb < 8; ++mb) {
      if constexpr (nb < 7) acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[nb], fa[buf][mb], acc[mb][nb], 0, 0, 0);
      else acc[mb][7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb7[buf], fa[buf][mb], acc[mb][7], 0, 0, 0);
    }
#if defined(ABL_NOREAD)
    constexpr bool do_reads = false;
#else
    constexpr bool do_reads = true;
#endif
    if constexpr (pf && do_reads) {
      if constexpr (nb < 7) read_b(nb);
      if constexpr (nb == 0) {           // (nothing is read behind the LAST column: the step's closing lgkmcnt(0) would wait for it)
        read_a_one(IC<buf ^ 1>{}, 0);
        read_b7(IC<buf ^ 1>{});
      }
      if constexpr (nb < 7) read_a_one(IC<buf ^ 1>{}, nb + 1);
    }
    if constexpr (PARK && trk >= 0 && (nb == 1 || nb == 2)) {
      if (prev_valid) store_piece(IC<(trk >= 0 ? trk : 0) + (nb - 1)>{}, park_addr(), c_lane_off());
    }
    issue_one(NB);
  };
  auto mma = [&](auto BUF, auto PF, auto TRK) __attribute__((always_inline)) {
    mma_col(BUF, IC<0>{}, PF, TRK); mma_col(BUF, IC<1>{}, PF, TRK); mma_col(BUF, IC<2>{}, PF, TRK); mma_col(BUF, IC<3>{}, PF, TRK);
    mma_col(BUF, IC<4>{}, PF, TRK); mma_col(BUF, IC<5>{}, PF, TRK); mma_col(BUF, IC<6>{}, PF, TRK); mma_col(BUF, IC<7>{}, PF, TRK);
    issue_end();
  };
  auto end_step = [&]() __attribute__((always_inline)) {
    SB();
#ifdef ABL_NOWAIT
    __builtin_amdgcn_s_waitcnt(0xC07F);      // (attribution: lgkmcnt(0) only -- the fragments are read whether their batch has landed or not)
#else
    __builtin_amdgcn_s_waitcnt(0x0078);      // vmcnt(8): at most
#endif
    // ... at most the eight youngest requests are still out -- this step's DMA batch (its first
                                             // two instructions too must have landed in a step that also trickled two stores: harmless,
                                             // they are ~900 clocks old); loads complete in order, so everything older has landed whatever
                                             // the stores do; lgkmcnt(0): this wave's reads of the slot the next K-step refills are done
#if !defined(ABL_NOBAR)
    __builtin_amdgcn_s_barrier();
#endif
    SB();
  };
  auto substep = [&](auto BUF, auto TRK) __attribute__((always_inline)) {
    issue_begin();
    next_slot();
    mma(BUF, IC<1>{}, TRK);
    end_step();
  };
  // The finished tile leaves the accumulators DURING its last K-step: a column pair is final as soon as its sixteen MFMAs of that step
  // are issued, and is converted (AGPR -> VGPR, + bias, -> bf16, two column blocks merged into 16-byte row pieces) one pair behind, in
  // the shadow of the later pairs' MFMAs -- only the last pair's conversion is exposed.  (Converting in front of the NEXT tile's first
  // MFMAs into the same registers is what one would write in assembly; hipcc keeps old and new accumulators apart and spills.)
  // Pieces of pairs 0, 1 stay in registers, those of pairs 2, 3 go to the wave's LDS park.
  auto convert_pair = [&](auto P, char* pk_w) __attribute__((always_inline)) {
    constexpr int p = decltype(P)::value;
    const f32x4 b0 = cvt4(braw[2 * p]), b1 = cvt4(braw[2 * p + 1]);
#pragma unroll
    for (int mb = 0; mb < 8; ++mb) {
      const Piece v = make_piece(acc[mb][2 * p], acc[mb][2 * p + 1], b0, b1);
      if constexpr (p < 2) park[p * 8 + mb] = v;
      else *(i32x4*)(pk_w + ((p - 2) * 8 + mb) * 1024) = v.i;
    }
  };
  // the last K-step of a tile: no fragments fetched ahead (the first step of the next tile reads its own), conversion in the shadow
  auto last_substep = [&]() __attribute__((always_inline)) {
    issue_begin();
    char* pk_w = park_addr();
    mma_col(IC<1>{}, IC<0>{}, IC<0>{}, IC<-1>{});
    mma_col(IC<1>{}, IC<1>{}, IC<0>{}, IC<-1>{});
    SB();
    mma_col(IC<1>{}, IC<2>{}, IC<0>{}, IC<-1>{});
    mma_col(IC<1>{}, IC<3>{}, IC<0>{}, IC<-1>{});
    convert_pair(IC<0>{}, pk_w);
    SB();
    mma_col(IC<1>{}, IC<4>{}, IC<0>{}, IC<-1>{});
    mma_col(IC<1>{}, IC<5>{}, IC<0>{}, IC<-1>{});
    convert_pair(IC<1>{}, pk_w);
    SB();
    mma_col(IC<1>{}, IC<6>{}, IC<0>{}, IC<-1>{});
    convert_pair(IC<2>{}, pk_w);
    SB();
    mma_col(IC<1>{}, IC<7>{}, IC<0>{}, IC<-1>{});
    issue_end();
    convert_pair(IC<3>{}, pk_w);                         // (exposed: the last pair is only final now)
    end_step();
  };
  // K-step 0 of a tile: its fragments are read here (the previous step fetched nothing ahead), C = 0
  auto first_step = [&]() __attribute__((always_inline)) {
    auto first_pair = [&](auto P) __attribute__((always_inline)) {
      constexpr int p = decltype(P)::value;
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int mb = 0; mb < 8; ++mb) {
        acc[mb][2 * p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[2 * p], fa[0][mb], z, 0, 0, 0);
        if constexpr (p < 3) acc[mb][2 * p + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[2 * p + 1], fa[0][mb], z, 0, 0, 0);
        else acc[mb][7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb7[0], fa[0][mb], z, 0, 0, 0);
      }
      read_b(2 * p);
      if constexpr (p < 3) read_b(2 * p + 1);
      read_a_one(IC<1>{}, 2 * p);
      read_a_one(IC<1>{}, 2 * p + 1);
      if constexpr (p == 0) read_b7(IC<1>{});
      issue_one(IC<2 * p>{});
      issue_one(IC<2 * p + 1>{});
    };
    next_slot();                                       // K-step 0 of the new tile (landed and published two barriers ago)
    read_a(IC<0>{});
#pragma unroll
    for (int nb = 0; nb < 7; ++nb) read_b(nb);
    read_b7(IC<0>{});
    __builtin_amdgcn_s_waitcnt(0xC07F);                // lgkmcnt(0) + barrier: this step's DMA batch refills the very slot just read
    __builtin_amdgcn_s_barrier();                      // (in every other K-step the fragments were read one step -- one barrier -- earlier)
    issue_begin();
    next_slot();                                       // ... and K-step 1 rolls in behind its MFMAs
    first_pair(IC<0>{});
    first_pair(IC<1>{});
    first_pair(IC<2>{});
    first_pair(IC<3>{});
    issue_end();
    end_step();
  };

  auto flush_all = [&]() __attribute__((always_inline)) {
    const char* pk = park_addr();
    const uint32_t cl = c_lane_off();
    store_piece(IC<0>{}, pk, cl); store_piece(IC<1>{}, pk, cl); store_piece(IC<2>{}, pk, cl); store_piece(IC<3>{}, pk, cl);
    store_piece(IC<4>{}, pk, cl); store_piece(IC<5>{}, pk, cl); store_piece(IC<6>{}, pk, cl); store_piece(IC<7>{}, pk, cl);
    store_piece(IC<8>{}, pk, cl); store_piece(IC<9>{}, pk, cl); store_piece(IC<10>{}, pk, cl); store_piece(IC<11>{}, pk, cl);
    store_piece(IC<12>{}, pk, cl); store_piece(IC<13>{}, pk, cl); store_piece(IC<14>{}, pk, cl); store_piece(IC<15>{}, pk, cl);
    store_piece(IC<16>{}, pk, cl); store_piece(IC<17>{}, pk, cl); store_piece(IC<18>{}, pk, cl); store_piece(IC<19>{}, pk, cl);
    store_piece(IC<20>{}, pk, cl); store_piece(IC<21>{}, pk, cl); store_piece(IC<22>{}, pk, cl); store_piece(IC<23>{}, pk, cl);
    store_piece(IC<24>{}, pk, cl); store_piece(IC<25>{}, pk, cl); store_piece(IC<26>{}, pk, cl); store_piece(IC<27>{}, pk, cl);
    store_piece(IC<28>{}, pk, cl); store_piece(IC<29>{}, pk, cl); store_piece(IC<30>{}, pk, cl); store_piece(IC<31>{}, pk, cl);
  };

  // ---- prologue (once per workgroup): K-steps 0, 1, 2 of the first tile; fragments of K-step 0
  issue_next();
  issue_next();
  issue_next();
  __builtin_amdgcn_s_waitcnt(0x0F70);        // (vmcnt(0): simple, once)
  __builtin_amdgcn_s_barrier();

  for (int tile = jx; tile < x_count; tile += nx) {
    int tm, tn;
    decode(tile, tm, tn);
    first_step();                            // K-step 0
    substep(IC<1>{}, IC<0>{});               // K-step 1: pieces 0, 1
    // K-steps 2 .. 16: pieces 2 (s - 1), 2 (s - 1) + 1
    substep(IC<0>{}, IC<2>{});
    substep(IC<1>{}, IC<4>{});
    substep(IC<0>{}, IC<6>{});
    substep(IC<1>{}, IC<8>{});
    substep(IC<0>{}, IC<10>{});
    substep(IC<1>{}, IC<12>{});
    substep(IC<0>{}, IC<14>{});
    substep(IC<1>{}, IC<16>{});
    substep(IC<0>{}, IC<18>{});
    substep(IC<1>{}, IC<20>{});
    substep(IC<0>{}, IC<22>{});
    substep(IC<1>{}, IC<24>{});
    substep(IC<0>{}, IC<26>{});
    substep(IC<1>{}, IC<28>{});
    substep(IC<0>{}, IC<30>{});
    substep(IC<1>{}, IC<-1>{});              // K-step 17
#pragma unroll 1
    for (int kt = 9; kt < nk - 1; ++kt) {
      substep(IC<0>{}, IC<-1>{});
      substep(IC<1>{}, IC<-1>{});
    }
    load_bias(tn);                                       // this tile's bias, in front of the second-to-last K-step's DMA batch (that step's vmcnt(8) retires it)
    substep(IC<0>{}, IC<-1>{});
    last_substep();                                      // ... and converts the finished tile
    // this tile becomes the one on its way out
    c_prev = make_rsrc(C + ((long long)tm * TM * ldc + tn * TN), tile_bytes);
    prev_valid = true;
    if constexpr (!PARK) flush_all();        // reference point: the tile is stored here, exposed
  }
  // ---- after the last tile: convert, park and store everything (exposed: nothing left to hide it under)
  if constexpr (PARK) flush_all();
  __builtin_amdgcn_s_waitcnt(0x0F70);        // the dead DMA units have been zero-filled before the workgroup leaves
}

// ------------------------------------------------------------------------------------------------------------------- host
static uint16_t f2bf_host(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static float bf2f_host(uint16_t h) {
  uint32_t u = (uint32_t)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 50432, N = argc > 2 ? atoi(argv[2]) : 2304, K = argc > 3 ? atoi(argv[3]) : 768;
  if (M % TM || N % TN || K % 64 || K < 64 * 10) {
    printf("M and N must be multiples of 256, K a multiple of 64 and >= 640 (M=%d would need %d padding rows)\n", M, (TM - M % TM) % TM);
    return 1;
  }
  std::vector<uint16_t> ha((size_t)M * K), hb((size_t)N * K), hbias(N), hc((size_t)M * N);
  uint32_t st = 12345;
  auto rnd = [&]() __attribute__((always_inline)) { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xffff) / 65536.0f - 0.5f; };
  for (auto& v : ha) v = f2bf_host(rnd());
  for (auto& v : hb) v = f2bf_host(rnd());
  for (auto& v : hbias) v = f2bf_host(rnd());
  bf16 *dA, *dB, *dC, *dBias;
  hipMalloc(&dA, ha.size() * 2); hipMalloc(&dB, hb.size() * 2); hipMalloc(&dC, hc.size() * 2); hipMalloc(&dBias, hbias.size() * 2);
  hipMemcpy(dA, ha.data(), ha.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(dB, hb.data(), hb.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(dBias, hbias.data(), hbias.size() * 2, hipMemcpyHostToDevice);
  hipMemset(dC, 0xff, hc.size() * 2);
  (void)hipFuncSetAttribute((const void*)g256p<0>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
  (void)hipFuncSetAttribute((const void*)g256p<1>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
  int dev = 0, ncu = 256;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ncu = prop.multiProcessorCount;
  const int tiles_m = M / TM, tiles_n = N / TN, ntiles = tiles_m * tiles_n;
  const int grid = ntiles < ncu ? ntiles : ncu;
  const int only = argc > 4 ? atoi(argv[4]) : -1;            // 0 / 1: run one variant only
  const int gm_arg = argc > 5 ? atoi(argv[5]) : 4;           // m-tiles per group of the XCD-aware walk
  for (int park = 0; park < 2; ++park) {
    if (only >= 0 && only != park) continue;
    auto launch = [&]() {
      if (park) hipLaunchKernelGGL(g256p<1>, dim3(grid), dim3(256), SMEM, 0, dA, dB, dBias, dC, M, N, K, tiles_n, ntiles, gm_arg);
      else hipLaunchKernelGGL(g256p<0>, dim3(grid), dim3(256), SMEM, 0, dA, dB, dBias, dC, M, N, K, tiles_n, ntiles, gm_arg);
    };
    hipMemset(dC, 0xff, hc.size() * 2);
    launch();
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { printf("park=%d: launch failed: %s\n", park, hipGetErrorString(e)); return 1; }
    hipMemcpy(hc.data(), dC, hc.size() * 2, hipMemcpyDeviceToHost);
    double maxerr = 0;
    uint32_t sv = 777;
    for (int t = 0; t < 6000; ++t) {
      sv = sv * 1664525u + 1013904223u; int m = (sv >> 4) % M;
      sv = sv * 1664525u + 1013904223u; int n = (sv >> 4) % N;
      if (t < 512) { m = (t & 1) ? M - 1 - (t >> 1) % 256 : (t >> 1) % 256; n = (t * 37) % N; }      // first and last tile rows, every column class
      double ref = bf2f_host(hbias[n]);
      for (int k = 0; k < K; ++k) ref += (double)bf2f_host(ha[(size_t)m * K + k]) * bf2f_host(hb[(size_t)n * K + k]);
      const double err = fabs(ref - bf2f_host(hc[(size_t)m * N + n])) / (fabs(ref) + 1.0);
      if (!(err <= maxerr)) maxerr = err;
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) launch();
    hipEventRecord(e0);
    const int it = 20;
    for (int i = 0; i < it; ++i) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double t = ms / it * 1e-3;
    printf("g256p<park=%d> (4 waves x 128x128, persistent ring%s; %d workgroups for %d tiles) M=%d N=%d K=%d: %.1f us  %.1f TFLOP/s  max rel err %.3g %s\n", park,
           park ? ", parked epilogue" : ", exposed epilogue", grid, ntiles, M, N, K, t * 1e6, 2.0 * M * N * K / t / 1e12, maxerr, ABL_WRONG ? "(ablation: timing only)" : maxerr < 2e-2 ? "(ok)" : "(WRONG)");
  }
  return 0;
}
