// g128x256_probe.hip -- does a SECOND workgroup on the CU hide the prologue / epilogue of a short-K GEMM tile?
//
// The production kernel (csrc/gemm256.hip) owns a CU with one 8-wave workgroup (135 KiB of LDS, 256 registers per wave):
// at K = 768 a tile is 2.2 us prologue + 15.7 us loop + 4.9-10 us store-bound epilogue, and nothing overlaps the first and the
// last.  This probe is the other shape the register file allows: FOUR waves per workgroup (1 x 4, the
// same 128 x 64 wave tile), a 128 x 256 output tile, K steps of 32 through a 3-stage LDS-DMA ring of 24 KiB stages (72 KiB:
// two workgroups per CU, 8 waves), NT form (both operands k-contiguous), bias + bf16 epilogue staged through LDS and
// stored row-contiguously in 16-byte pieces -- enough of a real tile to compare with the production kernel on the ViT
// forward shapes.  Not a product path.
// Build: hipcc --offload-arch=gfx950 -O3 -o g128x256_probe g128x256_probe.hip ; run: ./g128x256_probe M N K
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) float f32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(3))) void lds_void;

constexpr int BM = 128, BN = 256, BK = 32;
constexpr int A_BYTES = BM * BK * 2;             // 8 KiB
constexpr int B_BYTES = BN * BK * 2;             // 16 KiB
constexpr int STAGE = A_BYTES + B_BYTES;         // 24 KiB
constexpr int NSTAGE = 3;
constexpr int CPITCH = BN * 2 + 16;              // staged output tile: 128 rows x 528 B = 66 KiB (inside the 72 KiB ring)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
#define SB() __builtin_amdgcn_sched_barrier(0)
typedef __attribute__((ext_vector_type(4))) int i32x4;
// LDS-DMA as inline asm: hipcc then neither waits for these loads on its own nor orders its LDS reads behind them; the waits
// are the explicit s_waitcnt vmcnt(N) of the loop
__device__ __forceinline__ void dma16(i32x4 rsrc, uint32_t lds_addr, uint32_t voff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" : : "s"(lds_addr), "v"(voff), "s"(rsrc) : "memory");
}
__device__ __forceinline__ i32x4 raw_rsrc(const void* ptr, uint32_t bytes) {
  const uint64_t a = (uint64_t)ptr;
  return i32x4{(int)(uint32_t)a, (int)((uint32_t)(a >> 32) & 0xffffu), (int)bytes, 0x00020000};
}

#ifndef WPE
#define WPE 2
#endif
#ifndef VARIANT
#define VARIANT 1
#endif

__global__ __launch_bounds__(256, WPE) void g128x256(const bf16* __restrict__ A, const bf16* __restrict__ B, const bf16* __restrict__ bias,
                                                     bf16* __restrict__ C, int M, int N, int K, int tiles_m, int tiles_n) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // workgroup -> tile: contiguous range per XCD (workgroup b runs on XCD b % 8), n fastest inside it (the n-tiles of one
  // m-tile share their A rows in that XCD's L2)
  const int nwg = tiles_m * tiles_n;
  const int bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
  const int lin = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int tile_m = lin / tiles_n, tile_n = lin - tile_m * tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int nk = K / BK;

  const i32x4 ra = raw_rsrc(A, (uint32_t)((size_t)M * K * 2));
  const i32x4 rb = raw_rsrc(B, (uint32_t)((size_t)N * K * 2));

  // DMA: a 1 KiB piece = 16 rows x 64 B.  A stage = 8 pieces (wave w: pieces 2w, 2w+1), B stage = 16 pieces (wave w: 4w..4w+3).
  // lane -> row 16 p + l / 4; LDS slot l % 4 holds source chunk (l % 4) ^ ((row >> 2) & 3)
  uint32_t off[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const bool isA = j < 2;
    const int piece = isA ? wave * 2 + j : wave * 4 + (j - 2);
    const int row = piece * 16 + (lane >> 2);
    const int kc = (lane & 3) ^ ((lane >> 4) & 3);
    if (isA) off[j] = (m0 + row) < M ? (uint32_t)(((size_t)(m0 + row) * K + kc * 8) * 2) : 0x80000000u;
    else off[j] = (n0 + row) < N ? (uint32_t)(((size_t)(n0 + row) * K + kc * 8) * 2) : 0x80000000u;
  }
  auto dma_piece = [&](int kt, int j) {      // piece j (0..5) of this wave's share of stage kt
    const uint32_t kb = kt < nk ? (uint32_t)(kt * BK * 2) : 0x80000000u;
    char* st = smem + (kt % NSTAGE) * STAGE;
    char* dst = j < 2 ? st + (wave * 2 + j) * 1024 : st + A_BYTES + (wave * 4 + (j - 2)) * 1024;
    dma16(j < 2 ? ra : rb, (uint32_t)(uintptr_t)(lds_void*)dst, off[j] + kb);
  };
  auto dma_stage = [&](int kt) {
    const uint32_t kb = kt < nk ? (uint32_t)(kt * BK * 2) : 0x80000000u;
    char* st = smem + (kt % NSTAGE) * STAGE;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      char* dst = j < 2 ? st + (wave * 2 + j) * 1024 : st + A_BYTES + (wave * 4 + (j - 2)) * 1024;
      dma16(j < 2 ? ra : rb, (uint32_t)(uintptr_t)(lds_void*)dst, off[j] + kb);
    }
  };

  // fragment of a 32-row block, k-sub s (16 of the 32 k): lane row (lane & 31), 16-byte chunk s * 2 + (lane >> 5), swizzled
  const int sw = ((lane & 31) >> 2) & 3;
  int fo[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) fo[s] = (lane & 31) * 64 + (((s * 2 + (lane >> 5)) ^ sw) << 4);
  const int bbase = A_BYTES + wave * 64 * 64;      // this wave's 64 B rows (= output columns)

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  dma_stage(0);
  dma_stage(1);
  // one K step: {wait for stage kt; barrier; DMA of stage kt + 2; 12 fragment reads; 16 MFMAs}.  VARIANT 1 issues the whole
  // fragment set before the first MFMA (48 registers) and raises the wave priority over the MFMA run -- the other workgroup
  // of the CU is meant to fill the read latency; VARIANT 0 leaves the interleaving to the compiler (it keeps 16 fragment
  // registers and alternates reads and MFMAs).
  auto step = [&](int kt, const char* st) {
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#if VARIANT < 2
    dma_stage(kt + 2);      // (past the end: every lane out of range -> no traffic, the wait counts stay uniform)
#endif
    bf16x8 fa[2][4], fb[2][2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
      for (int j = 0; j < 2; ++j) fb[s2][j] = *(const bf16x8*)(st + bbase + j * 2048 + fo[s2]);
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[s2][i] = *(const bf16x8*)(st + i * 2048 + fo[s2]);
    }
#if VARIANT >= 1
    SB();
    __builtin_amdgcn_s_setprio(1);
#endif
#if VARIANT >= 2      // VARIANT 2: the six DMA pieces of stage kt + 2 woven between the MFMAs (one after every second MFMA) instead of in a burst
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int s2 = q >> 3, i = (q >> 1) & 3, j = q & 1;
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[s2][j], fa[s2][i], acc[i][j], 0, 0, 0);
      SB();
      if ((q & 1) && q < 12) {
        dma_piece(kt + 2, q >> 1);
        SB();
      }
    }
#else
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[s2][j], fa[s2][i], acc[i][j], 0, 0, 0);
#endif
#if VARIANT >= 1
    __builtin_amdgcn_s_setprio(0);
    SB();
#endif
  };
  int kt = 0;
  for (; kt + 3 <= nk; kt += 3) {      // stage slots as compile-time offsets
    step(kt, smem);
    step(kt + 1, smem + STAGE);
    step(kt + 2, smem + 2 * STAGE);
  }
  for (; kt < nk; ++kt) step(kt, smem + (kt % NSTAGE) * STAGE);
  // ---- epilogue: (acc + bias) -> bf16, staged in LDS (the ring is dead), stored row-contiguously in 16-byte pieces
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  bf16x4 braw[2][4];      // all bias pieces requested up front: one round trip (N is a multiple of 256 here)
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int q = 0; q < 4; ++q) braw[j][q] = *(const bf16x4*)(bias + n0 + wave * 64 + j * 32 + 8 * q + 4 * (lane >> 5));
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int nl = wave * 64 + j * 32 + 8 * q + 4 * (lane >> 5);
      const f32x4 bv = __builtin_convertvector(braw[j][q], f32x4);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * q + e] + bv[e];
        *(bf16x4*)(smem + (i * 32 + (lane & 31)) * CPITCH + nl * 2) = __builtin_convertvector(v, bf16x4);
      }
    }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < 16; ++it) {
    const int c = tid + 256 * it;
    const int row = c >> 5, cc = c & 31;
    const int m = m0 + row, n = n0 + cc * 8;
    if (m < M && n < N) *(bf16x8*)(C + (size_t)m * N + n) = *(const bf16x8*)(smem + row * CPITCH + cc * 16);
  }
}

static float bf2f_host(uint16_t b) {
  uint32_t u = (uint32_t)b << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
static uint16_t f2bf_host(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  u += 0x7fff + ((u >> 16) & 1);
  return (uint16_t)(u >> 16);
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 50432, N = argc > 2 ? atoi(argv[2]) : 2304, K = argc > 3 ? atoi(argv[3]) : 768;
  if (K % BK || N % BN) { printf("K must be a multiple of %d, N of %d\n", BK, BN); return 1; }
  std::vector<uint16_t> ha((size_t)M * K), hb((size_t)N * K), hbias(N), hc((size_t)M * N);
  uint32_t st = 12345;
  auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xffff) / 65536.0f - 0.5f; };
  for (auto& v : ha) v = f2bf_host(rnd());
  for (auto& v : hb) v = f2bf_host(rnd());
  for (auto& v : hbias) v = f2bf_host(rnd());
  bf16 *dA, *dB, *dC, *dBias;
  hipMalloc(&dA, ha.size() * 2); hipMalloc(&dB, hb.size() * 2); hipMalloc(&dC, hc.size() * 2); hipMalloc(&dBias, hbias.size() * 2);
  hipMemcpy(dA, ha.data(), ha.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(dB, hb.data(), hb.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(dBias, hbias.data(), hbias.size() * 2, hipMemcpyHostToDevice);
  (void)hipFuncSetAttribute((const void*)g128x256, hipFuncAttributeMaxDynamicSharedMemorySize, NSTAGE * STAGE);
  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
  int occ = 0;
  (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, g128x256, 256, NSTAGE * STAGE);
  auto launch = [&]() { hipLaunchKernelGGL(g128x256, dim3(tiles_m * tiles_n), dim3(256), NSTAGE * STAGE, 0, dA, dB, dBias, dC, M, N, K, tiles_m, tiles_n); };
  launch();
  hipError_t e = hipDeviceSynchronize();
  if (e != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(e)); return 1; }
  hipMemcpy(hc.data(), dC, hc.size() * 2, hipMemcpyDeviceToHost);
  double maxerr = 0;
  for (int t = 0; t < 4000; ++t) {
    st = st * 1664525u + 1013904223u; const int m = (st >> 4) % M;
    st = st * 1664525u + 1013904223u; const int n = (st >> 4) % N;
    double ref = bf2f_host(hbias[n]);
    for (int k = 0; k < K; ++k) ref += (double)bf2f_host(ha[(size_t)m * K + k]) * bf2f_host(hb[(size_t)n * K + k]);
    const double err = fabs(ref - bf2f_host(hc[(size_t)m * N + n])) / (fabs(ref) + 1.0);
    if (err > maxerr) maxerr = err;
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
  printf("g128x256 variant %d (4 waves, %d workgroups per CU by occupancy query) M=%d N=%d K=%d: %.1f us  %.1f TFLOP/s  max rel err %.3g\n", VARIANT, occ, M, N, K, t * 1e6,
         2.0 * M * N * K / t / 1e12, maxerr);
  return 0;
}
