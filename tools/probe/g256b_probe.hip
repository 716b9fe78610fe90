// g256b_probe.hip -- main-loop probe: 256x256x32 NT bf16 GEMM, 4 waves (2x2, each 128x128), one workgroup per CU,
// 4 x 32 KiB LDS stages filled by LDS-DMA; DMA pieces and fragment reads are woven evenly between the MFMAs
// (the 64 B/clk TA path stalls the issuing wave when DMAs come in bursts -- see g256_probe.hip ablations).
// Build: hipcc --offload-arch=gfx950 -O3 -o g256_probe g256_probe.hip ; run: ./g256_probe M N K
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
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(3))) void lds_void;

#ifndef VARIANT
#define VARIANT 0
#endif

constexpr int BM = 256, BN = 256, BK = 32;
constexpr int OPER = 256 * 32 * 2;   // 16 KiB per operand tile
constexpr int STAGE = 2 * OPER;      // 32 KiB
constexpr int NSTAGE = 4;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}

#define SB() __builtin_amdgcn_sched_barrier(0)

__global__ __launch_bounds__(256, 1) void g256(const bf16* __restrict__ A, const bf16* __restrict__ B, bf16* __restrict__ C,
                                               int M, int N, int K, int tiles_n, long long* dbg) {
  const long long c0 = clock64(), w0 = wall_clock64();
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wrow = wave >> 1, wcol = wave & 1;
  const int tile_m = blockIdx.x / tiles_n, tile_n = blockIdx.x % tiles_n;
#ifdef FAKE_L2
  const int m0 = 0, n0 = 0;
#else
  const int m0 = tile_m * BM, n0 = tile_n * BN;
#endif
  const int nk = K / BK;

  const __amdgpu_buffer_rsrc_t ra = make_rsrc(A, (uint32_t)((size_t)M * K * 2));
  const __amdgpu_buffer_rsrc_t rb = make_rsrc(B, (uint32_t)((size_t)N * K * 2));

  // DMA: a 1 KiB piece = 16 rows x 64 B; wave w owns pieces 4w..4w+3 of each operand.  lane -> row 16p + l/4,
  // LDS slot l%4 holds source chunk (l%4) ^ ((row>>2)&3)
  uint32_t off[8];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = (wave * 4 + j) * 16 + (lane >> 2);
    const int kc = (lane & 3) ^ ((lane >> 4) & 3);
    off[j] = (m0 + row) < M ? (uint32_t)(((size_t)(m0 + row) * K + kc * 8) * 2) : 0x80000000u;
    off[4 + j] = (n0 + row) < N ? (uint32_t)(((size_t)(n0 + row) * K + kc * 8) * 2) : 0x80000000u;
  }
  // piece x in 0..7 of K step kt: x<4 -> A piece, else B piece
  auto dma = [&](int kt, int x) {
#ifdef FAKE_L2
    const uint32_t kb = kt < nk ? (uint32_t)((kt & 63) * BK * 2) : 0x80000000u;
#else
    const uint32_t kb = kt < nk ? (uint32_t)(kt * BK * 2) : 0x80000000u;
#endif
    char* dst = smem + (kt & (NSTAGE - 1)) * STAGE + (x >> 2) * OPER + (wave * 4 + (x & 3)) * 1024;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(x < 4 ? ra : rb, (lds_void*)dst, 16, off[x] + kb, 0, 0, 0);
  };

  const int sw = ((lane & 31) >> 2) & 3;
  int fo[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) fo[s] = (lane & 31) * 64 + (((s * 2 + (lane >> 5)) ^ sw) << 4);
  const int abase = wrow * 128 * 64, bbase = OPER + wcol * 128 * 64;

  f32x16 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  bf16x8 fa[2][4], fb[2][4];
  // fragment x in 0..7 (order a0 b0 b1 b2 b3 a1 a2 a3) of k-sub s of the stage holding step kt
  auto rd = [&](int kt, int s, int slot, int x) {
    const char* base = smem + (kt & (NSTAGE - 1)) * STAGE + fo[s];
    if (x == 0) fa[slot][0] = *(const bf16x8*)(base + abase);
    else if (x < 5) fb[slot][x - 1] = *(const bf16x8*)(base + bbase + (x - 1) * 2048);
    else fa[slot][x - 4] = *(const bf16x8*)(base + abase + (x - 4) * 2048);
  };
  // one phase: 16 MFMAs on register slot `slot`, with 8 fragment reads and 4 DMA pieces woven in
  auto phase = [&](int slot, int rkt, int rs, int dkt, int dx0) {
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      acc[q >> 2][q & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[slot][q & 3], fa[slot][q >> 2], acc[q >> 2][q & 3], 0, 0, 0);
      SB();
#ifndef ABL_NOREAD
      if ((q & 3) != 3 && q < 10) { rd(rkt, rs, slot ^ 1, q - (q >> 2)); SB(); }   // q = 0,1,2,4,5,6,8,9 -> fragment 0..7
#endif
#ifndef ABL_NODMA
      if ((q & 3) == 3) { dma(dkt, dx0 + (q >> 2)); SB(); }
#endif
    }
  };

#pragma unroll
  for (int st = 0; st < 3; ++st)
#pragma unroll
    for (int x = 0; x < 8; ++x) dma(st, x);
#pragma unroll
  for (int x = 0; x < 4; ++x) dma(3, x);
  __builtin_amdgcn_s_waitcnt(0x4F74);  // vmcnt(20): step 0 has landed
  __syncthreads();
#pragma unroll
  for (int x = 0; x < 8; ++x) rd(0, 0, 0, x);

  for (int kt = 0; kt < nk; ++kt) {
    phase(0, kt, 1, kt + 3, 4);
#ifndef ABL_NOBAR
    __builtin_amdgcn_s_waitcnt(0x4070);  // vmcnt(16) lgkmcnt(0): step kt+1 has landed, my reads of step kt are done
    __builtin_amdgcn_s_barrier();
#endif
    SB();
    phase(1, kt + 1, 0, kt + 4, 0);
  }
  if (dbg && blockIdx.x == 0 && tid == 0) { dbg[0] = clock64() - c0; dbg[1] = wall_clock64() - w0; }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int m = m0 + wrow * 128 + i * 32 + (lane & 31);
        const int n = n0 + wcol * 128 + j * 32 + 8 * q + 4 * (lane >> 5);
        if (m < M && n < N) {
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * q + e];
          *(bf16x4*)(C + (size_t)m * N + n) = __builtin_convertvector(v, bf16x4);
        }
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
  const int M = argc > 1 ? atoi(argv[1]) : 8192, N = argc > 2 ? atoi(argv[2]) : 8192, K = argc > 3 ? atoi(argv[3]) : 8192;
  std::vector<uint16_t> ha((size_t)M * K), hb((size_t)N * K), hc((size_t)M * N);
  uint32_t st = 12345;
  auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xffff) / 65536.0f - 0.5f; };
  for (auto& v : ha) v = f2bf_host(rnd());
  for (auto& v : hb) v = f2bf_host(rnd());
  bf16 *dA, *dB, *dC;
  hipMalloc(&dA, ha.size() * 2); hipMalloc(&dB, hb.size() * 2); hipMalloc(&dC, hc.size() * 2);
  hipMemcpy(dA, ha.data(), ha.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(dB, hb.data(), hb.size() * 2, hipMemcpyHostToDevice);
  long long* dDbg; hipMalloc(&dDbg, 16);
  (void)hipFuncSetAttribute((const void*)g256, hipFuncAttributeMaxDynamicSharedMemorySize, NSTAGE * STAGE);
  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
  auto launch = [&]() { hipLaunchKernelGGL(g256, dim3(tiles_m * tiles_n), dim3(256), NSTAGE * STAGE, 0, dA, dB, dC, M, N, K, tiles_n, dDbg); };
  launch();
  hipError_t e = hipDeviceSynchronize();
  if (e != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(e)); return 1; }
  hipMemcpy(hc.data(), dC, hc.size() * 2, hipMemcpyDeviceToHost);
  double maxerr = 0;
  for (int t = 0; t < 4000; ++t) {
    st = st * 1664525u + 1013904223u; const int m = (st >> 4) % M;
    st = st * 1664525u + 1013904223u; const int n = (st >> 4) % N;
    double ref = 0;
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
  long long hd[2]; hipMemcpy(hd, dDbg, 16, hipMemcpyDeviceToHost);
  printf("main loop of wg0: %lld shader clocks, %lld wall ticks (100 MHz) -> %.0f MHz, %.0f clocks per K step\n", hd[0], hd[1], hd[0] / (hd[1] / 100.0), (double)hd[0] / (K / 64) /* per 64 of K */);
  printf("variant %d M=%d N=%d K=%d: %.1f us  %.1f TFLOP/s  max rel err %.3g\n", VARIANT, M, N, K, t * 1e6, 2.0 * M * N * K / t / 1e12, maxerr);
  return 0;
}
