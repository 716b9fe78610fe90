// g256_probe.hip -- main-loop probe: 256x256x64 NT bf16 GEMM, 4 waves (2x2, each 128x128), one workgroup
// per CU, 2 x 64 KiB LDS stages filled by LDS-DMA, one barrier per K step placed before the last k-sub
// of the step (so the next step's first fragments and DMA issue hide under 16 MFMAs).
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

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int OPER = 256 * 64 * 2;   // 32 KiB per operand tile
constexpr int STAGE = 2 * OPER;      // 64 KiB

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}

__global__ __launch_bounds__(256, 1) void g256(const bf16* __restrict__ A, const bf16* __restrict__ B, bf16* __restrict__ C,
                                               int M, int N, int K, int tiles_n, long long* dbg) {
  const long long c0 = clock64(), w0 = wall_clock64();
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wrow = wave >> 1, wcol = wave & 1;
  const int tile_m = blockIdx.x / tiles_n, tile_n = blockIdx.x % tiles_n;
#ifdef FAKE_L2   // every workgroup streams the same 2 x 1 MiB panel window: pure L2 hits (results are wrong on purpose)
  const int m0 = 0, n0 = 0;
#else
  const int m0 = tile_m * BM, n0 = tile_n * BN;
#endif
  const int nk = K / BK;

  const __amdgpu_buffer_rsrc_t ra = make_rsrc(A, (uint32_t)((size_t)M * K * 2));
  const __amdgpu_buffer_rsrc_t rb = make_rsrc(B, (uint32_t)((size_t)N * K * 2));

  // DMA source offsets: wave w, piece j (0..7) of each operand = rows (w*8+j)*8 + lane/8, swizzled 16-byte chunk
  uint32_t offa[8], offb[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int row = (wave * 8 + j) * 8 + (lane >> 3);
    const int kc = (lane & 7) ^ ((row >> 1) & 7);
    offa[j] = (m0 + row) < M ? (uint32_t)(((size_t)(m0 + row) * K + kc * 8) * 2) : 0x80000000u;
    offb[j] = (n0 + row) < N ? (uint32_t)(((size_t)(n0 + row) * K + kc * 8) * 2) : 0x80000000u;
  }
  auto issue = [&](int kt, int stage) {
    char* pa = smem + stage * STAGE + wave * 8192;
    char* pb = pa + OPER;
#ifdef FAKE_L2
    const uint32_t kb = kt < nk ? (uint32_t)((kt & 31) * BK * 2) : 0x80000000u;
#else
    const uint32_t kb = kt < nk ? (uint32_t)(kt * BK * 2) : 0x80000000u;   // past the end: out-of-range -> zeros
#endif
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void*)(pa + j * 1024), 16, offa[j] + kb, 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_void*)(pb + j * 1024), 16, offb[j] + kb, 0, 0, 0);
    }
  };

  // fragment addresses: row = sub*32 + (lane&31), chunk = s*2 + (lane>>5), swizzle key ((row>>1)&7) is lane-only
  const int sw = ((lane & 31) >> 1) & 7;
  int fo[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) fo[s] = (lane & 31) * 128 + (((s * 2 + (lane >> 5)) ^ sw) << 4);
  const int abase = wrow * 128 * 128, bbase = OPER + wcol * 128 * 128;

  f32x16 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  bf16x8 fa[2][4], fb[2][4];
  auto readfrags = [&](int stage, int s, int slot) {
    const char* base = smem + stage * STAGE + fo[s];
#ifdef ABL_NOREAD
    if (M != 12345) return;
#endif
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      fa[slot][t] = *(const bf16x8*)(base + abase + t * 4096);
      fb[slot][t] = *(const bf16x8*)(base + bbase + t * 4096);
    }
  };
  auto mfmas = [&](int slot) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[slot][j], fa[slot][i], acc[i][j], 0, 0, 0);
  };

  issue(0, 0);
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
  __syncthreads();
  issue(1, 1);
  readfrags(0, 0, 0);

  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      readfrags(cur, s + 1, (s + 1) & 1);
#if VARIANT == 1
      // interleave: 1 MFMA : 1 DS read for the first 8, then the remaining MFMAs
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
      mfmas(s & 1);
      __builtin_amdgcn_sched_barrier(0);
#else
      __builtin_amdgcn_sched_barrier(0);
      mfmas(s & 1);
      __builtin_amdgcn_sched_barrier(0);
#endif
    }
    // last k-sub of this step: everybody is done reading stage `cur` (its fragments are in registers) and the
    // DMA of the next step has landed -> barrier, then refill `cur` and fetch the next step's first fragments
#ifndef ABL_NOBAR
    __builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0) lgkmcnt(0)
    __builtin_amdgcn_s_barrier();
#endif
    readfrags(cur ^ 1, 0, 0);
#ifndef ABL_NODMA
    issue(kt + 2, cur);
#endif
#if VARIANT == 1
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    }
    mfmas(1);
    __builtin_amdgcn_sched_barrier(0);
#else
    __builtin_amdgcn_sched_barrier(0);
    mfmas(1);
    __builtin_amdgcn_sched_barrier(0);
#endif
  }

  if (dbg && blockIdx.x == 0 && tid == 0) { dbg[0] = clock64() - c0; dbg[1] = wall_clock64() - w0; }
  // plain epilogue (8-byte stores; not the point of this probe)
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
  (void)hipFuncSetAttribute((const void*)g256, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE);
  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
  auto launch = [&]() { hipLaunchKernelGGL(g256, dim3(tiles_m * tiles_n), dim3(256), 2 * STAGE, 0, dA, dB, dC, M, N, K, tiles_n, dDbg); };
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
  printf("main loop of wg0: %lld shader clocks, %lld wall ticks (100 MHz) -> %.0f MHz, %.0f clocks per K step\n", hd[0], hd[1], hd[0] / (hd[1] / 100.0), (double)hd[0] / (K / 64));
  printf("variant %d M=%d N=%d K=%d: %.1f us  %.1f TFLOP/s  max rel err %.3g\n", VARIANT, M, N, K, t * 1e6, 2.0 * M * N * K / t / 1e12, maxerr);
  return 0;
}
