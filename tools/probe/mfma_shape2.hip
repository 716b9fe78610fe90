// Calibration probe 4: the two bf16 MFMA shapes at TWO waves per SIMD (128x64 per-wave tile, 128 accumulator
// registers), random operands, no memory traffic: sustained rate and the shader clock each settles at.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int SHAPE>
__global__ __launch_bounds__(256, 2) void k(const bf16x8* __restrict__ src, float* out, int iters, long long* dbg) {
  const long long c0 = clock64(), w0 = wall_clock64();
  float s = 0.f;
  const int t = threadIdx.x;
  if constexpr (SHAPE == 32) {
    f32x16 acc[4][2];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    bf16x8 fa[2][4], fb[2][2];
    for (int ss = 0; ss < 2; ++ss) { for (int q = 0; q < 4; ++q) fa[ss][q] = src[(ss * 6 + q) * 256 + t]; for (int q = 0; q < 2; ++q) fb[ss][q] = src[(ss * 6 + 4 + q) * 256 + t]; }
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int ss = 0; ss < 2; ++ss)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[ss][j], fa[ss][i], acc[i][j], 0, 0, 0);
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) s += acc[i][j][e];
  } else {
    f32x4 acc[8][4];
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;
    bf16x8 fa[8], fb[4];
    for (int q = 0; q < 8; ++q) fa[q] = src[q * 256 + t];
    for (int q = 0; q < 4; ++q) fb[q] = src[(8 + q) * 256 + t];
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) for (int e = 0; e < 4; ++e) s += acc[i][j][e];
  }
  if (blockIdx.x == 0 && t == 0) { dbg[0] = clock64() - c0; dbg[1] = wall_clock64() - w0; }
  out[blockIdx.x * blockDim.x + t] = s;
}
int main() {
  const int n = 12 * 256 * 8;
  unsigned short* h = (unsigned short*)malloc(n * 2);
  bf16x8* d; float* out; long long* dbg;
  (void)hipMalloc(&d, n * 2); (void)hipMalloc(&out, 512 * 256 * 4); (void)hipMalloc(&dbg, 16);
  for (int i = 0; i < n; ++i) { float f = (rand() / (float)RAND_MAX) - 0.5f; unsigned u; memcpy(&u, &f, 4); h[i] = (unsigned short)(u >> 16); }
  (void)hipMemcpy(d, h, n * 2, hipMemcpyHostToDevice);
  const int iters = 8000;
  hipEvent_t s, e; (void)hipEventCreate(&s); (void)hipEventCreate(&e);
  for (int shape = 0; shape < 2; ++shape) {
    auto launch = [&]() {
      if (shape == 0) hipLaunchKernelGGL(k<32>, dim3(512), dim3(256), 0, 0, d, out, iters, dbg);
      else hipLaunchKernelGGL(k<16>, dim3(512), dim3(256), 0, 0, d, out, iters, dbg);
    };
    for (int r = 0; r < 3; ++r) launch();
    (void)hipEventRecord(s);
    for (int r = 0; r < 10; ++r) launch();
    (void)hipEventRecord(e); (void)hipEventSynchronize(e);
    float ms; (void)hipEventElapsedTime(&ms, s, e); ms /= 10;
    long long hd[2]; (void)hipMemcpy(hd, dbg, 16, hipMemcpyDeviceToHost);
    const double flops = 2.0 * 128 * 64 * 32 * (double)iters * 8 * 256;
    printf("%s, 2 waves/SIMD: %.3f ms  %.1f TFLOP/s  shader clock %.0f MHz  (%.1f clocks per 128x64x32 wave step; 512 = this wave's share of a full pipe... both waves: 1024 per pair)\n",
           shape == 0 ? "v_mfma_f32_32x32x16_bf16" : "v_mfma_f32_16x16x32_bf16", ms, flops / ms / 1e9, hd[0] / (hd[1] / 100.0), (double)hd[0] / iters);
  }
  return 0;
}
