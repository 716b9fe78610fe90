// Calibration probe 3: sustained random-data rate of the two bf16 MFMA shapes on a 128x128 per-wave tile
// (1 wave per SIMD, 4 waves per CU, no memory traffic in the loop), with the shader clock each one settles at.
//   32x32x16: acc[4][4] f32x16, 4+4 fragments per k16          16x16x32: acc[8][8] f32x4, 8+8 fragments per k32
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int SHAPE>
__global__ __launch_bounds__(256, 1) void k(const bf16x8* __restrict__ src, float* out, int iters, long long* dbg) {
  const long long c0 = clock64(), w0 = wall_clock64();
  float s = 0.f;
  if constexpr (SHAPE == 32) {
    f32x16 acc[4][4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    bf16x8 fa[2][4], fb[2][4];
    for (int ss = 0; ss < 2; ++ss) for (int t = 0; t < 4; ++t) { fa[ss][t] = src[(ss * 8 + t) * 256 + threadIdx.x]; fb[ss][t] = src[(ss * 8 + 4 + t) * 256 + threadIdx.x]; }
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int ss = 0; ss < 2; ++ss)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[ss][j], fa[ss][i], acc[i][j], 0, 0, 0);
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) s += acc[i][j][e];
  } else {
    f32x4 acc[8][8];
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j) for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;
    bf16x8 fa[8], fb[8];
    for (int t = 0; t < 8; ++t) { fa[t] = src[t * 256 + threadIdx.x]; fb[t] = src[(8 + t) * 256 + threadIdx.x]; }
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j) for (int e = 0; e < 4; ++e) s += acc[i][j][e];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) { dbg[0] = clock64() - c0; dbg[1] = wall_clock64() - w0; }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
  const int n = 16 * 256 * 8;
  unsigned short* h = (unsigned short*)malloc(n * 2);
  bf16x8* d; float* out; long long* dbg;
  (void)hipMalloc(&d, n * 2); (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&dbg, 16);
  for (int i = 0; i < n; ++i) { float f = (rand() / (float)RAND_MAX) - 0.5f; unsigned u; memcpy(&u, &f, 4); h[i] = (unsigned short)(u >> 16); }
  (void)hipMemcpy(d, h, n * 2, hipMemcpyHostToDevice);
  const int iters = 8000;
  hipEvent_t s, e; (void)hipEventCreate(&s); (void)hipEventCreate(&e);
  for (int shape = 0; shape < 2; ++shape) {
    auto launch = [&]() {
      if (shape == 0) hipLaunchKernelGGL(k<32>, dim3(256), dim3(256), 0, 0, d, out, iters, dbg);
      else hipLaunchKernelGGL(k<16>, dim3(256), dim3(256), 0, 0, d, out, iters, dbg);
    };
    for (int r = 0; r < 3; ++r) launch();
    (void)hipEventRecord(s);
    for (int r = 0; r < 10; ++r) launch();
    (void)hipEventRecord(e); (void)hipEventSynchronize(e);
    float ms; (void)hipEventElapsedTime(&ms, s, e); ms /= 10;
    long long hd[2]; (void)hipMemcpy(hd, dbg, 16, hipMemcpyDeviceToHost);
    const double flops = 2.0 * 128 * 128 * 32 * (double)iters * 4 * 256;
    printf("%s: %.3f ms  %.1f TFLOP/s  shader clock %.0f MHz  (%.1f clocks per 128x128x32 wave step; 1024 = MFMA pipe full)\n",
           shape == 0 ? "v_mfma_f32_32x32x16_bf16" : "v_mfma_f32_16x16x32_bf16", ms, flops / ms / 1e9, hd[0] / (hd[1] / 100.0), (double)hd[0] / iters);
  }
  return 0;
}
