// Calibration probe 2: the ping-pong kernel's register-level MFMA pattern (acc[4][2], fa[2][4], fb[2][2]) on
// constant vs random operand data, 2 waves per SIMD, no memory traffic in the loop.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
template <int MODE>
__global__ __launch_bounds__(512) void k(const bf16x8* __restrict__ src, float* out, int iters) {
  const int grp = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8);
  f32x16 acc[4][2];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  bf16x8 fa[2][4], fb[2][2];
  for (int s = 0; s < 2; ++s) {
    for (int t = 0; t < 4; ++t) fa[s][t] = src[(s * 6 + t) * 512 + threadIdx.x];
    for (int t = 0; t < 2; ++t) fb[s][t] = src[(s * 6 + 4 + t) * 512 + threadIdx.x];
  }
  for (int it = 0; it < iters; ++it) {
    if (MODE == 3) { asm volatile("s_barrier" ::: "memory"); if ((it & 1) != grp) continue; }
#pragma unroll
    for (int ss = 0; ss < 2; ++ss)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[ss][j], fa[ss][i], acc[i][j], 0, 0, 0);
    if (MODE == 1) asm volatile("s_barrier" ::: "memory");                       // barrier every 16 MFMAs, all waves computing
    if (MODE == 2) { asm volatile("s_barrier" ::: "memory"); asm volatile("s_barrier" ::: "memory"); }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) s += acc[i][j][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
  const int n = 12 * 512 * 8;
  unsigned short* h = (unsigned short*)malloc(n * 2);
  bf16x8* d; float* out; hipMalloc(&d, n * 2); hipMalloc(&out, 256 * 512 * 4);
  for (int mode = 1; mode < 2; ++mode) {
    for (int i = 0; i < n; ++i) {
      float f = mode ? ((rand() / (float)RAND_MAX) - 0.5f) : 1.0f;
      unsigned u; memcpy(&u, &f, 4); h[i] = (unsigned short)(u >> 16);
    }
    hipMemcpy(d, h, n * 2, hipMemcpyHostToDevice);
    int iters = 4000; dim3 grid(256), block(512);
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    for (int km = 0; km < 4; ++km) {
      auto launch = [&]() {
        if (km == 0) hipLaunchKernelGGL(k<0>, grid, block, 0, 0, d, out, iters);
        if (km == 1) hipLaunchKernelGGL(k<1>, grid, block, 0, 0, d, out, iters);
        if (km == 2) hipLaunchKernelGGL(k<2>, grid, block, 0, 0, d, out, iters);
        if (km == 3) hipLaunchKernelGGL(k<3>, grid, block, 0, 0, d, out, iters);
      };
      launch();
      hipEventRecord(s);
      for (int r = 0; r < 5; ++r) launch();
      hipEventRecord(e); hipEventSynchronize(e);
      float ms; hipEventElapsedTime(&ms, s, e); ms /= 5;
      double flops = 2.0 * 32 * 32 * 16 * 16.0 * iters * 8 * 256 * (km == 3 ? 0.5 : 1.0);
      printf("mode %d (0 free-running, 1 barrier/16 MFMA, 2 two barriers, 3 ping-pong alternate): %.3f ms  %.1f TF/s\n", km, ms, flops / ms / 1e9);
    }
  }
  return 0;
}
