// Calibration probe: pure v_mfma_f32_32x32x16_bf16 stream, NACC independent accumulators per wave.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
template <int NACC>
__global__ __launch_bounds__(512) void k(float* out, int iters, float seed) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(seed + threadIdx.x * 0.001f + e); b[e] = (__bf16)(seed * 0.5f - e * 0.25f + (threadIdx.x & 7)); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
void run(int threads, int blocks_per_cu) {
  float* out; hipMalloc(&out, 256 * 8 * 512 * 4);
  int iters = 2000; dim3 grid(256 * blocks_per_cu), block(threads);
  hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
  hipLaunchKernelGGL(k<NACC>, grid, block, 0, 0, out, iters, 1.0f);
  hipEventRecord(s);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<NACC>, grid, block, 0, 0, out, iters, 1.0f);
  hipEventRecord(e); hipEventSynchronize(e);
  float ms; hipEventElapsedTime(&ms, s, e); ms /= 5;
  double flops = 2.0 * 32 * 32 * 16 * (double)NACC * iters * (threads / 64) * grid.x;
  printf("NACC=%d threads=%d blocks/CU=%d: %.3f ms  %.1f TF/s\n", NACC, threads, blocks_per_cu, ms, flops / ms / 1e9);
  hipFree(out);
}
int main() {
  run<4>(256, 1); run<8>(256, 1); run<8>(512, 1); run<8>(256, 2); run<4>(256, 4); run<8>(1024, 1);
  return 0;
}
