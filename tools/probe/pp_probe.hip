// Ping-pong probe: builds the 256x256 kernel's phase structure up step by step on random data.
//  step 0: alternate groups, compute-only          step 1: + 12 ds_read_b128 per load phase
//  step 2: + 4 LDS-DMA pieces per compute phase    step 3: + counted vmcnt waits
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(3))) void lds_void;
__device__ __forceinline__ int lds_addr_kc32(int row, int kc) { return row * 64 + ((kc ^ ((row >> 2) & 3)) << 4); }
template <int STEP>
__global__ __launch_bounds__(512) void k(const __bf16* __restrict__ src, float* out, int iters, unsigned nbytes) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), grp = wave >> 2, wi = wave & 3;
  for (int i = tid; i < 32768; i += 512) ((float*)smem)[i] = (float)((i * 2654435761u) >> 20) * 1e-3f - 2.0f;
  __syncthreads();
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
  f32x16 acc[4][2];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  bf16x8 fa[2][4], fb[2][2];
  auto load_frags = [&](int u) {
    const char* pa = smem + (u & 3) * 32768;
    const char* pb = pa + 16384;
#pragma unroll
    for (int ss = 0; ss < 2; ++ss) {
      const int chunk = ss * 2 + (lane >> 5);
#pragma unroll
      for (int t = 0; t < 4; ++t) fa[ss][t] = *(const bf16x8*)(pa + lds_addr_kc32(grp * 128 + t * 32 + (lane & 31), chunk));
#pragma unroll
      for (int t = 0; t < 2; ++t) fb[ss][t] = *(const bf16x8*)(pb + lds_addr_kc32(wi * 64 + t * 32 + (lane & 31), chunk));
    }
  };
  load_frags(0);
  const unsigned lane_off = ((blockIdx.x & 3) * 4096u + wi * 1024u + lane * 16u);   // 4 distinct streams: mostly L2 hits, like real tiles
  for (int u = 0; u < iters; ++u) {
    // load phase of this group (other group computes)
    if (STEP >= 1) load_frags(u);
    if (STEP >= 3 && grp == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (grp == 1 && u == 0) asm volatile("s_barrier" ::: "memory");   // stagger group 1 by one phase
    // compute phase
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ss = 0; ss < 2; ++ss)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[ss][j], fa[ss][i], acc[i][j], 0, 0, 0);
        if (STEP >= 2 && (i & 1) == 0) {
          __builtin_amdgcn_sched_barrier(0);
          char* dst = smem + ((u + 3) & 3) * 32768 + (ss ? 0 : 16384) + grp * 8192 + wi * 2048 + (i >> 1) * 1024;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)dst, 16, (lane_off + (unsigned)(u * 64 + ss * 32 + i * 8) * 16384u) % (nbytes - 16), 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    __builtin_amdgcn_s_setprio(0);
    if (STEP >= 3 && grp == 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
  }
  if (grp == 0) asm volatile("s_barrier" ::: "memory");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) s += acc[i][j][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
  const size_t n = 256u << 20;   // 256 MiB source
  unsigned short* h = (unsigned short*)malloc(n);
  for (size_t i = 0; i < n / 2; ++i) { float f = (rand() / (float)RAND_MAX) - 0.5f; unsigned u; memcpy(&u, &f, 4); h[i] = (unsigned short)(u >> 16); }
  __bf16* d; float* out; hipMalloc(&d, n); hipMalloc(&out, 256 * 512 * 4);
  hipMemcpy(d, h, n, hipMemcpyHostToDevice);
  int iters = 4000; dim3 grid(256), block(512);
  hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
  for (int st = 0; st < 4; ++st) {
    auto launch = [&]() {
      if (st == 0) hipLaunchKernelGGL(k<0>, grid, block, 131072, 0, d, out, iters, (unsigned)n);
      if (st == 1) hipLaunchKernelGGL(k<1>, grid, block, 131072, 0, d, out, iters, (unsigned)n);
      if (st == 2) hipLaunchKernelGGL(k<2>, grid, block, 131072, 0, d, out, iters, (unsigned)n);
      if (st == 3) hipLaunchKernelGGL(k<3>, grid, block, 131072, 0, d, out, iters, (unsigned)n);
    };
    if (st == 0) { hipFuncSetAttribute((const void*)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072); hipFuncSetAttribute((const void*)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
                   hipFuncSetAttribute((const void*)k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072); hipFuncSetAttribute((const void*)k<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072); }
    launch();
    hipEventRecord(s);
    for (int r = 0; r < 3; ++r) launch();
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e); ms /= 3;
    double flops = 2.0 * 32 * 32 * 16 * 16.0 * iters * 8 * 256;
    printf("step %d: %.3f ms  %.1f TF/s  (err=%s)\n", st, ms, flops / ms / 1e9, hipGetErrorString(hipGetLastError()));
  }
  return 0;
}
