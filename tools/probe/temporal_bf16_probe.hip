// temporal_bf16_probe.hip -- would the temporal attention (csrc/attention.hip: temporal_attn_kernel, one wave per 8- or 16-frame
// problem, VALU + LDS) gain from keeping its rows in LDS as bf16 instead of fp32?
//
// The production kernel widens q / k / v (/ dO) to fp32 when it parks them in LDS (pitch hd + 4 floats): 10 / 13 KiB per wave at 8
// frames, 20 / 28 KiB at 16 -- LDS is what limits its occupancy (16 / 12 waves per CU at 8 frames, 7 / 5 at 16).  The rows ARE bf16
// (q * scale is rounded to bf16 by the reference itself), so bf16 rows at the conflict-free 208-byte pitch hold the same values in half
// the bytes; the two dot-product phases (S = q k^T, dP = dO v^T) then run on v_dot2c_f32_bf16 (two MACs per instruction, fp32
// accumulate, no widening), the three p-weighted row sums widen their operand on the fly.
// This probe is that kernel, stand-alone, checked against and timed beside the production entry points of libmpv_hip.so.
// Not a product path.
// Build (repo root): hipcc --offload-arch=gfx950 -O3 -o tools/probe/temporal_bf16_probe tools/probe/temporal_bf16_probe.hip \
//                          -Lyouku-mplug_amd -lmpv_hip -Wl,-rpath,'$ORIGIN/../../youku-mplug_amd'
// Run (GPU box):     tools/probe/temporal_bf16_probe [B] [T]        (T = 8 or 16; B clips)
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

extern "C" int mpv_temporal_attn_fwd(const void* qkv, void* out, int n_outer, int64_t outer_stride, int n_inner, int64_t inner_offset,
                                     int64_t t_stride, int T, int heads, int head_dim, float scale, hipStream_t stream);
extern "C" int mpv_temporal_attn_bwd(const void* qkv, const void* dout, void* dqkv, int n_outer, int64_t outer_stride, int n_inner,
                                     int64_t inner_offset, int64_t t_stride, int T, int heads, int head_dim, float scale, hipStream_t stream);

__device__ __forceinline__ f32x4 cvt4(bf16x4 v) { return __builtin_convertvector(v, f32x4); }
__device__ __forceinline__ bf16x4 cvt4(f32x4 v) { return __builtin_convertvector(v, bf16x4); }
__device__ __forceinline__ float bf2f(bf16 v) { return (float)v; }
__device__ __forceinline__ bf16 f2bf(float v) { return (bf16)v; }
#define WAVE_SYNC() __builtin_amdgcn_wave_barrier()

struct TempArgs {
  const bf16* qkv;
  const bf16* dout;
  bf16* out;
  bf16* dqkv;
  int n_outer, n_inner;
  long long outer_stride, inner_offset, t_stride;
  int heads;
  float scale;
};

// 8 bf16 of a against 8 bf16 of b, accumulated into two fp32 chains (v_dot2c_f32_bf16)
__device__ __forceinline__ void dot8(bf16x8 a, bf16x8 b, float& c0, float& c1) {
  c0 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(a, a, 0, 1), __builtin_shufflevector(b, b, 0, 1), c0, false);
  c1 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(a, a, 2, 3), __builtin_shufflevector(b, b, 2, 3), c1, false);
  c0 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(a, a, 4, 5), __builtin_shufflevector(b, b, 4, 5), c0, false);
  c1 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(a, a, 6, 7), __builtin_shufflevector(b, b, 6, 7), c1, false);
}

template <bool BWD, int T, int HD>
__global__ __launch_bounds__(256) void temporal_bf16_kernel(const TempArgs p) {
  extern __shared__ __attribute__((aligned(16))) char tsm_raw[];
  constexpr int LDB = HD + 8;                          // bf16 row pitch: 208 bytes at head_dim 96 (rows of a 16-byte column land on distinct banks)
  constexpr int H4 = HD / 4, H8 = HD / 8;
  constexpr int ROWS = (BWD ? 4 : 3) * T * LDB * 2;    // bytes of the bf16 images
  constexpr int PER_WAVE = (ROWS + (BWD ? 2 : 1) * T * (T + 1) * 4 + 15) & ~15;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwv = (int)(blockDim.x >> 6);
  const int D = p.heads * HD;
  char* base = tsm_raw + wave * PER_WAVE;
  bf16* qs = (bf16*)base;
  bf16* ks = qs + T * LDB;
  bf16* vs = ks + T * LDB;
  bf16* dos = vs + T * LDB;                            // BWD only
  float* ps = (float*)(base + ROWS);                   // [T][T+1] probabilities
  float* dss = ps + T * (T + 1);                       // BWD only: dS
  const long long nprob = (long long)p.n_outer * p.n_inner * p.heads;
  const long long pstride = (long long)gridDim.x * nwv;
  constexpr int NX = (T * H4 + 63) / 64;
  bf16x4 rq[NX], rk[NX], rv[NX], rd[BWD ? NX : 1];
  auto row_of = [&](long long pr, int& h) {
    h = (int)(pr % p.heads);
    const long long seq = pr / p.heads;
    const long long o = seq / p.n_inner, i = seq % p.n_inner;
    return o * p.outer_stride + p.inner_offset + i;
  };
  auto request = [&](long long pr) {
    int h;
    const long long row0 = row_of(pr, h);
#pragma unroll
    for (int j = 0; j < NX; ++j) {
      const int x = lane + 64 * j;
      if (x < T * H4) {
        const int t = x / H4, c4 = x - t * H4;
        const bf16* src = p.qkv + (row0 + t * p.t_stride) * (3LL * D) + h * HD + c4 * 4;
        rq[j] = *(const bf16x4*)src;
        rk[j] = *(const bf16x4*)(src + D);
        rv[j] = *(const bf16x4*)(src + 2 * D);
        if constexpr (BWD) rd[j] = *(const bf16x4*)(p.dout + (row0 + t * p.t_stride) * (long long)D + h * HD + c4 * 4);
      }
    }
  };
  long long pr0 = (long long)blockIdx.x * nwv + wave;
  if (pr0 < nprob) request(pr0);
  for (long long pr = pr0; pr < nprob; pr += pstride) {
    int h;
    const long long row0 = row_of(pr, h);
#pragma unroll
    for (int j = 0; j < NX; ++j) {
      const int x = lane + 64 * j;
      if (x < T * H4) {
        const int t = x / H4, c4 = x - t * H4;
        *(bf16x4*)(qs + t * LDB + c4 * 4) = cvt4(cvt4(rq[j]) * p.scale);      // q * scale rounds to bf16 (reference :179)
        *(bf16x4*)(ks + t * LDB + c4 * 4) = rk[j];
        *(bf16x4*)(vs + t * LDB + c4 * 4) = rv[j];
        if constexpr (BWD) *(bf16x4*)(dos + t * LDB + c4 * 4) = rd[j];
      }
    }
    if (pr + pstride < nprob) request(pr + pstride);
    WAVE_SYNC();
#pragma unroll
    for (int x = lane; x < T * T; x += 64) {
      const int a = x / T, bb = x - a * T;
      float s0 = 0.f, s1 = 0.f, d0 = 0.f, d1 = 0.f;
#pragma unroll
      for (int c8 = 0; c8 < H8; ++c8) {
        dot8(*(const bf16x8*)(qs + a * LDB + c8 * 8), *(const bf16x8*)(ks + bb * LDB + c8 * 8), s0, s1);
        if constexpr (BWD) dot8(*(const bf16x8*)(dos + a * LDB + c8 * 8), *(const bf16x8*)(vs + bb * LDB + c8 * 8), d0, d1);
      }
      ps[a * (T + 1) + bb] = s0 + s1;
      if constexpr (BWD) dss[a * (T + 1) + bb] = d0 + d1;                  // dP = dO V^T
    }
    WAVE_SYNC();
    if (lane < T) {
      float mx = -INFINITY;
#pragma unroll
      for (int j = 0; j < T; ++j) mx = fmaxf(mx, ps[lane * (T + 1) + j]);
      float sum = 0.f;
#pragma unroll
      for (int j = 0; j < T; ++j) sum += __expf(ps[lane * (T + 1) + j] - mx);
      const float inv = 1.0f / sum;
      float dl = 0.f;
#pragma unroll
      for (int j = 0; j < T; ++j) {
        const float pv = __expf(ps[lane * (T + 1) + j] - mx) * inv;
        ps[lane * (T + 1) + j] = BWD ? pv : bf2f(f2bf(pv));              // forward: probabilities cast to bf16 (:201)
        if constexpr (BWD) dl += pv * dss[lane * (T + 1) + j];
      }
      if constexpr (BWD)
#pragma unroll
        for (int j = 0; j < T; ++j) dss[lane * (T + 1) + j] = ps[lane * (T + 1) + j] * (dss[lane * (T + 1) + j] - dl);
    }
    WAVE_SYNC();
#pragma unroll
    for (int x = lane; x < T * H4; x += 64) {
      const int a = x / H4, c4 = x - a * H4;
      if constexpr (!BWD) {
        f32x4 o4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < T; ++j) o4 += cvt4(*(const bf16x4*)(vs + j * LDB + c4 * 4)) * ps[a * (T + 1) + j];
        *(bf16x4*)(p.out + (row0 + a * p.t_stride) * (long long)D + h * HD + c4 * 4) = cvt4(o4);
      } else {
        f32x4 dq = {0.f, 0.f, 0.f, 0.f}, dk = dq, dv = dq;
#pragma unroll
        for (int j = 0; j < T; ++j) {
          dq += cvt4(*(const bf16x4*)(ks + j * LDB + c4 * 4)) * dss[a * (T + 1) + j];
          dk += cvt4(*(const bf16x4*)(qs + j * LDB + c4 * 4)) * dss[j * (T + 1) + a];
          dv += cvt4(*(const bf16x4*)(dos + j * LDB + c4 * 4)) * ps[j * (T + 1) + a];
        }
        bf16* dst = p.dqkv + (row0 + a * p.t_stride) * (3LL * D) + h * HD + c4 * 4;
        *(bf16x4*)dst = cvt4(dq * p.scale);
        *(bf16x4*)(dst + D) = cvt4(dk);
        *(bf16x4*)(dst + 2 * D) = cvt4(dv);
      }
    }
    WAVE_SYNC();
  }
}

template <bool BWD, int T, int HD>
static void launch(const TempArgs& t, int nwv_req, hipStream_t s, int* nwv_out, size_t* lds_out) {
  constexpr int LDB = HD + 8;
  const size_t per_wave = (((size_t)(BWD ? 4 : 3) * T * LDB * 2 + (BWD ? 2 : 1) * T * (T + 1) * 4) + 15) & ~(size_t)15;
  int nwv = nwv_req;
  if (nwv <= 0) {          // the most waves a CU's 160 KiB of LDS holds
    const size_t cu = 160 * 1024;
    int best = 4, best_w = (int)(cu / (4 * per_wave)) * 4;
    for (int n = 3; n >= 1; --n) {
      const int w = (int)(cu / (n * per_wave)) * n;
      if (w > best_w) best = n, best_w = w;
    }
    nwv = best;
  }
  const long long nprob = (long long)t.n_outer * t.n_inner * t.heads;
  const int grid = (int)((nprob + nwv - 1) / nwv < 16384 ? (nprob + nwv - 1) / nwv : 16384);
  (void)hipFuncSetAttribute((const void*)temporal_bf16_kernel<BWD, T, HD>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
  hipLaunchKernelGGL((temporal_bf16_kernel<BWD, T, HD>), dim3(grid), dim3(64 * nwv), nwv * per_wave, s, t);
  if (nwv_out) *nwv_out = nwv;
  if (lds_out) *lds_out = per_wave;
}

static float bf(uint16_t v) {
  uint32_t u = (uint32_t)v << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

template <int T>
static int run(int B) {
  constexpr int HD = 96;
  const int N = 196, N1 = N + 1, heads = 8, D = heads * HD;
  const long long rows = (long long)B * T * N1;
  const float scale = 1.0f / sqrtf((float)HD);
  std::vector<uint16_t> hq((size_t)rows * 3 * D), hd_((size_t)rows * D);
  uint32_t st = 12345u;
  auto rnd = [&]() {
    st = st * 1664525u + 1013904223u;
    const float f = ((st >> 8) & 0xffff) / 32768.0f - 1.0f;
    uint32_t u;
    memcpy(&u, &f, 4);
    return (uint16_t)((u + 0x8000u) >> 16);
  };
  for (auto& v : hq) v = rnd();
  for (auto& v : hd_) v = rnd();
  uint16_t *qkv, *dout, *out0, *out1, *dq0, *dq1;
  hipMalloc(&qkv, hq.size() * 2); hipMalloc(&dout, hd_.size() * 2);
  hipMalloc(&out0, hd_.size() * 2); hipMalloc(&out1, hd_.size() * 2);
  hipMalloc(&dq0, hq.size() * 2); hipMalloc(&dq1, hq.size() * 2);
  hipMemcpy(qkv, hq.data(), hq.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(dout, hd_.data(), hd_.size() * 2, hipMemcpyHostToDevice);
  hipMemset(out0, 0, hd_.size() * 2); hipMemset(out1, 0, hd_.size() * 2);
  hipMemset(dq0, 0, hq.size() * 2); hipMemset(dq1, 0, hq.size() * 2);
  TempArgs t = {};
  t.qkv = (const bf16*)qkv; t.dout = (const bf16*)dout;
  t.n_outer = B; t.n_inner = N; t.outer_stride = (long long)T * N1; t.inner_offset = 1; t.t_stride = N1; t.heads = heads; t.scale = scale;
  hipStream_t s = 0;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  auto timeit = [&](auto fn) {
    float best = 1e9f;
    for (int r = 0; r < 3; ++r) {
      fn(); fn();
      hipEventRecord(e0, s);
      for (int i = 0; i < 10; ++i) fn();
      hipEventRecord(e1, s);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      best = fminf(best, ms / 10.f);
    }
    return best * 1e3f;
  };
  // production
  int rc = mpv_temporal_attn_fwd(qkv, out0, B, (int64_t)T * N1, N, 1, N1, T, heads, HD, scale, s);
  rc |= mpv_temporal_attn_bwd(qkv, dout, dq0, B, (int64_t)T * N1, N, 1, N1, T, heads, HD, scale, s);
  if (rc) { printf("production entry points failed: %d\n", rc); return 1; }
  const float pf = timeit([&] { mpv_temporal_attn_fwd(qkv, out0, B, (int64_t)T * N1, N, 1, N1, T, heads, HD, scale, s); });
  const float pb = timeit([&] { mpv_temporal_attn_bwd(qkv, dout, dq0, B, (int64_t)T * N1, N, 1, N1, T, heads, HD, scale, s); });
  const double by = (double)B * T * N * D * 2;
  printf("T=%2d B=%3d production (fp32 rows)        : fwd %8.1f us (%5.0f GB/s)  bwd %8.1f us (%5.0f GB/s)\n", T, B, pf, 4 * by / pf / 1e3, pb, 8 * by / pb / 1e3);
  for (int nwv_req : {0, 4, 2, 1}) {
    TempArgs tf = t, tb = t;
    tf.out = (bf16*)out1; tb.dqkv = (bf16*)dq1;
    int nf = 0, nb = 0;
    size_t lf = 0, lb = 0;
    launch<false, T, HD>(tf, nwv_req, s, &nf, &lf);
    launch<true, T, HD>(tb, nwv_req, s, &nb, &lb);
    if (hipDeviceSynchronize() != hipSuccess) { printf("probe launch failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
    const float qf = timeit([&] { launch<false, T, HD>(tf, nwv_req, s, nullptr, nullptr); });
    const float qb = timeit([&] { launch<true, T, HD>(tb, nwv_req, s, nullptr, nullptr); });
    // compare with the production outputs
    std::vector<uint16_t> a(hd_.size()), b(hd_.size()), ga(hq.size()), gb(hq.size());
    hipMemcpy(a.data(), out0, a.size() * 2, hipMemcpyDeviceToHost); hipMemcpy(b.data(), out1, b.size() * 2, hipMemcpyDeviceToHost);
    hipMemcpy(ga.data(), dq0, ga.size() * 2, hipMemcpyDeviceToHost); hipMemcpy(gb.data(), dq1, gb.size() * 2, hipMemcpyDeviceToHost);
    double eo = 0, mo = 0, eg = 0, mg = 0;
    for (size_t i = 0; i < a.size(); ++i) { eo = fmax(eo, fabs((double)bf(a[i]) - bf(b[i]))); mo = fmax(mo, fabs((double)bf(a[i]))); }
    for (size_t i = 0; i < ga.size(); ++i) { eg = fmax(eg, fabs((double)bf(ga[i]) - bf(gb[i]))); mg = fmax(mg, fabs((double)bf(ga[i]))); }
    printf("T=%2d B=%3d bf16 rows + dot2, %d/%d waves/wg (%4.1f/%4.1f KiB per wave): fwd %8.1f us (%5.0f GB/s)  bwd %8.1f us (%5.0f GB/s) | max |diff| / max |ref|: out %.2e  dqkv %.2e\n",
           T, B, nf, nb, lf / 1024.0, lb / 1024.0, qf, 4 * by / qf / 1e3, qb, 8 * by / qb / 1e3, eo / mo, eg / mg);
  }
  return 0;
}

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 32, T = argc > 2 ? atoi(argv[2]) : 8;
  if (T == 8) return run<8>(B);
  if (T == 16) return run<16>(B);
  if (T == 4) return run<4>(B);
  printf("T must be 4, 8 or 16\n");
  return 2;
}
