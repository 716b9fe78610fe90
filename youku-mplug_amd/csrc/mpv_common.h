// mpv_common.h -- shared device helpers for the mPLUG-Video gfx950 kernels.
// CDNA4 only (wave64, MFMA 32x32x16 bf16, ds_read_b64_tr_b16); no CUDA/compat paths.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) float f32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(2))) int i32x2;
typedef __attribute__((ext_vector_type(4))) int i32x4;

#define MPV_OK 0
#define MPV_E_SHAPE (-1)
#define MPV_E_ALIGN (-2)
#define MPV_E_ARCH (-3)
#define MPV_E_HIP (-4)
#define MPV_E_ARG (-5)

// thread-local error text (mpv_last_error)
void mpv_set_error(const char* fmt, ...);
int mpv_check_launch(const char* what);

#define MPV_REQUIRE(cond, code, ...)        \
  do {                                      \
    if (!(cond)) {                          \
      mpv_set_error(__VA_ARGS__);           \
      return (code);                        \
    }                                       \
  } while (0)

// ---------------------------------------------------------------------------------------
// Row map: logical row r -> physical row (r / group) * stride + (r % group) + offset.
// group == 0 means identity.  Used to scatter/gather token rows around the per-frame cls
// slot ([B,T,1+N,D] stream layout) and the extra bias-kv token of the abstractor.
struct RowMap {
  int group, stride, offset;
};
__host__ __device__ __forceinline__ long long map_row(RowMap m, long long r) {
  if (m.group == 0) return r;
  const unsigned ru = (unsigned)r, g = (unsigned)m.group;   // rows < 2^31: 32-bit divide (the 64-bit one is a long routine)
  const unsigned q = ru / g;
  return (long long)q * (long long)m.stride + (long long)(ru - q * g) + m.offset;
}

// ---------------------------------------------------------------------------------------
__device__ __forceinline__ float bf2f(bf16 v) { return (float)v; }
__device__ __forceinline__ bf16 f2bf(float v) { return (bf16)v; }

__device__ __forceinline__ f32x4 cvt4(bf16x4 v) { return __builtin_convertvector(v, f32x4); }
__device__ __forceinline__ bf16x4 cvt4(f32x4 v) { return __builtin_convertvector(v, bf16x4); }
__device__ __forceinline__ f32x8 cvt8(bf16x8 v) { return __builtin_convertvector(v, f32x8); }
__device__ __forceinline__ bf16x8 cvt8(f32x8 v) { return __builtin_convertvector(v, bf16x8); }

// ---------------------------------------------------------------------------------------
// Counter-based dropout RNG: two rounds of a 32-bit avalanche mix keyed by a 64-bit seed.
// keep(idx) is a pure function of (seed, idx) so forward and backward regenerate the same
// mask without storing it.  (The reference uses torch's Philox stream; masks cannot be
// bit-compatible with it, parity is tested with dropout off and statistically with it on.)
__host__ __device__ __forceinline__ uint32_t mpv_mix32(uint32_t x) {
  x ^= x >> 16;
  x *= 0x7feb352du;
  x ^= x >> 15;
  x *= 0x846ca68bu;
  x ^= x >> 16;
  return x;
}
// The seed halves go through the hash themselves before they meet the counter (an XOR of the raw seed into the counter
// would make the streams of two seeds index-permutations of each other); both seed hashes are loop-invariant.
__host__ __device__ __forceinline__ uint32_t mpv_rand32(uint64_t seed, uint64_t idx) {
  uint32_t lo = (uint32_t)idx, hi = (uint32_t)(idx >> 32);
  uint32_t x = mpv_mix32(lo + mpv_mix32((uint32_t)seed));
  x = mpv_mix32(x + mpv_mix32((uint32_t)(seed >> 32) ^ 0x85ebca6bu) + hi * 0x9e3779b9u);
  return x;
}
// keep-threshold on the top 24 bits: keep iff r24 >= p * 2^24
__host__ __device__ __forceinline__ uint32_t mpv_drop_threshold(float p) {
  return (uint32_t)(p * 16777216.0f);
}
__host__ __device__ __forceinline__ bool mpv_keep(uint64_t seed, uint64_t idx, uint32_t thr) {
  return (mpv_rand32(seed, idx) >> 8) >= thr;
}

// ---------------------------------------------------------------------------------------
// Activation math for the GEMM epilogues.  A 256x256 output tile is 128 activations per thread with the matrix pipe idle,
// so these are built from the hardware transcendentals (v_exp_f32, v_rcp_f32: ~1 ulp) instead of the libm routines
// (measured: erff/tanhf epilogues cost 10-20 us per tile, more than the K = 768 main loop).  Errors are far below bf16
// resolution: erf by Abramowitz-Stegun 7.1.26 (|err| <= 1.5e-7), tanh(u) = 1 - 2 / (e^{2u} + 1).
__device__ __forceinline__ float mpv_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }
// returns erf(x / sqrt(2)) and hands back e = exp(-x^2 / 2) for the derivative
__device__ __forceinline__ float mpv_erf_rsqrt2(float x, float& e) {
  const float u = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, u, 1.0f));
  float q = fmaf(1.061405429f, t, -1.453152027f);
  q = fmaf(q, t, 1.421413741f);
  q = fmaf(q, t, -0.284496736f);
  q = fmaf(q, t, 0.254829592f);
  e = mpv_exp(-u * u);
  return copysignf(fmaf(-q * t, e, 1.0f), x);
}
__device__ __forceinline__ float mpv_tanh(float u) {
  return fmaf(-2.0f, __builtin_amdgcn_rcpf(mpv_exp(2.0f * u) + 1.0f), 1.0f);   // e^{2u} = inf -> 1, = 0 -> -1
}
__device__ __forceinline__ float gelu_erf_f(float x) {
  float e;
  return 0.5f * x * (1.0f + mpv_erf_rsqrt2(x, e));
}
__device__ __forceinline__ float gelu_erf_grad_f(float x) {
  float e;
  const float cdf = 0.5f * (1.0f + mpv_erf_rsqrt2(x, e));
  return fmaf(x * 0.3989422804014327f, e, cdf);
}
__device__ __forceinline__ float gelu_tanh_f(float x) {
  return 0.5f * x * (1.0f + mpv_tanh(0.79788456f * x * fmaf(0.044715f * x, x, 1.0f)));
}
__device__ __forceinline__ float gelu_tanh_grad_f(float x) {
  const float t = mpv_tanh(0.79788456f * x * fmaf(0.044715f * x, x, 1.0f));
  return 0.5f * x * ((1.0f - t * t) * fmaf(0.1070322243f * x, x, 0.79788456f)) + 0.5f * (1.0f + t);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// buffer resource over [ptr, ptr+bytes): out-of-range 16-byte loads return zeros.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
