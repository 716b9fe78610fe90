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
#define MPV_LN_DPARAM_DEFER 2   // include/mpv.h: mpv_layernorm_bwd leaves its dgamma/dbeta partials for mpv_layernorm_dparam_finish

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
// Counter-based dropout RNG: a 32-bit avalanche mix keyed by a 64-bit seed.
// keep is a pure function of (seed, element index) so forward and backward regenerate the same
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
// One hash serves TWO adjacent elements (16 random bits each): the 32-bit integer multiply is quarter rate on CDNA, so a
// hash per element (two rounds, 4 multiplies) cost ~35 issue slots per element -- 10 us per 256x256 tile in a GEMM
// epilogue.  `ctr` is the 64-bit index of the pair's FIRST element; every site pairs along its fastest dimension with
// the even coordinate first (hidden states: columns n, n+1 with n even; attention: keys k, k+1 with k even), and the
// forward and backward kernels of a site use the same rule, so masks are reproduced exactly.  The seed halves go through
// the hash themselves (wave-uniform, hoisted) before they meet the counter; the high counter word only matters beyond
// 2^32 elements and is folded in by XOR.  Drop probability is quantised to 1/65536 (0.1 -> 0.100006).
__host__ __device__ __forceinline__ uint32_t mpv_rand_pair(uint64_t seed, uint64_t ctr) {
  const uint32_t ka = mpv_mix32((uint32_t)seed), kb = mpv_mix32((uint32_t)(seed >> 32) ^ 0x85ebca6bu);
  return mpv_mix32(((uint32_t)ctr + ka) ^ ((uint32_t)(ctr >> 32) + kb));
}
// Indirect seeds (include/mpv.h, "dropout seeds"): a seed argument with bit 63 set carries the DEVICE address of the 64-bit seed in
// its low 63 bits; the kernel reads it when it runs.  Kernel arguments are frozen when a step is captured into a HIP graph --
// this is what lets a replay draw fresh masks (the engine rewrites the 8 bytes before every replay).  Wave-uniform: one scalar load.
#define MPV_SEED_INDIRECT_BIT (1ull << 63)
__device__ __forceinline__ uint64_t mpv_resolve_seed(uint64_t s) {
#ifdef MPV_AB_DIRECT_SEED      // measurement build only (cost of the indirection: tools/ab_same_box.sh lib ...)
  return s;
#else
  if (!(s & MPV_SEED_INDIRECT_BIT)) return s;
  return *(const uint64_t*)(uintptr_t)(s & ~MPV_SEED_INDIRECT_BIT);
#endif
}
// The seed's two hashed halves, for kernels that keep them in SCALAR registers over a long loop (readfirstlane: the value is
// wave-uniform, but the compiler does not always know it and parks it in VGPRs of kernels that have none to spare).
struct MpvSeedKeys {
  uint32_t ka, kb;
};
__device__ __forceinline__ MpvSeedKeys mpv_seed_keys(uint64_t seed) {
  return MpvSeedKeys{(uint32_t)__builtin_amdgcn_readfirstlane((int)mpv_mix32((uint32_t)seed)),
                     (uint32_t)__builtin_amdgcn_readfirstlane((int)mpv_mix32((uint32_t)(seed >> 32) ^ 0x85ebca6bu))};
}
__device__ __forceinline__ uint32_t mpv_rand_pair_k(MpvSeedKeys k, uint64_t ctr) {      // == mpv_rand_pair(seed, ctr)
  return mpv_mix32(((uint32_t)ctr + k.ka) ^ ((uint32_t)(ctr >> 32) + k.kb));
}
// keep-threshold on 16 bits: keep iff r16 >= p * 2^16
__host__ __device__ __forceinline__ uint32_t mpv_drop_threshold(float p) {
  return (uint32_t)(p * 65536.0f + 0.5f);
}
// element `half` (0/1) of the pair whose first element has index ctr
__host__ __device__ __forceinline__ bool mpv_keep(uint64_t seed, uint64_t ctr, int half, uint32_t thr) {
  const uint32_t r = mpv_rand_pair(seed, ctr);
  return (half ? r >> 16 : r & 0xffffu) >= thr;
}
// dropout on N (4 or 8) consecutive elements starting at index `base` (an even coordinate of the fastest dimension)
template <typename V, int N>
__device__ __forceinline__ V mpv_dropout_vec(V v, uint64_t seed, uint64_t base, uint32_t thr, float scale) {
#pragma unroll
  for (int e = 0; e < N; e += 2) {
    const uint32_t r = mpv_rand_pair(seed, base + e);
    v[e] = (r & 0xffffu) >= thr ? v[e] * scale : 0.f;
    v[e + 1] = (r >> 16) >= thr ? v[e + 1] * scale : 0.f;
  }
  return v;
}

// ---------------------------------------------------------------------------------------
// Activation math for the GEMM epilogues.  A 256x256 output tile is 128 activations per thread with the matrix pipe idle:
// the epilogue is VALU-bound on this math (measured: libm erff/tanhf 10-20 us per tile; v_exp_f32 + v_rcp_f32 forms
// 9-11 us, the quarter-rate transcendentals being half of it; the K = 768 main loop is 15.7 us).  Both GELU flavours are
// therefore evaluated as x * (1/2 + S(x)) with the odd part S(x) = Phi~(x) - 1/2 a clamped odd minimax polynomial
// x * P(x^2) on |x| <= 4 -- pure FMAs, which the compiler packs two lanes wide (v_pk_fma_f32) when handed f32x2 --
// and likewise GELU'(x) - 1/2 = x * Q(x^2).  Fitted in tools/probe/fit_gelu_poly.py (Lawson-weighted least squares,
// verified in fp32 Horner form): max abs error of Phi~ 2.2e-5 (erf) / 2.8e-5 (tanh), of GELU' 7.4e-5 / 1.1e-4 inside
// the clamp, 5e-4 beyond it (the frozen tail value), all a fraction of a bf16 half-ulp (2e-3 at 1.0).
__device__ __forceinline__ float mpv_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }
__device__ __forceinline__ float mpv_tanh(float u) {
  return fmaf(-2.0f, __builtin_amdgcn_rcpf(mpv_exp(2.0f * u) + 1.0f), 1.0f);   // e^{2u} = inf -> 1, = 0 -> -1
}

#define MPV_GELU_CLAMP 4.0f
__device__ __forceinline__ float mpv_fma_t(float a, float b, float c) { return fmaf(a, b, c); }
__device__ __forceinline__ f32x2 mpv_fma_t(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ float mpv_clamp_t(float x) { return __builtin_amdgcn_fmed3f(x, -MPV_GELU_CLAMP, MPV_GELU_CLAMP); }
__device__ __forceinline__ f32x2 mpv_clamp_t(f32x2 x) { return f32x2{mpv_clamp_t(x[0]), mpv_clamp_t(x[1])}; }
template <typename T>
__device__ __forceinline__ T mpv_splat(float v);
template <>
__device__ __forceinline__ float mpv_splat<float>(float v) { return v; }
template <>
__device__ __forceinline__ f32x2 mpv_splat<f32x2>(float v) { return f32x2{v, v}; }

// P(w) of S(x) = x * P(x^2): KIND 1 = erf GELU (Phi(x) - 1/2), 2 = tanh GELU (tanh(0.79788456 x (1 + 0.044715 x^2)) / 2)
template <int KIND, typename T>
__device__ __forceinline__ T mpv_gelu_cdf_poly(T w) {
  if constexpr (KIND == 1) {
    T q = mpv_fma_t(mpv_splat<T>(-1.580771181e-09f), w, mpv_splat<T>(1.217102152e-07f));
    q = mpv_fma_t(q, w, mpv_splat<T>(-4.100845217e-06f));
    q = mpv_fma_t(q, w, mpv_splat<T>(8.066713781e-05f));
    q = mpv_fma_t(q, w, mpv_splat<T>(-1.048202743e-03f));
    q = mpv_fma_t(q, w, mpv_splat<T>(9.664868936e-03f));
    q = mpv_fma_t(q, w, mpv_splat<T>(-6.617537141e-02f));
    return mpv_fma_t(q, w, mpv_splat<T>(3.988475204e-01f));
  } else {
    T q = mpv_fma_t(mpv_splat<T>(-1.796823246e-09f), w, mpv_splat<T>(1.362741813e-07f));
    q = mpv_fma_t(q, w, mpv_splat<T>(-4.502460342e-06f));
    q = mpv_fma_t(q, w, mpv_splat<T>(8.644891204e-05f));
    q = mpv_fma_t(q, w, mpv_splat<T>(-1.093709492e-03f));
    q = mpv_fma_t(q, w, mpv_splat<T>(9.846926667e-03f));
    q = mpv_fma_t(q, w, mpv_splat<T>(-6.644576788e-02f));
    return mpv_fma_t(q, w, mpv_splat<T>(3.988276422e-01f));
  }
}
// Q(w) of GELU'(x) - 1/2 = x * Q(x^2)
template <int KIND, typename T>
__device__ __forceinline__ T mpv_gelu_grad_poly(T w) {
  if constexpr (KIND == 1) {
    T q = mpv_fma_t(mpv_splat<T>(9.796049527e-10f), w, mpv_splat<T>(-8.218798797e-08f));
    q = mpv_fma_t(q, w, mpv_splat<T>(3.028347010e-06f));
    q = mpv_fma_t(q, w, mpv_splat<T>(-6.495756679e-05f));
    q = mpv_fma_t(q, w, mpv_splat<T>(9.073265246e-04f));
    q = mpv_fma_t(q, w, mpv_splat<T>(-8.716319688e-03f));
    q = mpv_fma_t(q, w, mpv_splat<T>(5.845610052e-02f));
    q = mpv_fma_t(q, w, mpv_splat<T>(-2.648265362e-01f));
    return mpv_fma_t(q, w, mpv_splat<T>(7.976095676e-01f));
  } else {
    T q = mpv_fma_t(mpv_splat<T>(1.172013597e-09f), w, mpv_splat<T>(-9.666642597e-08f));
    q = mpv_fma_t(q, w, mpv_splat<T>(3.483976570e-06f));
    q = mpv_fma_t(q, w, mpv_splat<T>(-7.268741319e-05f));
    q = mpv_fma_t(q, w, mpv_splat<T>(9.829062037e-04f));
    q = mpv_fma_t(q, w, mpv_splat<T>(-9.133556858e-03f));
    q = mpv_fma_t(q, w, mpv_splat<T>(5.960476771e-02f));
    q = mpv_fma_t(q, w, mpv_splat<T>(-2.658853233e-01f));
    return mpv_fma_t(q, w, mpv_splat<T>(7.975339890e-01f));
  }
}
// GELU(x) = x~ * (1/2 + xc * P(xc^2)), xc = clamp(x, -4, 4), x~ = max(x, -4).  Left of the clamp the bracket is frozen at
// Phi~(-4) = 3e-5 +- 2e-5 (not 0), so the multiplier is frozen there too: the far negative tail returns >= -2.2e-4 where the
// true value is -0 (with the raw x it grew linearly: -3e-3 at x = -100; CLIP towers do produce such pre-activations).
__device__ __forceinline__ float mpv_tail_t(float x) { return fmaxf(x, -MPV_GELU_CLAMP); }
__device__ __forceinline__ f32x2 mpv_tail_t(f32x2 x) { return f32x2{fmaxf(x[0], -MPV_GELU_CLAMP), fmaxf(x[1], -MPV_GELU_CLAMP)}; }
template <int KIND, typename T>
__device__ __forceinline__ T mpv_gelu_t(T x) {
  const T xc = mpv_clamp_t(x);
  const T s = xc * mpv_gelu_cdf_poly<KIND>(xc * xc);
  const T xe = mpv_tail_t(x);
  return mpv_fma_t(xe, s, xe * mpv_splat<T>(0.5f));
}
// dy * GELU'(x) = dy * (1/2 + xc * Q(xc^2))
template <int KIND, typename T>
__device__ __forceinline__ T mpv_gelu_grad_mul_t(T dy, T x) {
  const T xc = mpv_clamp_t(x);
  return mpv_fma_t(dy * xc, mpv_gelu_grad_poly<KIND>(xc * xc), dy * mpv_splat<T>(0.5f));
}
// tanh GELU and its derivative from ONE evaluation of the odd polynomial: with s = tanh(u) / 2 (u = 0.79788456 x (1 + 0.044715 x^2)),
// GELU = x~ (1/2 + s) and GELU' = 1/2 + s + x/2 (1 - 4 s^2)(0.79788456 + 0.1070322243 x^2) -- four more packed FMAs instead of
// the second degree-8 polynomial (the forward epilogue that parks GELU' for the dgrad: preact_deriv).  Same clamp as both.
template <typename T>
__device__ __forceinline__ void mpv_gelu_tanh_both_t(T x, T& gelu, T& deriv) {
  const T xc = mpv_clamp_t(x);
  const T w = xc * xc;
  const T s = xc * mpv_gelu_cdf_poly<2>(w);
  const T xe = mpv_tail_t(x);
  gelu = mpv_fma_t(xe, s, xe * mpv_splat<T>(0.5f));
  const T sech2h = mpv_fma_t(s * mpv_splat<T>(-2.0f), s, mpv_splat<T>(0.5f));                       // (1 - 4 s^2) / 2
  const T du = mpv_fma_t(w, mpv_splat<T>(0.1070322243f), mpv_splat<T>(0.79788456f));                // u'(x)
  deriv = mpv_fma_t(xc * sech2h, du, s + mpv_splat<T>(0.5f));
}
__device__ __forceinline__ float gelu_erf_f(float x) { return mpv_gelu_t<1>(x); }
__device__ __forceinline__ float gelu_erf_grad_f(float x) { return mpv_gelu_grad_mul_t<1>(1.0f, x); }
__device__ __forceinline__ float gelu_tanh_f(float x) { return mpv_gelu_t<2>(x); }
__device__ __forceinline__ float gelu_tanh_grad_f(float x) { return mpv_gelu_grad_mul_t<2>(1.0f, x); }

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
