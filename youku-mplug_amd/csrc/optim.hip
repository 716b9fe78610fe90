// optim.hip -- flat-buffer AdamW + global gradient norm for gfx950 (HBM-bound, ~30 B/param).
// Replaces DeepSpeed FusedAdam + bf16 master-weight handling + global-norm clipping
// (run_pretrain_distributed_gpt3.py:137; utils.py:490-529); math of optim/adamw.py:66-115.
#include "mpv_common.h"
#include "../../include/mpv.h"

namespace {

// Round 6: the per-workgroup sums go to a partials array and ONE workgroup adds them up in index order (grad_sumsq_final_kernel).  Until
// round 5 every workgroup added its sum to the scalar with atomicAdd: the order of 2048 floating-point additions then depended on which
// workgroup arrived first, the norm differed in its last bits from process to process, and whenever the clip was ACTIVE (norm > max_norm)
// so did the clip coefficient, the update and every later step -- found through config D's final loss, which varied from run to run on one
// tree (profiles/r06_c8_first_step_across_processes.log: the first entry point whose output differed between processes).
__global__ __launch_bounds__(256) void grad_sumsq_kernel(const bf16* __restrict__ g, long long n, float* __restrict__ out) {
  __shared__ float red[4];
  float s = 0.f;
  const long long n8 = n / 8;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
    const f32x8 v = cvt8(*(const bf16x8*)(g + i * 8));
#pragma unroll
    for (int e = 0; e < 8; ++e) s += v[e] * v[e];
  }
  if (blockIdx.x == 0 && threadIdx.x < (n - n8 * 8)) {
    const float v = bf2f(g[n8 * 8 + threadIdx.x]);
    s += v * v;
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = red[0] + red[1] + red[2] + red[3];      // out: this launch's partials
}
// *sumsq += sum of `nblk` partials (<= 2048), a fixed order: thread t adds partials t, t + 256, ...; wave_sum; four waves in order
__global__ __launch_bounds__(256) void grad_sumsq_final_kernel(const float* __restrict__ part, int nblk, float* __restrict__ sumsq) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < nblk; i += 256) s += part[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) *sumsq += (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void adamw_kernel(bf16* __restrict__ p16, float* __restrict__ p, float* __restrict__ m,
                                                    float* __restrict__ v, const bf16* __restrict__ g, long long n, float lr,
                                                    float b1, float b2, float eps, float wd, float inv_bc1, float inv_sqrt_bc2,
                                                    float grad_scale, const float* __restrict__ sumsq, float max_norm) {
  float gs = grad_scale;
  if (sumsq && max_norm > 0.f) {
    const float norm = sqrtf(*sumsq) * grad_scale;      // norm of the scaled gradients
    const float coef = max_norm / (norm + 1e-6f);        // torch.nn.utils.clip_grad_norm_
    if (coef < 1.0f) gs *= coef;
  }
  const long long n4 = n / 4;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    f32x4 pp = *(const f32x4*)(p + i * 4), mm = *(const f32x4*)(m + i * 4), vv = *(const f32x4*)(v + i * 4);
    const f32x4 gg = cvt4(*(const bf16x4*)(g + i * 4)) * gs;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      pp[e] *= (1.0f - lr * wd);
      mm[e] = mm[e] * b1 + (1.0f - b1) * gg[e];
      vv[e] = vv[e] * b2 + (1.0f - b2) * gg[e] * gg[e];
      const float denom = sqrtf(vv[e]) * inv_sqrt_bc2 + eps;
      pp[e] -= (lr * inv_bc1) * mm[e] / denom;
    }
    *(f32x4*)(p + i * 4) = pp;
    *(f32x4*)(m + i * 4) = mm;
    *(f32x4*)(v + i * 4) = vv;
    *(bf16x4*)(p16 + i * 4) = cvt4(pp);
  }
}

struct GroupHyper {
  float lr[8];
  float wd[8];
};

// Grouped variant: the flat buffer is laid out in backward-completion order (so DP buckets are
// contiguous slices); every 256-element tile carries the id of its parameter group.
__global__ __launch_bounds__(256) void adamw_grouped_kernel(bf16* __restrict__ p16, float* __restrict__ p, float* __restrict__ m,
                                                            float* __restrict__ v, const bf16* __restrict__ g, long long ntiles,
                                                            const uint8_t* __restrict__ tile_group, GroupHyper hp, float b1,
                                                            float b2, float eps, float inv_bc1, float inv_sqrt_bc2,
                                                            float grad_scale, const float* __restrict__ sumsq, float max_norm,
                                                            const float* __restrict__ hyper_dev) {
  if (hyper_dev) {      // step-dependent hyper-parameters from device memory (mpv_adamw_step_grouped_dev): lr[8], wd[8], 1/bc1, 1/sqrt(bc2)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      hp.lr[i] = hyper_dev[i];
      hp.wd[i] = hyper_dev[8 + i];
    }
    inv_bc1 = hyper_dev[16];
    inv_sqrt_bc2 = hyper_dev[17];
  }
  float gs = grad_scale;
  if (sumsq && max_norm > 0.f) {
    const float norm = sqrtf(*sumsq) * grad_scale;
    const float coef = max_norm / (norm + 1e-6f);
    if (coef < 1.0f) gs *= coef;
  }
  const int lane4 = threadIdx.x & 63;       // 64 lanes x 4 elements = one 256-element tile per wave
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // Two tiles per wave and iteration, every load unconditional (round 4): the tile's group id used to be fetched first and waited for
  // (the `continue` of a padding tile made the data loads depend on it) -- two round trips per 3.5 KB of a wave.  Now the ids of the
  // NEXT pair are requested with the current pair's data, padding tiles (rare) are loaded and computed like any other and only their
  // stores are skipped.
  const long long stride = (long long)gridDim.x * 8;
  long long t = ((long long)blockIdx.x * 4 + wave) * 2;
  if (t >= ntiles) return;
  const long long last = ntiles - 1;
  int g0 = tile_group[t], g1 = tile_group[min(t + 1, last)];
  for (; t < ntiles; t += stride) {
    const long long tn = t + stride;
    const int n0 = tile_group[min(tn, last)], n1 = tile_group[min(tn + 1, last)];
    const bool two = t + 1 < ntiles;
    const long long i0 = t * 256 + lane4 * 4, i1 = (two ? t + 1 : t) * 256 + lane4 * 4;
    f32x4 pa = *(const f32x4*)(p + i0), ma = *(const f32x4*)(m + i0), va = *(const f32x4*)(v + i0);
    f32x4 pb = *(const f32x4*)(p + i1), mb = *(const f32x4*)(m + i1), vb = *(const f32x4*)(v + i1);
    const f32x4 ga = cvt4(*(const bf16x4*)(g + i0)) * gs, gb = cvt4(*(const bf16x4*)(g + i1)) * gs;
    auto upd = [&](f32x4& pp, f32x4& mm, f32x4& vv, const f32x4& gg, int grp) {
      const float lr = hp.lr[grp & 7], wd = hp.wd[grp & 7];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        pp[e] *= (1.0f - lr * wd);
        mm[e] = mm[e] * b1 + (1.0f - b1) * gg[e];
        vv[e] = vv[e] * b2 + (1.0f - b2) * gg[e] * gg[e];
        const float denom = sqrtf(vv[e]) * inv_sqrt_bc2 + eps;
        pp[e] -= (lr * inv_bc1) * mm[e] / denom;
      }
    };
    upd(pa, ma, va, ga, g0);
    upd(pb, mb, vb, gb, g1);
    if (g0 < 8) {                            // (>= 8: padding / frozen tile)
      *(f32x4*)(p + i0) = pa;
      *(f32x4*)(m + i0) = ma;
      *(f32x4*)(v + i0) = va;
      *(bf16x4*)(p16 + i0) = cvt4(pa);
    }
    if (two && g1 < 8) {
      *(f32x4*)(p + i1) = pb;
      *(f32x4*)(m + i1) = mb;
      *(f32x4*)(v + i1) = vb;
      *(bf16x4*)(p16 + i1) = cvt4(pb);
    }
    g0 = n0;
    g1 = n1;
  }
}

}  // namespace

extern "C" int mpv_adamw_step_grouped(void* param_bf16, float* master, float* exp_avg, float* exp_avg_sq,
                                      const void* grad_bf16, int64_t n, const uint8_t* tile_group, const float* lrs,
                                      const float* wds, int ngroups, float beta1, float beta2, float eps, int step,
                                      float grad_scale, const float* sumsq, float max_norm, hipStream_t stream) {
  MPV_REQUIRE(param_bf16 && master && exp_avg && exp_avg_sq && grad_bf16 && tile_group && lrs && wds, MPV_E_ARG,
              "mpv_adamw_step_grouped: null pointer");
  MPV_REQUIRE(n >= 0 && n % 256 == 0, MPV_E_SHAPE, "mpv_adamw_step_grouped: n (%lld) must be a multiple of the 256-element tile", (long long)n);
  MPV_REQUIRE(ngroups >= 1 && ngroups <= 8, MPV_E_ARG, "mpv_adamw_step_grouped: 1..8 parameter groups");
  MPV_REQUIRE(step >= 1, MPV_E_ARG, "mpv_adamw_step_grouped: step counts from 1");
  if (n == 0) return MPV_OK;
  GroupHyper hp = {};
  for (int i = 0; i < ngroups; ++i) {
    hp.lr[i] = lrs[i];
    hp.wd[i] = wds[i];
  }
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  const long long ntiles = n / 256;
  long long blocks = (ntiles + 7) / 8;      // a wave takes two tiles per iteration
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(adamw_grouped_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (bf16*)param_bf16, master, exp_avg,
                     exp_avg_sq, (const bf16*)grad_bf16, ntiles, tile_group, hp, beta1, beta2, eps, (float)(1.0 / bc1),
                     (float)(1.0 / sqrt(bc2)), grad_scale, sumsq, max_norm, (const float*)nullptr);
  return mpv_check_launch("mpv_adamw_step_grouped");
}

extern "C" int mpv_adamw_hyper_pack(const float* lrs, const float* wds, int ngroups, float beta1, float beta2, int step, float* out18) {
  MPV_REQUIRE(lrs && wds && out18 && ngroups >= 1 && ngroups <= 8 && step >= 1, MPV_E_ARG, "mpv_adamw_hyper_pack: bad argument");
  for (int i = 0; i < 8; ++i) {
    out18[i] = i < ngroups ? lrs[i] : 0.f;
    out18[8 + i] = i < ngroups ? wds[i] : 0.f;
  }
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);      // exactly as mpv_adamw_step_grouped
  out18[16] = (float)(1.0 / bc1);
  out18[17] = (float)(1.0 / sqrt(bc2));
  return MPV_OK;
}

extern "C" int mpv_adamw_step_grouped_dev(void* param_bf16, float* master, float* exp_avg, float* exp_avg_sq,
                                          const void* grad_bf16, int64_t n, const uint8_t* tile_group, const float* hyper_dev,
                                          float beta1, float beta2, float eps, float grad_scale, const float* sumsq,
                                          float max_norm, hipStream_t stream) {
  MPV_REQUIRE(param_bf16 && master && exp_avg && exp_avg_sq && grad_bf16 && tile_group && hyper_dev, MPV_E_ARG,
              "mpv_adamw_step_grouped_dev: null pointer");
  MPV_REQUIRE(n >= 0 && n % 256 == 0, MPV_E_SHAPE, "mpv_adamw_step_grouped_dev: n (%lld) must be a multiple of the 256-element tile", (long long)n);
  if (n == 0) return MPV_OK;
  GroupHyper hp = {};
  const long long ntiles = n / 256;
  long long blocks = (ntiles + 7) / 8;      // a wave takes two tiles per iteration
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(adamw_grouped_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (bf16*)param_bf16, master, exp_avg,
                     exp_avg_sq, (const bf16*)grad_bf16, ntiles, tile_group, hp, beta1, beta2, eps, 1.0f, 1.0f, grad_scale, sumsq,
                     max_norm, hyper_dev);
  return mpv_check_launch("mpv_adamw_step_grouped_dev");
}

extern "C" size_t mpv_grad_sumsq_workspace_size(void) { return 2048 * sizeof(float); }

extern "C" int mpv_grad_sumsq(const void* grad, int64_t n, float* sumsq, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  MPV_REQUIRE(grad && sumsq, MPV_E_ARG, "mpv_grad_sumsq: null pointer");
  MPV_REQUIRE(n >= 0 && (((uintptr_t)grad) & 15) == 0, MPV_E_ALIGN, "mpv_grad_sumsq: grad must be 16-byte aligned");
  MPV_REQUIRE(workspace && workspace_bytes >= 2048 * sizeof(float) && (((uintptr_t)workspace) & 3) == 0, MPV_E_ARG,
              "mpv_grad_sumsq: needs mpv_grad_sumsq_workspace_size() bytes of workspace (the per-workgroup partial sums)");
  if (n == 0) return MPV_OK;
  long long blocks = (n / 8 + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(grad_sumsq_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (const bf16*)grad, (long long)n, (float*)workspace);
  hipLaunchKernelGGL(grad_sumsq_final_kernel, dim3(1), dim3(256), 0, stream, (const float*)workspace, (int)blocks, sumsq);
  return mpv_check_launch("mpv_grad_sumsq");
}

extern "C" int mpv_adamw_step(void* param_bf16, float* master, float* exp_avg, float* exp_avg_sq, const void* grad_bf16,
                              int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                              float grad_scale, const float* sumsq, float max_norm, hipStream_t stream) {
  MPV_REQUIRE(param_bf16 && master && exp_avg && exp_avg_sq && grad_bf16, MPV_E_ARG, "mpv_adamw_step: null pointer");
  MPV_REQUIRE(n >= 0 && n % 4 == 0, MPV_E_SHAPE, "mpv_adamw_step: n (%lld) must be a multiple of 4 (pad the flat buffer)", (long long)n);
  MPV_REQUIRE(step >= 1, MPV_E_ARG, "mpv_adamw_step: step counts from 1");
  if (n == 0) return MPV_OK;
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  long long blocks = (n / 4 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (bf16*)param_bf16, master, exp_avg, exp_avg_sq,
                     (const bf16*)grad_bf16, (long long)n, lr, beta1, beta2, eps, weight_decay, (float)(1.0 / bc1),
                     (float)(1.0 / sqrt(bc2)), grad_scale, sumsq, max_norm);
  return mpv_check_launch("mpv_adamw_step");
}
