// augment.hip -- the clip-consistent RandAugment of the reference's loaders on the GPU, for gfx950.
//
// Replaces (reference file:line): dataset/video_utils/randaugment_video.py -- contrast_func :120-130, brightness_func :133-139,
// sharpness_func :142-160, shear_x/y_func :163-167 / :198-202, translate_x/y_func :170-187, rotate_func :67-75 (the nine ops of
// dataset/__init__.py:65-66, 75-76), applied by TemporalConsistentRandomAugment._aug :355-361 to every frame of a clip; plus the
// two ends of the pipeline around it: the uint8 view the augmentation takes of the resized clip (frames.numpy().astype(np.uint8),
// :343) and ClipToTensor + Normalize behind it (dataset/__init__.py:68-69).
// Integer / table arithmetic throughout: the results are bit-exact against oracle/augment.py (tests/test_kernels_gpu.py).  The
// opencv pieces (warpAffine's fixed-point bilinear remap, filter2D's float accumulation) follow the restatement there -- see its
// header for what is pinned against the reference and what is not.
#include "mpv_common.h"
#include "../../include/mpv.h"

namespace {

constexpr int AUG_BLOCK = 256;
static unsigned aug_grid(long long n) {
  long long b = (n + AUG_BLOCK - 1) / AUG_BLOCK;
  return (unsigned)(b < 1 ? 1 : (b > 65535 ? 65535 : b));
}

// per-frame channel sums (exact integers: any summation order gives numpy's float64 sums)
__global__ void channel_sums_kernel(const uint8_t* __restrict__ frames, long long hw, unsigned long long* __restrict__ sums) {
  const int t = blockIdx.y;
  const uint8_t* f = frames + (long long)t * hw * 3;
  unsigned long long s[3] = {0, 0, 0};
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < hw; i += (long long)gridDim.x * blockDim.x) {
    s[0] += f[i * 3];
    s[1] += f[i * 3 + 1];
    s[2] += f[i * 3 + 2];
  }
  __shared__ unsigned long long red[3][AUG_BLOCK / 64];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    unsigned long long v = s[c];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0) red[c][threadIdx.x >> 6] = v;
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    unsigned long long v = 0;
    for (int w = 0; w < AUG_BLOCK / 64; ++w) v += red[threadIdx.x][w];
    atomicAdd(sums + t * 3 + threadIdx.x, v);
  }
}

// table[img] with the table of contrast_func (op 0: per-frame mean, float64 arithmetic as numpy evaluates it) or of
// brightness_func (op 1: float32); one 256-entry table per block in LDS
__global__ void pointwise_kernel(uint8_t* __restrict__ frames, long long hw, int op, double factor, const unsigned long long* __restrict__ sums) {
#pragma clang fp contract(off)
  __shared__ uint8_t table[256];
  const int t = blockIdx.y;
  {
    const int el = threadIdx.x;
    double v;
    if (op == 0) {
      const double n = (double)hw;
      const double m0 = (double)sums[t * 3] / n, m1 = (double)sums[t * 3 + 1] / n, m2 = (double)sums[t * 3 + 2] / n;
      const double mean = (m0 * 0.114 + m1 * 0.587) + m2 * 0.299;      // np.sum of a 3-vector: left to right
      v = ((double)el - mean) * factor + mean;
      v = v < 0.0 ? 0.0 : (v > 255.0 ? 255.0 : v);
    } else {
      float w = (float)el * (float)factor;
      w = w < 0.f ? 0.f : (w > 255.f ? 255.f : w);
      v = (double)w;
    }
    table[el] = (uint8_t)(int)v;                                        // .astype(np.uint8): truncation
  }
  __syncthreads();
  uint8_t* f = frames + (long long)t * hw * 3;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < hw * 3; i += (long long)gridDim.x * blockDim.x) f[i] = table[f[i]];
}

__device__ __forceinline__ int reflect101(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }

// sharpness_func: degenerate = filter2D(img, ones / 13 with 5 / 13 in the centre) (float32 accumulation in tap order, BORDER_REFLECT_101,
// round half to even), interior pixels blended in float32 and truncated; factor 0: the filtered image everywhere
__global__ void sharpness_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int T, int H, int W, float factor, int degenerate_only) {
#pragma clang fp contract(off)
  const float k1 = 1.0f / 13.0f, k5 = 5.0f / 13.0f;
  const long long n = (long long)T * H * W;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(i % W), y = (int)((i / W) % H);
    const long long fb = (i / ((long long)W * H)) * (long long)H * W * 3;
    const bool interior = x > 0 && x < W - 1 && y > 0 && y < H - 1;
    for (int c = 0; c < 3; ++c) {
      const uint8_t px = in[fb + ((long long)y * W + x) * 3 + c];
      if (!interior && !degenerate_only) {
        out[fb + ((long long)y * W + x) * 3 + c] = px;
        continue;
      }
      float acc = 0.f;
      for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
          const int yy = reflect101(y + dy, H), xx = reflect101(x + dx, W);
          acc = acc + ((dy == 0 && dx == 0) ? k5 : k1) * (float)in[fb + ((long long)yy * W + xx) * 3 + c];
        }
      float d = rintf(acc);
      d = d < 0.f ? 0.f : (d > 255.f ? 255.f : d);
      if (degenerate_only) {
        out[fb + ((long long)y * W + x) * 3 + c] = (uint8_t)d;
      } else {
        const float r = d + factor * ((float)px - d);
        out[fb + ((long long)y * W + x) * 3 + c] = (uint8_t)(int)r;      // float32 -> uint8 as numpy casts: through a truncated integer
      }
    }
  }
}

struct WarpArgs {
  const uint8_t* in;
  uint8_t* out;
  int T, H, W;
  double m[6];         // the INVERTED 2 x 3 matrix (dst -> src)
  int fill[3];
};
// cv::warpAffine, INTER_LINEAR, BORDER_CONSTANT (oracle/augment.py: warp_affine_linear)
__global__ void warp_affine_kernel(const WarpArgs p) {
#pragma clang fp contract(off)
  const long long n = (long long)p.T * p.H * p.W;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(i % p.W), y = (int)((i / p.W) % p.H);
    const long long fb = (i / ((long long)p.W * p.H)) * (long long)p.H * p.W * 3;
    const long long adelta = (long long)rint(p.m[0] * (double)x * 1024.0), bdelta = (long long)rint(p.m[3] * (double)x * 1024.0);
    const long long X0 = (long long)rint((p.m[1] * (double)y + p.m[2]) * 1024.0) + 16, Y0 = (long long)rint((p.m[4] * (double)y + p.m[5]) * 1024.0) + 16;
    const long long X = (X0 + adelta) >> 5, Y = (Y0 + bdelta) >> 5;
    long long sx = X >> 5, sy = Y >> 5;
    sx = sx < -32768 ? -32768 : (sx > 32767 ? 32767 : sx);
    sy = sy < -32768 ? -32768 : (sy > 32767 ? 32767 : sy);
    const int fx = (int)(X & 31), fy = (int)(Y & 31);
    const int w00 = (32 - fx) * (32 - fy) * 32, w01 = fx * (32 - fy) * 32, w10 = (32 - fx) * fy * 32, w11 = fx * fy * 32;
    const bool x0in = sx >= 0 && sx < p.W, x1in = sx + 1 >= 0 && sx + 1 < p.W, y0in = sy >= 0 && sy < p.H, y1in = sy + 1 >= 0 && sy + 1 < p.H;
    for (int c = 0; c < 3; ++c) {
      auto tap = [&](bool in_y, bool in_x, long long yy, long long xx) -> int {
        return (in_y && in_x) ? (int)p.in[fb + (yy * p.W + xx) * 3 + c] : p.fill[c];
      };
      const int acc = tap(y0in, x0in, sy, sx) * w00 + tap(y0in, x1in, sy, sx + 1) * w01 + tap(y1in, x0in, sy + 1, sx) * w10 + tap(y1in, x1in, sy + 1, sx + 1) * w11;
      int v = (acc + (1 << 14)) >> 15;
      v = v < 0 ? 0 : (v > 255 ? 255 : v);
      p.out[fb + ((long long)y * p.W + x) * 3 + c] = (uint8_t)v;
    }
  }
}

// ClipToTensor + Normalize: uint8 [T][H][W][3] -> bf16, element (c, t, y, x) at c * c_stride + t * t_stride + y * W + x
__global__ void u8_normalize_kernel(const uint8_t* __restrict__ frames, int T, int H, int W, float m0, float m1, float m2, float s0, float s1, float s2,
                                    bf16* __restrict__ out, long long c_stride, long long t_stride) {
#pragma clang fp contract(off)
  const long long n = (long long)T * H * W;
  const float mean[3] = {m0, m1, m2}, stdv[3] = {s0, s1, s2};
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const long long t = i / ((long long)H * W), r = i - t * (long long)H * W;
    for (int c = 0; c < 3; ++c) {
      const float q = (float)frames[i * 3 + c];
      out[c * c_stride + t * t_stride + r] = f2bf((q / 255.f - mean[c]) / stdv[c]);
    }
  }
}

}  // namespace

extern "C" int mpv_video_aug_pointwise(uint8_t* frames, int T, int H, int W, int op, double factor, uint64_t* sums_ws, hipStream_t stream) {
  MPV_REQUIRE(frames && T > 0 && H > 0 && W > 0, MPV_E_ARG, "mpv_video_aug_pointwise: bad argument");
  MPV_REQUIRE(op == 0 || op == 1, MPV_E_ARG, "mpv_video_aug_pointwise: op must be 0 (contrast) or 1 (brightness)");
  MPV_REQUIRE(op == 1 || sums_ws, MPV_E_ARG, "mpv_video_aug_pointwise: contrast needs a workspace of 3 * T uint64");
  const long long hw = (long long)H * W;
  unsigned gx = aug_grid(hw) > 256 ? 256 : aug_grid(hw);
  if (op == 0) {
    if (hipMemsetAsync(sums_ws, 0, sizeof(uint64_t) * 3 * T, stream) != hipSuccess) {
      mpv_set_error("mpv_video_aug_pointwise: hipMemsetAsync failed");
      return MPV_E_HIP;
    }
    hipLaunchKernelGGL(channel_sums_kernel, dim3(gx, T), dim3(AUG_BLOCK), 0, stream, (const uint8_t*)frames, hw, (unsigned long long*)sums_ws);
  }
  hipLaunchKernelGGL(pointwise_kernel, dim3(gx, T), dim3(AUG_BLOCK), 0, stream, frames, hw, op, factor, (const unsigned long long*)sums_ws);
  return mpv_check_launch("mpv_video_aug_pointwise");
}

extern "C" int mpv_video_aug_sharpness(const uint8_t* in, uint8_t* out, int T, int H, int W, double factor, hipStream_t stream) {
  MPV_REQUIRE(in && out && in != out && T > 0 && H > 2 && W > 2, MPV_E_ARG, "mpv_video_aug_sharpness: bad argument (out of place, frames of at least 3 x 3)");
  hipLaunchKernelGGL(sharpness_kernel, dim3(aug_grid((long long)T * H * W)), dim3(AUG_BLOCK), 0, stream, in, out, T, H, W, (float)factor,
                     factor == 0.0 ? 1 : 0);
  return mpv_check_launch("mpv_video_aug_sharpness");
}

extern "C" int mpv_video_aug_warp_affine(const uint8_t* in, uint8_t* out, int T, int H, int W, const double* inverse_matrix6, const uint8_t* fill3,
                                        hipStream_t stream) {
  MPV_REQUIRE(in && out && in != out && inverse_matrix6 && fill3 && T > 0 && H > 0 && W > 0, MPV_E_ARG, "mpv_video_aug_warp_affine: bad argument (out of place)");
  WarpArgs a = {};
  a.in = in; a.out = out; a.T = T; a.H = H; a.W = W;
  for (int i = 0; i < 6; ++i) a.m[i] = inverse_matrix6[i];
  for (int c = 0; c < 3; ++c) a.fill[c] = fill3[c];
  hipLaunchKernelGGL(warp_affine_kernel, dim3(aug_grid((long long)T * H * W)), dim3(AUG_BLOCK), 0, stream, a);
  return mpv_check_launch("mpv_video_aug_warp_affine");
}

extern "C" int mpv_video_u8_normalize(const uint8_t* frames, int T, int H, int W, const float* mean3, const float* std3, void* out, int64_t c_stride,
                                     int64_t t_stride, hipStream_t stream) {
  MPV_REQUIRE(frames && mean3 && std3 && out && T > 0 && H > 0 && W > 0, MPV_E_ARG, "mpv_video_u8_normalize: bad argument");
  hipLaunchKernelGGL(u8_normalize_kernel, dim3(aug_grid((long long)T * H * W)), dim3(AUG_BLOCK), 0, stream, frames, T, H, W, mean3[0], mean3[1], mean3[2],
                     std3[0], std3[1], std3[2], (bf16*)out, (long long)c_stride, (long long)t_stride);
  return mpv_check_launch("mpv_video_u8_normalize");
}
