// gemm_args.h -- argument block shared by the bf16 GEMM kernels (gemm.hip: 128x128 two-workgroups-per-CU kernel,
// gemm256.hip: 256x256 eight-phase kernel).  Internal to the library.
#pragma once
#include "mpv_common.h"

struct GemmArgs {
  const bf16* A;
  const bf16* B;
  void* C;
  int M, N, K;
  long long lda, ldb, ldc;
  RowMap amap, cmap, kmap;
  uint32_t a_bytes, b_bytes;
  const bf16* bias;
  int act;
  bf16* preact;
  const bf16* residual;
  long long ldr;
  const bf16* actz;
  long long ldz;
  int act_bwd;
  int preact_deriv;     // preact receives act'(z) instead of z (mpv.h: preact_deriv / MPV_ACT_DERIV)
  float drop_scale;
  uint32_t drop_thr;
  uint64_t seed, drop_offset;
  const float* alpha_dev;
  float alpha;
  int out_f32;
  int accumulate;
  int k_per_split;
  int tiles_n, tiles_m;
  int nwg, splits;
  int m_base;           // gemm256: first row of the row band this launch covers (M = one past its last row)
  int tile_rows;        // gemm256: 0 = pick 256 / 192 / 160 tile rows per problem, else pinned (tile_hint 192 / 160 / 256)
  int gm;               // gemm256: m-tiles per n-tile in an XCD's tile walk (0 -> default)
  int keep_c;           // gemm256: plain instead of non-temporal output stores (mpv.h: keep_output)
  int batch;            // gemm.hip: > 0 = the operand pointers come from the launch's pointer table, problem blockIdx.y (mpv_gemm_bf16_batched)
  float colscale;       // columns n < colscale_cols: bf16(bf16(acc + bias) * colscale) (mpv.h: colscale / colscale_cols; plain epilogue only)
  int colscale_cols;
  bf16* tap_out;        // rows m % tap_group == 0 also store bf16(acc * alpha + bias) at tap_out[(m / tap_group) * N + n]
  int tap_group;
  float* colsum_part;   // wgrad only: [splits][M] fp32 partial column sums of the A operand (bias gradient)
  // tail split: the last `tiles % 512` tiles (a mostly empty final round of the 512 resident slots) are cut
  // tail_g ways along K; partial tiles meet in tail_ws and the last workgroup to arrive finishes the tile
  int tail_start, tail_g, tail_steps;
  float* tail_ws;
  unsigned* tail_cnt;
};

// gemm256.hip: true if the 256x256 eight-phase kernel takes this problem (it launched it on `stream`), false if the
// caller should fall back to the 128x128 kernel.
bool mpv_gemm256_try_launch(const GemmArgs& g, int transA, int transB, hipStream_t stream);
