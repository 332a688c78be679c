// Shared by pwconv.hip (fp32 MFMA kernels) and pw_split.hip (the fp32-accurate split-bf16 kernels): the argument block of a
// pointwise / implicit GEMM launch.
#pragma once
#include "common.h"

struct GemmParams {
  const float* A; int lda;
  const float* scale; const float* shift; int act;
  const float* B; int ldb;
  const float* bias;
  float* Y; int ldy;
  float* partials;
  int M, K, N;
  int accumulate;
  int num_m_tiles;
  // split-K forward (dl3p_pwconv_fwd_wt_splitk; <B_KN = false, STATS = false> instantiations only): gridDim.y = column blocks x ksplit,
  // workgroup slice z multiplies columns [z kchunk, (z + 1) kchunk) of A with the same rows of B^T and writes slab z of Y = [ksplit][M][ldy]
  int ksplit, kchunk;
  int stagger;         // pw_split.hip's ablation build only (-DDL3P_SB_ABLATE, scripts/micro/sb_ablate.sh)
  int b_kn;            // pw_small_kernel: B stored [K][N] (forward) or [N][K] (data gradient)
  // fused BatchNorm-backward statistics (data gradient writing the gradient of a BN+activation output): with bb_z
  // set, the per-channel partials are (sum g', sum g' * xhat), g' = y * act'(z*scale+shift), xhat = (z-mean)*invstd,
  // i.e. exactly what dl3p_bn_bwd_reduce would compute from the finished gradient y in a separate pass
  const float* bb_z; int bb_ldz;
  const float* bb_scale; const float* bb_shift; const float* bb_mean; const float* bb_invstd; int bb_act;
  // implicit-GEMM gather of the A operand (dense k x k convolutions without a patch matrix in HBM; GA instantiations
  // only).  Row m = (n, y, x) over g_RH x g_RW; column k = tap * g_C + c; the element is the source tensor
  // [N][g_SH][g_SW][lda] at (sy, sx) = ((y * g_mul + g_ay + ky * g_d) >> g_shift, likewise x with g_ax), zero when that is
  // outside the source or (data gradient of a strided conv) not a multiple of the stride.
  //   forward:        rows = output pixels, source = input,   g_mul = stride, g_ay = -pad_t, g_d = +rate, g_shift = 0
  //   data gradient:  rows = input pixels,  source = dy,      g_mul = 1,      g_ay = +pad_t, g_d = -rate, g_shift = log2(stride)
  int g_RH, g_RW, g_SH, g_SW, g_C, g_kw, g_mul, g_ay, g_ax, g_d, g_shift;
  uint32_t g_cmagic;   // floor(2^32 / g_C) + 1: tap = umulhi(k, g_cmagic), exact for k < 2^16 (hosts check K)
  int g_kwmagic;       // 65536 / g_kw + 1: ky = (tap * g_kwmagic) >> 16 for tap < 64
  // pw_split.hip: the B operand pre-split into three bf16 planes (dl3p_split_bf16x3), [plane][Nout rows][bsp_pitch] with the
  // reduction index contiguous and zero-padded to a multiple of 32; bsp_plane = elements per plane
  const unsigned short* Bsp; int bsp_pitch; long long bsp_plane;
  // pw_split_rs.hip, FOLD instantiations (data gradient): the A operand is the BatchNorm-backward apply of (A = g, f_z) formed
  // while the row tile is staged -- dz = c0 (g act'(z scale + shift) - c1 - xhat c2), dl3p_bn_bwd_apply's arithmetic as
  // A g act' - C z + D -- and written once to f_dz (the weight gradient that follows reads it there)
  const float* f_z; int f_ldz; const float* f_scale; const float* f_shift; const float* f_mean; const float* f_invstd;
  const float* f_coef; int f_act; float* f_dz; int f_lddz;
};
