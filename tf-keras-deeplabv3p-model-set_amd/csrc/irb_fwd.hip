// Fused inverted-residual FORWARD (deeplabv3p_mobilenetv2.py:38-74, the high-resolution blocks):
//   z1 = x W1 (expand 1x1, K -> C = 6K)   a1 = act(BN1(z1))   z2 = DepthwiseConv2D(3x3, stride 1 | 2)(a1)
// without ever writing z1 (6x the block input: 406 MB at 16 x 257 x 257 x 96) to HBM:
//   (1) BN1's batch statistics come from ONE pass over the K-channel input: sum x and sum x x^T in float64 on the fp64
//       matrix pipe (dl3p_irb_cov_stats), then mean_z = W^T mean_x, var_z = w^T Cov(x) w per output channel
//       (dl3p_irb_bn_finalize_cov) -- the expand conv is linear and has no bias;
//   (2) dl3p_irb_fwd recomputes the expand tile by tile on v_mfma_f32_16x16x4_f32 with the pixel on the lane, applies
//       BN1 + activation in registers and runs the 3x3 depthwise conv on the result: vertical taps by walking the rows
//       with the partial output rows in registers, horizontal taps by a one-lane shift inside the 16-lane pixel row
//       (DPP), so the expanded activation lives in VGPRs only.  The raw depthwise output and its BatchNorm partial rows
//       leave the kernel, exactly what dl3p_dwconv2d_fwd leaves.
#include "irb_common.h"

typedef double irb_d4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------ covariance statistics
// rows of doubles [K + K*K]: sum_p x[p][k], then sum_p x[p][i] x[p][j] (full symmetric matrix)
#define IRB_COV_WAVES 8
template <int KT>
__global__ __launch_bounds__(64 * IRB_COV_WAVES) void irb_cov_kernel(const float* __restrict__ x, int ldx,
                                                                     const float* __restrict__ xs,
                                                                     const float* __restrict__ xh, int xact, long long M,
                                                                     int K, long long chunk, double* __restrict__ out) {
  constexpr int NT = KT == 1 ? 1 : 3;                 // tiles (0,0) [, (0,1), (1,1)] of the K x K matrix
  __shared__ double sm[IRB_COV_WAVES][NT * 256 + 64 * KT];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, g = lane >> 4;
  const long long p0 = (long long)blockIdx.x * chunk;
  const long long p1 = p0 + chunk < M ? p0 + chunk : M;
  const bool has0 = r < K, has1 = KT == 2 && 16 + r < K;
  const float s0c = (xs && has0) ? xs[r] : 1.f, h0c = (xh && has0) ? xh[r] : 0.f;
  const float s1c = (xs && has1) ? xs[16 + r] : 1.f, h1c = (xh && has1) ? xh[16 + r] : 0.f;
  irb_d4 d00 = {0, 0, 0, 0}, d01 = {0, 0, 0, 0}, d11 = {0, 0, 0, 0};
  double sum0 = 0.0, sum1 = 0.0;
  constexpr int U = 16;
  for (long long base = p0 + wave * 4; base < p1; base += 4 * IRB_COV_WAVES * U) {
    float v0[U], v1[U];
    bool ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long pix = base + (long long)u * 4 * IRB_COV_WAVES + g;
      ok[u] = pix < p1;
      const long long pc = ok[u] ? pix : p0;
      v0[u] = x[(size_t)pc * ldx + (has0 ? r : 0)];
      v1[u] = KT == 2 ? x[(size_t)pc * ldx + (has1 ? 16 + r : 0)] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float a = (ok[u] && has0) ? act_apply(fmaf(v0[u], s0c, h0c), xact) : 0.f;
      const double da = (double)a;
      sum0 += da;
      d00 = __builtin_amdgcn_mfma_f64_16x16x4f64(da, da, d00, 0, 0, 0);
      if constexpr (KT == 2) {
        const float b = (ok[u] && has1) ? act_apply(fmaf(v1[u], s1c, h1c), xact) : 0.f;
        const double db = (double)b;
        sum1 += db;
        d01 = __builtin_amdgcn_mfma_f64_16x16x4f64(da, db, d01, 0, 0, 0);
        d11 = __builtin_amdgcn_mfma_f64_16x16x4f64(db, db, d11, 0, 0, 0);
      }
    }
  }
  // f64 16x16x4 result map: col = lane & 15, row = (lane >> 4) + 4 * reg
  double* my = sm[wave];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    my[(g + 4 * i) * 16 + r] = d00[i];
    if constexpr (KT == 2) {
      my[256 + (g + 4 * i) * 16 + r] = d01[i];
      my[512 + (g + 4 * i) * 16 + r] = d11[i];
    }
  }
  my[NT * 256 + g * 16 * KT + r] = sum0;
  if constexpr (KT == 2) my[NT * 256 + g * 32 + 16 + r] = sum1;
  __syncthreads();
  double* row = out + (size_t)blockIdx.x * (K + K * K);
  for (int e = threadIdx.x; e < K + K * K; e += 64 * IRB_COV_WAVES) {
    double acc = 0.0;
    if (e < K) {
      for (int w = 0; w < IRB_COV_WAVES; ++w)
        for (int gg = 0; gg < 4; ++gg) acc += sm[w][NT * 256 + gg * 16 * KT + e];
    } else {
      const int i = (e - K) / K, jj = (e - K) % K;
      int idx;
      if (i < 16 && jj < 16) idx = i * 16 + jj;
      else if (i < 16) idx = 256 + i * 16 + (jj - 16);
      else if (jj < 16) idx = 256 + jj * 16 + (i - 16);        // tile (1,0) = transpose of (0,1)
      else idx = 512 + (i - 16) * 16 + (jj - 16);
      for (int w = 0; w < IRB_COV_WAVES; ++w) acc += sm[w][idx];
    }
    row[e] = acc;
  }
}

extern "C" int dl3p_irb_cov_rows_max(void) { return DL3P_NUM_CUS; }

extern "C" int dl3p_irb_cov_stats(const float* x, int ldx, const float* in_scale, const float* in_shift, int in_act,
                                  double* cov_rows, int rows_cap, int* rows_out, int M, int K, void* stream) {
  DL3P_CHECK_ARG(x && cov_rows && M > 0, "dl3p_irb_cov_stats: bad arguments");
  DL3P_CHECK_ARG(K == 16 || K == 24 || K == 32, "dl3p_irb_cov_stats: K=%d not in {16, 24, 32}", K);
  DL3P_CHECK_ARG(ldx >= K, "dl3p_irb_cov_stats: ld=%d < K", ldx);
  const long long gran = 4 * IRB_COV_WAVES * 16;
  long long chunk = ceil_div_ll(ceil_div_ll(M, DL3P_NUM_CUS), gran) * gran;
  const int rows = (int)ceil_div_ll(M, chunk);
  DL3P_CHECK_ARG(rows <= rows_cap, "dl3p_irb_cov_stats: %d rows of [K + K*K] doubles needed, room for %d (dl3p_irb_cov_rows_max)", rows,
                 rows_cap);
  if (rows_out) *rows_out = rows;
  if (K == 16)
    dl3p_launch(irb_cov_kernel<1>, dim3(rows), dim3(64 * IRB_COV_WAVES), 0, (hipStream_t)stream, x, ldx, in_scale, in_shift,
                in_act, (long long)M, K, chunk, cov_rows);
  else
    dl3p_launch(irb_cov_kernel<2>, dim3(rows), dim3(64 * IRB_COV_WAVES), 0, (hipStream_t)stream, x, ldx, in_scale, in_shift,
                in_act, (long long)M, K, chunk, cov_rows);
  DL3P_CHECK_LAUNCH("dl3p_irb_cov_stats");
  return DL3P_OK;
}

// sums[e] = sum over the rows, in row order (float64; what SyncBatchNorm all-reduces across ranks)
__global__ __launch_bounds__(1024) void irb_cov_reduce_kernel(const double* __restrict__ rows, int nrows, int n,
                                                              double* __restrict__ sums) {
  __shared__ double sm[64][16];
  const int ex = threadIdx.x & 15, ry = threadIdx.x >> 4;
  const int e = blockIdx.x * 16 + ex;
  double a0 = 0.0, a1 = 0.0;
  if (e < n) {
    int r = ry;
    for (; r + 64 < nrows; r += 128) {
      a0 += rows[(size_t)r * n + e];
      a1 += rows[(size_t)(r + 64) * n + e];
    }
    if (r < nrows) a0 += rows[(size_t)r * n + e];
  }
  sm[ry][ex] = a0 + a1;
  __syncthreads();
  if (ry == 0 && e < n) {
    double acc = 0.0;
#pragma unroll 8
    for (int qq = 0; qq < 64; ++qq) acc += sm[qq][ex];
    sums[e] = acc;
  }
}

extern "C" int dl3p_irb_cov_reduce(const double* cov_rows, int rows, int K, double* sums, void* stream) {
  DL3P_CHECK_ARG(cov_rows && sums && rows > 0 && K > 0, "dl3p_irb_cov_reduce: bad arguments");
  const int n = K + K * K;
  hipLaunchKernelGGL(irb_cov_reduce_kernel, dim3(ceil_div(n, 16)), dim3(1024), 0, (hipStream_t)stream, cov_rows, rows, n, sums);
  DL3P_CHECK_LAUNCH("dl3p_irb_cov_reduce");
  return DL3P_OK;
}

// BatchNorm coefficients of the expand output from the statistics of its INPUT: z = W^T x, so
// mean_z[c] = sum_k W[k][c] mean_x[k], var_z[c] = sum_ij W[i][c] Cov_x[i][j] W[j][c]; everything in float64.
// One workgroup per 8 output channels, 32 lanes per channel: lane i forms w_i * (Cov w)_i, the 32 lanes are added by shuffles.
__global__ __launch_bounds__(256) void irb_bn_finalize_cov_kernel(const double* __restrict__ sums, const float* __restrict__ w1,
                                                                  int K, int C, double count, const float* gamma,
                                                                  const float* beta, float eps, float momentum,
                                                                  float* moving_mean, float* moving_var, int update_moving,
                                                                  float* scale, float* shift, float* save_mean,
                                                                  float* save_invstd) {
  __shared__ double mu[32], cov[32 * 33];
  __shared__ double wsm[8][32];
  for (int k = threadIdx.x; k < 32; k += 256) mu[k] = k < K ? sums[k] / count : 0.0;
  __syncthreads();
  for (int e = threadIdx.x; e < 32 * 32; e += 256) {
    const int i = e >> 5, jj = e & 31;
    cov[i * 33 + jj] = (i < K && jj < K) ? sums[K + i * K + jj] / count - mu[i] * mu[jj] : 0.0;
  }
  const int cl = threadIdx.x >> 5, i = threadIdx.x & 31;
  const int c = blockIdx.x * 8 + cl;
  const double wi = (c < C && i < K) ? (double)w1[(size_t)i * C + c] : 0.0;
  wsm[cl][i] = wi;
  __syncthreads();
  double t = 0.0;
#pragma unroll 8
  for (int jj = 0; jj < 32; ++jj) t += cov[i * 33 + jj] * wsm[cl][jj];
  double var = wi * t, mean = wi * mu[i];
#pragma unroll
  for (int m = 16; m >= 1; m >>= 1) {
    var += __shfl_xor(var, m, 32);
    mean += __shfl_xor(mean, m, 32);
  }
  if (i != 0 || c >= C) return;
  if (var < 0.0) var = 0.0;
  const float invstd = (float)(1.0 / sqrt(var + (double)eps));
  const float sc = gamma[c] * invstd;
  scale[c] = sc;
  shift[c] = beta[c] - (float)mean * sc;
  save_mean[c] = (float)mean;
  save_invstd[c] = invstd;
  if (update_moving) {      // same rule as dl3p_bn_finalize (1 biased, 2 Bessel-corrected)
    const double mvar = (update_moving == 2 && count > 1.0) ? var * (count / (count - 1.0)) : var;
    moving_mean[c] = moving_mean[c] * momentum + (float)mean * (1.f - momentum);
    moving_var[c] = moving_var[c] * momentum + (float)mvar * (1.f - momentum);
  }
}

extern "C" int dl3p_irb_bn_finalize_cov(const double* sums, const float* w1, int K, int C, double count, const float* gamma,
                                        const float* beta, float eps, float momentum, float* moving_mean, float* moving_var,
                                        int update_moving, float* scale, float* shift, float* save_mean, float* save_invstd,
                                        void* stream) {
  DL3P_CHECK_ARG(sums && w1 && gamma && beta && scale && shift && save_mean && save_invstd && count > 0 && C > 0,
                 "dl3p_irb_bn_finalize_cov: bad arguments");
  DL3P_CHECK_ARG(K > 0 && K <= 32, "dl3p_irb_bn_finalize_cov: K=%d", K);
  DL3P_CHECK_ARG(!update_moving || (moving_mean && moving_var), "dl3p_irb_bn_finalize_cov: moving stats missing");
  hipLaunchKernelGGL(irb_bn_finalize_cov_kernel, dim3(ceil_div(C, 8)), dim3(256), 0, (hipStream_t)stream, sums, w1, K, C,
                     count, gamma, beta, eps, momentum, moving_mean, moving_var, update_moving, scale, shift, save_mean,
                     save_invstd);
  DL3P_CHECK_LAUNCH("dl3p_irb_bn_finalize_cov");
  return DL3P_OK;
}

// ------------------------------------------------------------------------------------------ fused forward
// One wave = one (image, column segment, band of output rows, group of CT 16-channel tiles).
// Stride 2: lane j of the segment owns output column ox0 + j; tile T0 holds the expanded activation at input column
// 2*ox - pad_l (tap kx = 0), T1 at + 1 (kx = 1); tap kx = 2 is T0 of lane j + 1, so 15 of the 16 lanes produce an output.
// Stride 1: lane j holds input column base + j and produces the output centred there from lanes j - 1, j, j + 1: 14 of 16.
// Rows: every input row is expanded once and added into the one, two (stride 2) or three (stride 1) output rows it feeds;
// the block input is fetched one loop iteration ahead into alternating register sets.
//
// Measured (scripts/micro/irb_pmc.sh): the kernel is bound by VECTOR-INSTRUCTION ISSUE, not by its MFMAs (15 of 80 us) and not by
// memory (3 of 80 us): every instruction of a wave holds the SIMD's issue for ~4 cycles and the waves of a SIMD do not overlap
// each other's vector work.  So the per-element instruction count is the cost, and everything linear is folded into the MFMA:
// with a plain affine in front of the block (XACT = NONE) the expand kernel is pre-multiplied by the input scale and the
// BatchNorm scale, and the MFMA chain STARTS from the folded shift (shift + scale * W^T in_shift), so the pre-activation u
// comes out of the matrix pipe and the vector pipe is left with: v_med3 (ReLU6), the nine depthwise taps, a DPP move per shifted
// tap, the statistics.  Wave-uniform quantities are forced into SGPRs (readfirstlane) so that addresses are scalar arithmetic.
template <int K, int CT, int S, int ACT, int XACT>
__global__ __launch_bounds__(256) void irb_fwd_kernel(IrbParams p) {
  constexpr bool FOLD = XACT == DL3P_ACT_NONE;
  constexpr int KQ = K / 4;
  extern __shared__ float sm[];                      // depthwise kernel [9][C], folded shift [C]
  const int C = p.C;
  float* s_bias = sm + 9 * C;
  for (int i = threadIdx.x; i < 9 * C; i += 256) sm[i] = p.wdw[i];
  for (int c = threadIdx.x; c < C; c += 256) {
    float b = p.h1[c];
    if (FOLD && p.xh) {
      float acc = 0.f;
      for (int k = 0; k < K; ++k) acc = fmaf(p.w1[(size_t)k * C + c], p.xh[k], acc);
      b = fmaf(p.s1[c], acc, b);
    }
    s_bias[c] = b;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 15, q = lane >> 4;
  const int unit = irb_wg_index(blockIdx.x, gridDim.x) * 4 + wave;      // (scalar)
  if (unit >= p.units) return;
  int rr = unit;
  const int cg = rr % p.ncg; rr /= p.ncg;
  const int band = rr % p.nband; rr /= p.nband;
  const int seg = rr % p.nseg;
  const int n = rr / p.nseg;
  const int row_id = unit / p.ncg;
  const int H = p.H, W = p.W, Ho = p.Ho, Wo = p.Wo;
  const int c0 = cg * CT * 16;

  float wf[CT][KQ], xs[KQ], xh[KQ];
  float4 sc[CT], sh[CT], wt[CT][9];
#pragma unroll
  for (int s = 0; s < KQ; ++s) {
    xs[s] = p.xs ? p.xs[q * KQ + s] : 1.f;
    xh[s] = p.xh ? p.xh[q * KQ + s] : 0.f;
  }
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    const float scj = p.s1[c0 + 16 * ct + j];
#pragma unroll
    for (int s = 0; s < KQ; ++s) {
      const float w = p.w1[(size_t)(q * KQ + s) * C + c0 + 16 * ct + j];
      wf[ct][s] = FOLD ? w * xs[s] * scj : w;
    }
    sc[ct] = ld4(p.s1 + c0 + 16 * ct + 4 * q);
    sh[ct] = ld4(s_bias + c0 + 16 * ct + 4 * q);       // FOLD: the complete shift of u; else the BatchNorm shift
#pragma unroll
    for (int t = 0; t < 9; ++t) wt[ct][t] = ld4(sm + t * C + c0 + 16 * ct + 4 * q);
  }
  const int xact = p.xact, act1 = p.act1;
  const int oy0 = band * p.band;
  const int oy1 = oy0 + p.band < Ho ? oy0 + p.band : Ho;

  // column geometry
  int ox, ixa, ixb = 0;
  bool out_ok;
  if constexpr (S == 2) {
    ox = seg * 15 + j;
    ixa = 2 * ox - p.pad_l;
    ixb = ixa + 1;
    out_ok = j < 15 && ox < Wo;
  } else {
    ox = seg * 14 + j - 1;
    ixa = seg * 14 - p.pad_l + j;
    out_ok = j >= 1 && j <= 14 && ox < Wo;
  }
  const bool va = ixa >= 0 && ixa < W, vb = ixb >= 0 && ixb < W;
  // does any lane of this segment's tiles lie outside the image?  (wave-uniform: the first and the last segment only)
  const bool edge = __any(!va) || (S == 2 && __any(!vb));
  const int ixac = ixa < 0 ? 0 : (ixa >= W ? W - 1 : ixa), ixbc = ixb < 0 ? 0 : (ixb >= W ? W - 1 : ixb);
  const float* xn_ = p.x + (size_t)n * H * W * p.ldx;
  const int offa = ixac * p.ldx + q * KQ, offb = ixbc * p.ldx + q * KQ;      // (per lane; the row base below is scalar)
  const size_t rowpitch = (size_t)W * p.ldx;

  auto load_row = [&](int iy, float (&ra)[KQ], float (&rb)[KQ]) {
    const int iyc = iy < 0 ? 0 : (iy >= H ? H - 1 : iy);
    const float* rp = xn_ + (size_t)iyc * rowpitch;
    irb_load_x<K>(rp + offa, ra);
    if constexpr (S == 2) irb_load_x<K>(rp + offb, rb);
  };
  // pre-activation tiles u[ct][t] of NT pixel tiles: the MFMA chains of all tiles interleaved, one k-step of every tile after
  // the other (no MFMA waits for its predecessor: 40-cycle dependent latency against a 32-cycle issue)
  auto expand4 = [&](float (&x0)[KQ], float (&x1)[KQ], float (&x2)[KQ], float (&x3)[KQ], float4 (&u)[CT][4]) {
    if constexpr (!FOLD) {
      irb_prologue<K, XACT>(x0, xs, xh, xact); irb_prologue<K, XACT>(x1, xs, xh, xact);
      irb_prologue<K, XACT>(x2, xs, xh, xact); irb_prologue<K, XACT>(x3, xs, xh, xact);
    }
    irb_f4 z[CT][4];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int t = 0; t < 4; ++t)
        z[ct][t] = FOLD ? (irb_f4){sh[ct].x, sh[ct].y, sh[ct].z, sh[ct].w} : (irb_f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < KQ; ++s)
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        z[ct][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[ct][s], x0[s], z[ct][0], 0, 0, 0);
        z[ct][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[ct][s], x1[s], z[ct][1], 0, 0, 0);
        z[ct][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[ct][s], x2[s], z[ct][2], 0, 0, 0);
        z[ct][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[ct][s], x3[s], z[ct][3], 0, 0, 0);
      }
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int t = 0; t < 4; ++t)
        u[ct][t] = FOLD ? irb_f4_to_float4(z[ct][t]) : fma4(irb_f4_to_float4(z[ct][t]), sc[ct], sh[ct]);
  };
  auto expand1 = [&](float (&x0)[KQ], float4 (&u)[CT]) {
    if constexpr (!FOLD) irb_prologue<K, XACT>(x0, xs, xh, xact);
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      irb_f4 z = FOLD ? (irb_f4){sh[ct].x, sh[ct].y, sh[ct].z, sh[ct].w} : (irb_f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < KQ; ++s) z = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[ct][s], x0[s], z, 0, 0, 0);
      u[ct] = FOLD ? irb_f4_to_float4(z) : fma4(irb_f4_to_float4(z), sc[ct], sh[ct]);
    }
  };

  float4 s1[CT], s2[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) { s1[ct] = zero4(); s2[ct] = zero4(); }
  float* ybase = p.y + ((size_t)n * Ho * Wo + (out_ok ? ox : 0)) * p.ldy + c0 + 4 * q;
  auto emit = [&](int o, const float4 (&acc)[CT]) {
    if (out_ok) {
      float* yp = ybase + (size_t)o * Wo * p.ldy;
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        st4(yp + 16 * ct, acc[ct]);
        s1[ct] = add4(s1[ct], acc[ct]);
        s2[ct] = fma4(acc[ct], acc[ct], s2[ct]);
      }
    }
  };

  if constexpr (S == 2) {
    // two register sets for the block input, used alternately (the loop is unrolled by two): the set an iteration fetches into is
    // the one the NEXT iteration computes from, so a fetch has a whole iteration to land and no register is copied
    float A0[KQ], A1[KQ], A2[KQ], A3[KQ], B0[KQ], B1[KQ], B2[KQ], B3[KQ];
    float4 cur[CT], nxt[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) cur[ct] = zero4();
    int iy = 2 * oy0 - p.pad_t;
    load_row(iy, B0, B1);
    load_row(iy + 1, A0, A1);
    load_row(iy + 2, A2, A3);
    if (iy >= 0 && iy < H) {                          // first even row: ky = 0 of the band's first output row
      float4 ua[CT], ub[CT];
      expand1(B0, ua);
      expand1(B1, ub);
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        const float4 ta = irb_sel4(va, irb_act4<ACT>(ua[ct], act1)), tb = irb_sel4(vb, irb_act4<ACT>(ub[ct], act1));
        const float4 tc = irb_from_next4(ta);
        cur[ct] = mul4(wt[ct][0], ta);
        cur[ct] = fma4(wt[ct][1], tb, cur[ct]);
        cur[ct] = fma4(wt[ct][2], tc, cur[ct]);
      }
    }
    // one output row: its odd input row (tiles xa0, xb0) and the even row below it (xa1, xb1)
    auto step = [&](int o, float (&xa0)[KQ], float (&xb0)[KQ], float (&xa1)[KQ], float (&xb1)[KQ], float (&na0)[KQ],
                    float (&nb0)[KQ], float (&na1)[KQ], float (&nb1)[KQ]) {
      iy += 2;                                        // iy = the even row that closes output row o
      load_row(iy + 1, na0, nb0);
      load_row(iy + 2, na1, nb1);
      float4 u[CT][4];
      expand4(xa0, xb0, xa1, xb1, u);
      const bool vro = iy - 1 < H, vre = iy < H;      // (wave-uniform; false only in the band that ends at the bottom border)
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        float4 t0 = irb_act4<ACT>(u[ct][0], act1), t1 = irb_act4<ACT>(u[ct][1], act1);
        float4 t2 = irb_act4<ACT>(u[ct][2], act1), t3 = irb_act4<ACT>(u[ct][3], act1);
        if (edge || !vro || !vre) {
          t0 = irb_sel4(va && vro, t0); t1 = irb_sel4(vb && vro, t1);
          t2 = irb_sel4(va && vre, t2); t3 = irb_sel4(vb && vre, t3);
        }
        const float4 s0 = irb_from_next4(t0), s2t = irb_from_next4(t2);
        float4 c = cur[ct];
        c = fma4(wt[ct][3], t0, c); c = fma4(wt[ct][4], t1, c); c = fma4(wt[ct][5], s0, c);
        c = fma4(wt[ct][6], t2, c); c = fma4(wt[ct][7], t3, c); c = fma4(wt[ct][8], s2t, c);
        cur[ct] = c;
        float4 nx = mul4(wt[ct][0], t2);
        nx = fma4(wt[ct][1], t3, nx); nx = fma4(wt[ct][2], s2t, nx);
        nxt[ct] = nx;
      }
      emit(o, cur);
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) cur[ct] = nxt[ct];
    };
    int o = oy0;
    for (; o + 1 < oy1; o += 2) {
      step(o, A0, A1, A2, A3, B0, B1, B2, B3);
      step(o + 1, B0, B1, B2, B3, A0, A1, A2, A3);
    }
    if (o < oy1) step(o, A0, A1, A2, A3, B0, B1, B2, B3);
  } else {
    // three register sets in rotation (the loop is unrolled by three): iteration t computes from set t % 3 and fetches row t + 2
    float X0[KQ], X1[KQ], X2[KQ], dummy[KQ];
    float4 a2[CT], a1[CT], a0[CT];                    // output rows t - 2, t - 1, t
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) { a2[ct] = zero4(); a1[ct] = zero4(); }
    const int nrows = oy1 - oy0;
    int iy = oy0 - p.pad_t;
    load_row(iy, X0, dummy);
    load_row(iy + 1, X1, dummy);
    auto step = [&](int t, float (&xc)[KQ], float (&xn)[KQ]) {
      load_row(iy + 2, xn, dummy);
      if (iy >= 0 && iy < H) {
        float4 u[CT];
        expand1(xc, u);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          float4 ta = irb_act4<ACT>(u[ct], act1);
          if (edge) ta = irb_sel4(va, ta);
          const float4 tl = irb_from_prev4(ta), tr = irb_from_next4(ta);
          a2[ct] = fma4(wt[ct][6], tl, a2[ct]);
          a2[ct] = fma4(wt[ct][7], ta, a2[ct]);
          a2[ct] = fma4(wt[ct][8], tr, a2[ct]);
          a1[ct] = fma4(wt[ct][3], tl, a1[ct]);
          a1[ct] = fma4(wt[ct][4], ta, a1[ct]);
          a1[ct] = fma4(wt[ct][5], tr, a1[ct]);
          a0[ct] = mul4(wt[ct][0], tl);
          a0[ct] = fma4(wt[ct][1], ta, a0[ct]);
          a0[ct] = fma4(wt[ct][2], tr, a0[ct]);
        }
      } else {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) a0[ct] = zero4();
      }
      if (t >= 2) emit(oy0 + t - 2, a2);
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) { a2[ct] = a1[ct]; a1[ct] = a0[ct]; }
      ++iy;
    };
    int t = 0;
    for (; t + 2 < nrows + 2; t += 3) {
      step(t, X0, X2);
      step(t + 1, X1, X0);
      step(t + 2, X2, X1);
    }
    if (t < nrows + 2) { step(t, X0, X2); ++t; }
    if (t < nrows + 2) { step(t, X1, X0); ++t; }
  }

  if (p.partials) {
    float* prow = p.partials + (size_t)row_id * 2 * C + c0 + 4 * q;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      const float4 a = irb_row_sum4(s1[ct]), b = irb_row_sum4(s2[ct]);
      if (j == 0) {
        st4(prow + 16 * ct, a);
        st4(prow + C + 16 * ct, b);
      }
    }
  }
}

static int irb_plan(IrbParams& p, int N, int Ho, int Wo, int C, int S, int CT, int lanes_out, int want_waves) {
  p.nseg = ceil_div(Wo, lanes_out);
  p.ncg = C / (16 * CT);
  long long per = (long long)N * p.nseg * p.ncg;
  int nband = (int)ceil_div_ll(want_waves, per);
  if (nband < 1) nband = 1;
  const int max_by_rows = DL3P_MAX_STAT_ROWS / (N * p.nseg);
  if (nband > max_by_rows) nband = max_by_rows;
  if (nband < 1) return -1;
  int band = ceil_div(Ho, nband);
  const int min_band = 4;
  if (band < min_band) band = Ho < min_band ? Ho : min_band;
  nband = ceil_div(Ho, band);
  if ((long long)N * p.nseg * nband > DL3P_MAX_STAT_ROWS) return -1;
  p.band = band;
  p.nband = nband;
  p.units = (int)(per * nband);
  return N * p.nseg * nband;
}

static int g_irb_ct = 0, g_irb_waves = 0;
extern "C" int dl3p_irb_set_plan(int ct, int want_waves) {
  // (the forward is instantiated for 1 .. 3 channel tiles per wave)
  DL3P_CHECK_ARG(ct >= 0 && ct <= 3 && want_waves >= 0, "dl3p_irb_set_plan: ct=%d not in 0..3 or want_waves=%d < 0", ct, want_waves);
  g_irb_ct = ct;
  g_irb_waves = want_waves;
  return DL3P_OK;
}
int dl3p_irb_fwd_plan_knob(int which) { return which == 0 ? g_irb_ct : g_irb_waves; }

static int irb_pick_ct(int C) {
  if (g_irb_ct > 0 && C % (16 * g_irb_ct) == 0) return g_irb_ct;
  // measured (scripts/micro/irb_bench.py): two tiles per wave where C allows it (257 x 257 x 16 -> 96: 62 us against 73 with one, 118 with
  // three at one wave per SIMD); 144 channels = 9 tiles stay at one
  return C % 32 == 0 ? 2 : 1;
}

extern "C" int dl3p_irb_supported(int N, int H, int W, int K, int C, int k, int stride, int rate, int pad_t, int pad_l, int Ho,
                                  int Wo) {
  if (!(K == 16 || K == 24 || K == 32) || C % 16 != 0 || C < 16 || C > 1024) return 0;
  if (k != 3 || rate != 1 || (stride != 1 && stride != 2)) return 0;
  if (pad_t < 0 || pad_t > 1 || pad_l < 0 || pad_l > 1) return 0;
  // the outputs must be what the taps reach: TF SAME or explicit padding of at most one pixel in front
  if ((Ho - 1) * stride - pad_t + 2 > H || (Wo - 1) * stride - pad_l + 2 > W) return 0;
  if ((long long)N * H * W >= (1ll << 31) / 256) return 0;
  IrbParams p = {};
  if (irb_plan(p, N, Ho, Wo, C, stride, irb_pick_ct(C), stride == 2 ? 15 : 14, 4096) < 0) return 0;
  return 1;
}

template <int K, int CT>
static void irb_fwd_launch(const IrbParams& p, int S, hipStream_t st) {
  const int wgs = ceil_div(p.units, 4);
  const int grid = ceil_div(wgs, 8) * 8;
  const size_t shm = (size_t)10 * p.C * sizeof(float);
  // the compile-time activation pair of every MobileNetV2 block (ReLU6 behind the expand BatchNorm, a plain affine in front of the
  // block); anything else takes the run-time evaluation
  const bool fast = p.act1 == DL3P_ACT_RELU6 && p.xact == DL3P_ACT_NONE;
  if (S == 2) {
    if (fast) dl3p_launch(irb_fwd_kernel<K, CT, 2, DL3P_ACT_RELU6, DL3P_ACT_NONE>, dim3(grid), dim3(256), shm, st, p);
    else dl3p_launch(irb_fwd_kernel<K, CT, 2, -1, -1>, dim3(grid), dim3(256), shm, st, p);
  } else {
    if (fast) dl3p_launch(irb_fwd_kernel<K, CT, 1, DL3P_ACT_RELU6, DL3P_ACT_NONE>, dim3(grid), dim3(256), shm, st, p);
    else dl3p_launch(irb_fwd_kernel<K, CT, 1, -1, -1>, dim3(grid), dim3(256), shm, st, p);
  }
}

extern "C" int dl3p_irb_fwd(const float* x, int ldx, const float* in_scale, const float* in_shift, int in_act, const float* w1,
                            const float* bn_scale, const float* bn_shift, int bn_act, const float* wdw, float* y, int ldy,
                            float* stat_partials, int* rows_out, int N, int H, int W, int K, int C, int stride, int pad_t,
                            int pad_l, int Ho, int Wo, void* stream) {
  DL3P_CHECK_ARG(x && w1 && bn_scale && bn_shift && wdw && y, "dl3p_irb_fwd: null pointer");
  DL3P_CHECK_ARG(dl3p_irb_supported(N, H, W, K, C, 3, stride, 1, pad_t, pad_l, Ho, Wo),
                 "dl3p_irb_fwd: unsupported shape N=%d H=%d W=%d K=%d C=%d stride=%d", N, H, W, K, C, stride);
  DL3P_CHECK_ARG(ldx >= K && ldx % 2 == 0 && (K != 16 && K != 32 || (ldx % 4 == 0 && aligned16(x))) && ldy >= C && ldy % 4 == 0 &&
                     aligned16(y),
                 "dl3p_irb_fwd: bad layout (ldx=%d, ldy=%d)", ldx, ldy);
  IrbParams p = {};
  p.x = x; p.ldx = ldx; p.xs = in_scale; p.xh = in_shift; p.xact = in_act; p.w1 = w1; p.s1 = bn_scale; p.h1 = bn_shift;
  p.act1 = bn_act; p.wdw = wdw; p.y = y; p.ldy = ldy; p.partials = stat_partials;
  p.N = N; p.H = H; p.W = W; p.C = C; p.Ho = Ho; p.Wo = Wo; p.pad_t = pad_t; p.pad_l = pad_l;
  const int CT = irb_pick_ct(C);
  const int rows = irb_plan(p, N, Ho, Wo, C, stride, CT, stride == 2 ? 15 : 14, g_irb_waves > 0 ? g_irb_waves : 4096);
  if (rows_out) *rows_out = rows;
  hipStream_t st = (hipStream_t)stream;
  bool launched = false;
#define IRB_FWD_CASE(KK, CC) if (K == KK && CT == CC) { irb_fwd_launch<KK, CC>(p, stride, st); launched = true; }
  IRB_FWD_CASE(16, 1); IRB_FWD_CASE(16, 2); IRB_FWD_CASE(16, 3);
  IRB_FWD_CASE(24, 1); IRB_FWD_CASE(24, 2); IRB_FWD_CASE(24, 3);
  IRB_FWD_CASE(32, 1); IRB_FWD_CASE(32, 2); IRB_FWD_CASE(32, 3);
#undef IRB_FWD_CASE
  DL3P_CHECK_ARG(launched, "dl3p_irb_fwd: no kernel for K=%d with %d channel tiles per wave", K, CT);
  DL3P_CHECK_LAUNCH("dl3p_irb_fwd");
  return DL3P_OK;
}

// lane-shift self test: out[l] = value of lane l+1 (first 64) / lane l-1 (next 64) of v = lane id, as the kernels see it
__global__ void irb_selftest_kernel(float* out) {
  const float v = (float)threadIdx.x;
  out[threadIdx.x] = irb_from_next(v);
  out[64 + threadIdx.x] = irb_from_prev(v);
}
extern "C" int dl3p_irb_selftest(float* out128, void* stream) {
  DL3P_CHECK_ARG(out128, "dl3p_irb_selftest: null pointer");
  hipLaunchKernelGGL(irb_selftest_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out128);
  DL3P_CHECK_LAUNCH("dl3p_irb_selftest");
  return DL3P_OK;
}
