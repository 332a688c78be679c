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
  constexpr int U = 4;
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

extern "C" int dl3p_irb_cov_stats(const float* x, int ldx, const float* in_scale, const float* in_shift, int in_act,
                                  double* cov_rows, int* rows_out, int M, int K, void* stream) {
  DL3P_CHECK_ARG(x && cov_rows && M > 0, "dl3p_irb_cov_stats: bad arguments");
  DL3P_CHECK_ARG(K == 16 || K == 24 || K == 32, "dl3p_irb_cov_stats: K=%d not in {16, 24, 32}", K);
  DL3P_CHECK_ARG(ldx >= K, "dl3p_irb_cov_stats: ld=%d < K", ldx);
  const long long gran = 4 * IRB_COV_WAVES * 4;
  long long chunk = ceil_div_ll(ceil_div_ll(M, DL3P_NUM_CUS), gran) * gran;
  const int rows = (int)ceil_div_ll(M, chunk);
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
// mean_z[c] = sum_k W[k][c] mean_x[k], var_z[c] = sum_ij W[i][c] Cov_x[i][j] W[j][c]; everything in float64
__global__ __launch_bounds__(256) void irb_bn_finalize_cov_kernel(const double* __restrict__ sums, const float* __restrict__ w1,
                                                                  int K, int C, double count, const float* gamma,
                                                                  const float* beta, float eps, float momentum,
                                                                  float* moving_mean, float* moving_var, int update_moving,
                                                                  float* scale, float* shift, float* save_mean,
                                                                  float* save_invstd) {
  __shared__ double mu[32], cov[32 * 32];
  __shared__ float wsm[32][256];                     // this thread's kernel column (no per-thread array: no scratch)
  for (int k = threadIdx.x; k < K; k += 256) mu[k] = sums[k] / count;
  __syncthreads();
  for (int e = threadIdx.x; e < K * K; e += 256) cov[e] = sums[K + e] / count - mu[e / K] * mu[e % K];
  __syncthreads();
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  double mean = 0.0;
  for (int k = 0; k < K; ++k) {
    const float wk = w1[(size_t)k * C + c];
    wsm[k][threadIdx.x] = wk;
    mean += (double)wk * mu[k];
  }
  double var = 0.0;
  for (int i = 0; i < K; ++i) {
    double t = 0.0;
    for (int jj = 0; jj < K; ++jj) t += cov[i * K + jj] * (double)wsm[jj][threadIdx.x];
    var += (double)wsm[i][threadIdx.x] * t;
  }
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
  hipLaunchKernelGGL(irb_bn_finalize_cov_kernel, dim3(ceil_div(C, 256)), dim3(256), 0, (hipStream_t)stream, sums, w1, K, C,
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
// Rows: every input row is expanded once and added into the one, two (stride 2) or three (stride 1) output rows it feeds.
template <int K, int CT, int S>
__global__ __launch_bounds__(256) void irb_fwd_kernel(IrbParams p) {
  extern __shared__ float sm[];                      // depthwise kernel [9][C]
  for (int i = threadIdx.x; i < 9 * p.C; i += 256) sm[i] = p.wdw[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, q = lane >> 4;
  const int unit = irb_wg_index(blockIdx.x, gridDim.x) * 4 + wave;
  if (unit >= p.units) return;
  int rr = unit;
  const int cg = rr % p.ncg; rr /= p.ncg;
  const int band = rr % p.nband; rr /= p.nband;
  const int seg = rr % p.nseg;
  const int n = rr / p.nseg;
  const int row_id = unit / p.ncg;
  const int H = p.H, W = p.W, C = p.C, Ho = p.Ho, Wo = p.Wo;
  const int c0 = cg * CT * 16;
  constexpr int KQ = K / 4;

  float wf[CT][KQ], xs[KQ], xh[KQ];
  float4 sc[CT], sh[CT];
#pragma unroll
  for (int s = 0; s < KQ; ++s) {
    xs[s] = p.xs ? p.xs[q * KQ + s] : 1.f;
    xh[s] = p.xh ? p.xh[q * KQ + s] : 0.f;
  }
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
    for (int s = 0; s < KQ; ++s) wf[ct][s] = p.w1[(size_t)(q * KQ + s) * C + c0 + 16 * ct + j];
    sc[ct] = ld4(p.s1 + c0 + 16 * ct + 4 * q);
    sh[ct] = ld4(p.h1 + c0 + 16 * ct + 4 * q);
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
  const int ixac = ixa < 0 ? 0 : (ixa >= W ? W - 1 : ixa), ixbc = ixb < 0 ? 0 : (ixb >= W ? W - 1 : ixb);
  const float* xn_ = p.x + (size_t)n * H * W * p.ldx + q * KQ;

  auto load_row = [&](int iy, float (&ra)[KQ], float (&rb)[KQ]) {
    const int iyc = iy < 0 ? 0 : (iy >= H ? H - 1 : iy);
    const float* rp = xn_ + (size_t)iyc * W * p.ldx;
    irb_load_x<K>(rp + (size_t)ixac * p.ldx, ra);
    if constexpr (S == 2) irb_load_x<K>(rp + (size_t)ixbc * p.ldx, rb);
  };
  // expanded, normalised, activated tile; zero where the pixel lies in the padding
  auto tile = [&](float (&xv)[KQ], bool valid, float4 (&t)[CT]) {
    irb_prologue<K>(xv, xs, xh, xact);
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      const irb_f4 z = irb_expand<K>(wf[ct], xv);
      t[ct].x = valid ? act_apply(fmaf(z[0], sc[ct].x, sh[ct].x), act1) : 0.f;
      t[ct].y = valid ? act_apply(fmaf(z[1], sc[ct].y, sh[ct].y), act1) : 0.f;
      t[ct].z = valid ? act_apply(fmaf(z[2], sc[ct].z, sh[ct].z), act1) : 0.f;
      t[ct].w = valid ? act_apply(fmaf(z[3], sc[ct].w, sh[ct].w), act1) : 0.f;
    }
  };
  auto wtap = [&](int ky, int kx, int ct) { return ld4(sm + (size_t)(3 * ky + kx) * C + c0 + 16 * ct + 4 * q); };
  auto shl4 = [&](float4 v) { return make_float4(irb_from_next(v.x), irb_from_next(v.y), irb_from_next(v.z), irb_from_next(v.w)); };
  auto shr4 = [&](float4 v) { return make_float4(irb_from_prev(v.x), irb_from_prev(v.y), irb_from_prev(v.z), irb_from_prev(v.w)); };

  float4 s1[CT], s2[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) { s1[ct] = zero4(); s2[ct] = zero4(); }
  auto emit = [&](int o, const float4 (&acc)[CT]) {
    float* yp = p.y + ((size_t)(n * Ho + o) * Wo + (out_ok ? ox : 0)) * p.ldy + c0 + 4 * q;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      if (out_ok) {
        st4(yp + 16 * ct, acc[ct]);
        s1[ct] = add4(s1[ct], acc[ct]);
        s2[ct] = fma4(acc[ct], acc[ct], s2[ct]);
      }
    }
  };

  float xc0[KQ], xc1[KQ], xn0[KQ], xn1[KQ];
  float4 ta[CT], tb[CT];
  if constexpr (S == 2) {
    float4 cur[CT], nxt[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) cur[ct] = zero4();
    int iy = 2 * oy0 - p.pad_t;
    load_row(iy, xc0, xc1);
    load_row(iy + 1, xn0, xn1);
    {
      const bool vr = iy >= 0 && iy < H;
      tile(xc0, vr && va, ta);
      tile(xc1, vr && vb, tb);
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        const float4 tc = shl4(ta[ct]);
        cur[ct] = fma4(wtap(0, 0, ct), ta[ct], cur[ct]);
        cur[ct] = fma4(wtap(0, 1, ct), tb[ct], cur[ct]);
        cur[ct] = fma4(wtap(0, 2, ct), tc, cur[ct]);
      }
    }
    for (int o = oy0; o < oy1; ++o) {
      // odd row (ky = 1)
      ++iy;
#pragma unroll
      for (int s = 0; s < KQ; ++s) { xc0[s] = xn0[s]; xc1[s] = xn1[s]; }
      load_row(iy + 1, xn0, xn1);
      {
        const bool vr = iy >= 0 && iy < H;
        tile(xc0, vr && va, ta);
        tile(xc1, vr && vb, tb);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          const float4 tc = shl4(ta[ct]);
          cur[ct] = fma4(wtap(1, 0, ct), ta[ct], cur[ct]);
          cur[ct] = fma4(wtap(1, 1, ct), tb[ct], cur[ct]);
          cur[ct] = fma4(wtap(1, 2, ct), tc, cur[ct]);
        }
      }
      // even row: ky = 2 of this output row, ky = 0 of the next
      ++iy;
#pragma unroll
      for (int s = 0; s < KQ; ++s) { xc0[s] = xn0[s]; xc1[s] = xn1[s]; }
      load_row(iy + 1, xn0, xn1);
      {
        const bool vr = iy >= 0 && iy < H;
        tile(xc0, vr && va, ta);
        tile(xc1, vr && vb, tb);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          const float4 tc = shl4(ta[ct]);
          cur[ct] = fma4(wtap(2, 0, ct), ta[ct], cur[ct]);
          cur[ct] = fma4(wtap(2, 1, ct), tb[ct], cur[ct]);
          cur[ct] = fma4(wtap(2, 2, ct), tc, cur[ct]);
          nxt[ct] = mul4(wtap(0, 0, ct), ta[ct]);
          nxt[ct] = fma4(wtap(0, 1, ct), tb[ct], nxt[ct]);
          nxt[ct] = fma4(wtap(0, 2, ct), tc, nxt[ct]);
        }
      }
      emit(o, cur);
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) cur[ct] = nxt[ct];
    }
  } else {
    float4 a2[CT], a1[CT], a0[CT];                    // output rows t - 2, t - 1, t
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) { a2[ct] = zero4(); a1[ct] = zero4(); }
    const int nrows = oy1 - oy0;
    int iy = oy0 - p.pad_t;
    load_row(iy, xn0, xn1);
    for (int t = 0; t < nrows + 2; ++t, ++iy) {
#pragma unroll
      for (int s = 0; s < KQ; ++s) xc0[s] = xn0[s];
      load_row(iy + 1, xn0, xn1);
      const bool vr = iy >= 0 && iy < H;
      tile(xc0, vr && va, ta);
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        const float4 tl = shr4(ta[ct]), tr = shl4(ta[ct]);
        a2[ct] = fma4(wtap(2, 0, ct), tl, a2[ct]);
        a2[ct] = fma4(wtap(2, 1, ct), ta[ct], a2[ct]);
        a2[ct] = fma4(wtap(2, 2, ct), tr, a2[ct]);
        a1[ct] = fma4(wtap(1, 0, ct), tl, a1[ct]);
        a1[ct] = fma4(wtap(1, 1, ct), ta[ct], a1[ct]);
        a1[ct] = fma4(wtap(1, 2, ct), tr, a1[ct]);
        a0[ct] = mul4(wtap(0, 0, ct), tl);
        a0[ct] = fma4(wtap(0, 1, ct), ta[ct], a0[ct]);
        a0[ct] = fma4(wtap(0, 2, ct), tr, a0[ct]);
      }
      if (t >= 2) emit(oy0 + t - 2, a2);
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) { a2[ct] = a1[ct]; a1[ct] = a0[ct]; }
    }
  }

  if (p.partials) {
    float* prow = p.partials + (size_t)row_id * 2 * C + c0 + 4 * q;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      float4 a, b;
      a.x = irb_row_sum(s1[ct].x); a.y = irb_row_sum(s1[ct].y); a.z = irb_row_sum(s1[ct].z); a.w = irb_row_sum(s1[ct].w);
      b.x = irb_row_sum(s2[ct].x); b.y = irb_row_sum(s2[ct].y); b.z = irb_row_sum(s2[ct].z); b.w = irb_row_sum(s2[ct].w);
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
extern "C" int dl3p_irb_set_plan(int ct, int want_waves) { g_irb_ct = ct; g_irb_waves = want_waves; return DL3P_OK; }

static int irb_pick_ct(int C) {
  if (g_irb_ct > 0 && C % (16 * g_irb_ct) == 0) return g_irb_ct;
  if (C % 48 == 0) return 3;
  if (C % 32 == 0) return 2;
  return 1;
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
  const size_t shm = (size_t)9 * p.C * sizeof(float);
  if (S == 2)
    dl3p_launch(irb_fwd_kernel<K, CT, 2>, dim3(grid), dim3(256), shm, st, p);
  else
    dl3p_launch(irb_fwd_kernel<K, CT, 1>, dim3(grid), dim3(256), shm, st, p);
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
#define IRB_FWD_CASE(KK, CC) if (K == KK && CT == CC) irb_fwd_launch<KK, CC>(p, stride, st)
  IRB_FWD_CASE(16, 1); IRB_FWD_CASE(16, 2); IRB_FWD_CASE(16, 3);
  IRB_FWD_CASE(24, 1); IRB_FWD_CASE(24, 2); IRB_FWD_CASE(24, 3);
  IRB_FWD_CASE(32, 1); IRB_FWD_CASE(32, 2); IRB_FWD_CASE(32, 3);
#undef IRB_FWD_CASE
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
