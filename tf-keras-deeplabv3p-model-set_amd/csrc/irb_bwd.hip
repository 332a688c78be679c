// Fused inverted-residual BACKWARD (deeplabv3p_mobilenetv2.py:38-74; forward: irb_fwd.hip).  The expanded tensor z1 and
// its gradient (each 6x the block input) are never in HBM: both passes recompute the expand tile from the block input on
// v_mfma_f32_16x16x4_f32 -- bit for bit the forward's values (same fragments, same k order).
//
//   pass A  dl3p_irb_bwd_sums   reads x, dz2 (gradient of the raw depthwise output)
//           da1 = depthwise^T(dz2), g' = da1 * act'(BN1(z1)); leaves the BatchNorm-backward partial rows
//           [sum g', sum g' xhat1] of BN1 and the depthwise kernel's gradient sum_p a1 * dz2 as slabs.
//   (dl3p_bn_bwd_finalize turns the rows into BN1's coefficient triple, as for every other BatchNorm)
//   pass B  dl3p_irb_bwd_data   reads x, dz2 again
//           dz1 = c0 (g' - c1 - xhat1 c2); expand kernel gradient sum_p x^T dz1 as slabs; dx = dz1 W1^T (+)= into the
//           gradient of the block input, optionally with the BatchNorm-backward rows of the BatchNorm in FRONT of the block.
//
// Tiles as in irb_common.h (pixel on the lane, 4 channels in the result registers).  A stride-2 depthwise conv makes four
// kinds of input pixel by the parity of (row + pad_t, column + pad_l): 4, 2, 2 or 1 of the 9 taps reach it.  A wave takes
// 16 "column steps" cs (input columns 2cs - pad_l and 2cs + 1 - pad_l) and walks "row steps" o the same way, so every tile
// has ONE wave-uniform tap set and dz2 is read as plain 16-byte rows: DA = dz2[o][cs], DB = dz2[o][cs - 1] and the same
// of row o - 1.  Stride 1: one kind, 9 taps, dz2 rows oy = iy + pad_t - ky at column shifts 0..2.
#include "irb_common.h"

struct TapConsts { float4 sc, sh, mu, is; };

// a = act(u), d = da * act'(u), accumulates the BatchNorm-backward sums; returns a (0 outside the image)
__device__ __forceinline__ float4 irb_bn_point(const irb_f4 z, const TapConsts& k, int act, bool valid, float4 da, float4& s1,
                                               float4& s2) {
  const float4 zz = make_float4(z[0], z[1], z[2], z[3]);
  const float4 u = fma4(zz, k.sc, k.sh);
  float4 a, d;
  a.x = valid ? act_apply(u.x, act) : 0.f; a.y = valid ? act_apply(u.y, act) : 0.f;
  a.z = valid ? act_apply(u.z, act) : 0.f; a.w = valid ? act_apply(u.w, act) : 0.f;
  d.x = valid ? da.x * act_grad(u.x, act) : 0.f; d.y = valid ? da.y * act_grad(u.y, act) : 0.f;
  d.z = valid ? da.z * act_grad(u.z, act) : 0.f; d.w = valid ? da.w * act_grad(u.w, act) : 0.f;
  const float4 xh = make_float4((zz.x - k.mu.x) * k.is.x, (zz.y - k.mu.y) * k.is.y, (zz.z - k.mu.z) * k.is.z,
                                (zz.w - k.mu.w) * k.is.w);
  s1 = add4(s1, d);
  s2 = fma4(d, xh, s2);
  return a;
}

__device__ __forceinline__ float4 irb_row_sum4(float4 v) {
  return make_float4(irb_row_sum(v.x), irb_row_sum(v.y), irb_row_sum(v.z), irb_row_sum(v.w));
}

// ------------------------------------------------------------------------------------------ pass A
// one wave = (image, 16 column steps, band of row steps, ONE 16-channel tile)
template <int K, int S>
__global__ __launch_bounds__(256) void irb_bwd_sums_kernel(IrbParams p) {
  extern __shared__ float sm[];                      // depthwise kernel [9][C]
  for (int i = threadIdx.x; i < 9 * p.C; i += 256) sm[i] = p.wdw[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, q = lane >> 4;
  const int unit = irb_wg_index(blockIdx.x, gridDim.x) * 4 + wave;
  if (unit >= p.units) return;
  int rr = unit;
  const int ct = rr % p.ncg; rr /= p.ncg;
  const int band = rr % p.nband; rr /= p.nband;
  const int seg = rr % p.nseg;
  const int n = rr / p.nseg;
  const int row_id = unit / p.ncg;
  const int H = p.H, W = p.W, C = p.C, Ho = p.Ho, Wo = p.Wo;
  constexpr int KQ = K / 4;
  const int cl = 16 * ct + 4 * q;

  float wf[KQ], xs[KQ], xh[KQ];
#pragma unroll
  for (int s = 0; s < KQ; ++s) {
    xs[s] = p.xs ? p.xs[q * KQ + s] : 1.f;
    xh[s] = p.xh ? p.xh[q * KQ + s] : 0.f;
    wf[s] = p.w1[(size_t)(q * KQ + s) * C + 16 * ct + j];
  }
  TapConsts kc;
  kc.sc = ld4(p.s1 + cl); kc.sh = ld4(p.h1 + cl); kc.mu = ld4(p.mu1 + cl); kc.is = ld4(p.is1 + cl);
  const int xact = p.xact, act1 = p.act1;
  float4 wt[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) wt[t] = ld4(sm + (size_t)t * C + cl);
  float4 gw[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) gw[t] = zero4();
  float4 s1 = zero4(), s2 = zero4();
  const float* xn_ = p.x + (size_t)n * H * W * p.ldx + q * KQ;
  const float* dyn = p.dy + (size_t)n * Ho * Wo * p.lddy + cl;
  const int s0 = band * p.band;

  if constexpr (S == 2) {
    const int NO = (H + p.pad_t + 1) / 2;
    const int s1e = s0 + p.band < NO ? s0 + p.band : NO;
    const int cs = 16 * seg + j;
    const int ixa = 2 * cs - p.pad_l, ixb = ixa + 1;
    const bool va = ixa >= 0 && ixa < W, vb = ixb < W;
    const int ixac = ixa < 0 ? 0 : (ixa >= W ? W - 1 : ixa), ixbc = ixb >= W ? W - 1 : ixb;
    auto loadD = [&](int o, int shift) {
      const int oxx = cs - shift;
      const bool ok = o >= 0 && o < Ho && oxx >= 0 && oxx < Wo;
      const float4 v = ld4(dyn + ((size_t)(ok ? o : 0) * Wo + (ok ? oxx : 0)) * p.lddy);
      return ok ? v : zero4();
    };
    auto load_x2 = [&](int iy, float (&ra)[KQ], float (&rb)[KQ]) {
      const int iyc = iy < 0 ? 0 : (iy >= H ? H - 1 : iy);
      const float* rp = xn_ + (size_t)iyc * W * p.ldx;
      irb_load_x<K>(rp + (size_t)ixac * p.ldx, ra);
      irb_load_x<K>(rp + (size_t)ixbc * p.ldx, rb);
    };
    float4 DAp = loadD(s0 - 1, 0), DBp = loadD(s0 - 1, 1);
    float xe0[KQ], xe1[KQ], xo0[KQ], xo1[KQ];
    for (int o = s0; o < s1e; ++o) {
      const int iyE = 2 * o - p.pad_t, iyO = iyE + 1;
      load_x2(iyE, xe0, xe1);
      load_x2(iyO, xo0, xo1);
      const float4 DA = loadD(o, 0), DB = loadD(o, 1);
      const bool vE = iyE >= 0 && iyE < H, vO = iyO < H;
      irb_prologue<K>(xe0, xs, xh, xact); irb_prologue<K>(xe1, xs, xh, xact);
      irb_prologue<K>(xo0, xs, xh, xact); irb_prologue<K>(xo1, xs, xh, xact);
      {   // (even row, even column): taps (0,0) (0,2) (2,0) (2,2)
        float4 da = mul4(wt[0], DA);
        da = fma4(wt[2], DB, da); da = fma4(wt[6], DAp, da); da = fma4(wt[8], DBp, da);
        const float4 a = irb_bn_point(irb_expand<K>(wf, xe0), kc, act1, vE && va, da, s1, s2);
        gw[0] = fma4(a, DA, gw[0]); gw[2] = fma4(a, DB, gw[2]); gw[6] = fma4(a, DAp, gw[6]); gw[8] = fma4(a, DBp, gw[8]);
      }
      {   // (even row, odd column): taps (0,1) (2,1)
        float4 da = mul4(wt[1], DA);
        da = fma4(wt[7], DAp, da);
        const float4 a = irb_bn_point(irb_expand<K>(wf, xe1), kc, act1, vE && vb, da, s1, s2);
        gw[1] = fma4(a, DA, gw[1]); gw[7] = fma4(a, DAp, gw[7]);
      }
      {   // (odd row, even column): taps (1,0) (1,2)
        float4 da = mul4(wt[3], DA);
        da = fma4(wt[5], DB, da);
        const float4 a = irb_bn_point(irb_expand<K>(wf, xo0), kc, act1, vO && va, da, s1, s2);
        gw[3] = fma4(a, DA, gw[3]); gw[5] = fma4(a, DB, gw[5]);
      }
      {   // (odd row, odd column): tap (1,1)
        const float4 da = mul4(wt[4], DA);
        const float4 a = irb_bn_point(irb_expand<K>(wf, xo1), kc, act1, vO && vb, da, s1, s2);
        gw[4] = fma4(a, DA, gw[4]);
      }
      DAp = DA; DBp = DB;
    }
  } else {
    const int s1e = s0 + p.band < H ? s0 + p.band : H;
    const int ix = 16 * seg + j;
    const bool vx = ix < W;
    const int ixc = vx ? ix : W - 1;
    auto loadD = [&](int oy, int c) {
      const int oxx = ix + p.pad_l - c;
      const bool ok = oy >= 0 && oy < Ho && oxx >= 0 && oxx < Wo;
      const float4 v = ld4(dyn + ((size_t)(ok ? oy : 0) * Wo + (ok ? oxx : 0)) * p.lddy);
      return ok ? v : zero4();
    };
    float4 D[3][3];                                   // D[r][c] = dz2[iy + pad_t - r][ix + pad_l - c]
#pragma unroll
    for (int c = 0; c < 3; ++c) { D[1][c] = loadD(s0 + p.pad_t - 1, c); D[2][c] = loadD(s0 + p.pad_t - 2, c); }
    float xv[KQ];
    for (int iy = s0; iy < s1e; ++iy) {
      irb_load_x<K>(xn_ + ((size_t)iy * W + ixc) * p.ldx, xv);
#pragma unroll
      for (int c = 0; c < 3; ++c) D[0][c] = loadD(iy + p.pad_t, c);
      irb_prologue<K>(xv, xs, xh, xact);
      float4 da = zero4();
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) da = fma4(wt[3 * r + c], D[r][c], da);
      const float4 a = irb_bn_point(irb_expand<K>(wf, xv), kc, act1, vx, da, s1, s2);
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) gw[3 * r + c] = fma4(a, D[r][c], gw[3 * r + c]);
#pragma unroll
      for (int c = 0; c < 3; ++c) { D[2][c] = D[1][c]; D[1][c] = D[0][c]; }
    }
  }

  // one slab / partial row per (image, segment, band); this wave's 16 channels of it
  float* slab = p.slabs + (size_t)row_id * 9 * C + cl;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const float4 v = irb_row_sum4(gw[t]);
    if (j == 0) st4(slab + (size_t)t * C, v);
  }
  const float4 a = irb_row_sum4(s1), b = irb_row_sum4(s2);
  if (j == 0) {
    float* prow = p.partials + (size_t)row_id * 2 * C + cl;
    st4(prow, a);
    st4(prow + C, b);
  }
}

// ------------------------------------------------------------------------------------------ pass B
// one WORKGROUP = (image, 16 column steps, band of row steps); its NW waves own CTW 16-channel tiles each.  Per step every
// wave forms dz1 for its channels, multiplies it into its slice of the expand-kernel gradient (through a wave-private LDS
// transpose: that product sums over pixels, the lane index) and into a partial dx (sums over channels, the register index:
// the result registers ARE the B operand); the partial dx tiles of the NW waves meet in LDS and are added in wave order.
template <int K, int CTW, int NW, int S>
__global__ __launch_bounds__(64 * NW) void irb_bwd_data_kernel(IrbParams p) {
  constexpr int KQ = K / 4, KT = (K + 15) / 16, NTL = S == 2 ? 4 : 1, TP = 20;
  extern __shared__ float sm[];
  const int C = p.C;
  float* s_w = sm;                                     // depthwise kernel [9][C]
  float* s_k = s_w + 9 * C;                            // sc, sh, mu, is, c0, c1, c2 [7][C]
  float* s_t = s_k + 7 * C;                            // transpose tiles [NW][16][TP]
  float4* s_dx = reinterpret_cast<float4*>(s_t + NW * 16 * TP);   // partial dx [NW][NTL*KT][64]
  for (int i = threadIdx.x; i < 9 * C; i += 64 * NW) s_w[i] = p.wdw[i];
  for (int i = threadIdx.x; i < C; i += 64 * NW) {
    s_k[i] = p.s1[i]; s_k[C + i] = p.h1[i]; s_k[2 * C + i] = p.mu1[i]; s_k[3 * C + i] = p.is1[i];
    s_k[4 * C + i] = p.coef1[i]; s_k[5 * C + i] = p.coef1[C + i]; s_k[6 * C + i] = p.coef1[2 * C + i];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, q = lane >> 4;
  const int unit = irb_wg_index(blockIdx.x, gridDim.x);
  if (unit >= p.units) return;                         // (whole workgroup)
  int rr = unit;
  const int band = rr % p.nband; rr /= p.nband;
  const int seg = rr % p.nseg;
  const int n = rr / p.nseg;
  const int H = p.H, W = p.W, Ho = p.Ho, Wo = p.Wo;
  const int cw = wave * CTW * 16;                      // first channel of this wave
  const bool want_dx = p.gx != nullptr;

  float wf[CTW][KQ], xs[KQ], xh[KQ], xsT[KT], xhT[KT];
  float4 wa[CTW][KT];                                  // dx A operand: W1[k = 16kt + j][c = 16ct + 4q ..]
#pragma unroll
  for (int s = 0; s < KQ; ++s) {
    xs[s] = p.xs ? p.xs[q * KQ + s] : 1.f;
    xh[s] = p.xh ? p.xh[q * KQ + s] : 0.f;
  }
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) {
    const int k = 16 * kt + j;
    xsT[kt] = (p.xs && k < K) ? p.xs[k] : 1.f;
    xhT[kt] = (p.xh && k < K) ? p.xh[k] : 0.f;
  }
#pragma unroll
  for (int c = 0; c < CTW; ++c) {
#pragma unroll
    for (int s = 0; s < KQ; ++s) wf[c][s] = p.w1[(size_t)(q * KQ + s) * C + cw + 16 * c + j];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      const int k = 16 * kt + j;
      wa[c][kt] = k < K ? ld4(p.w1 + (size_t)k * C + cw + 16 * c + 4 * q) : zero4();
    }
  }
  const int xact = p.xact, act1 = p.act1;
  irb_f4 gw[CTW][KT];
#pragma unroll
  for (int c = 0; c < CTW; ++c)
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) gw[c][kt] = (irb_f4){0.f, 0.f, 0.f, 0.f};
  float4 f0a[KT], f0b[KT];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) { f0a[kt] = zero4(); f0b[kt] = zero4(); }
  float* tbuf = s_t + wave * 16 * TP;

  const float* xn_ = p.x + (size_t)n * H * W * p.ldx;
  const float* dyn = p.dy + (size_t)n * Ho * Wo * p.lddy + cw + 4 * q;
  const int s0 = band * p.band;
  int s1e, cs = 16 * seg + j;
  int ixt[NTL];                                        // this lane's input column in tile t
  if constexpr (S == 2) {
    const int NO = (H + p.pad_t + 1) / 2;
    s1e = s0 + p.band < NO ? s0 + p.band : NO;
    ixt[0] = ixt[2] = 2 * cs - p.pad_l;
    ixt[1] = ixt[3] = 2 * cs - p.pad_l + 1;
  } else {
    s1e = s0 + p.band < H ? s0 + p.band : H;
    ixt[0] = cs;
  }
  // pixel 4q + s of tile t (the transposed operand of the weight-gradient product): its column
  auto col_of = [&](int t, int jj) {
    const int c2 = 16 * seg + jj;
    if constexpr (S == 2) return 2 * c2 - p.pad_l + (t & 1);
    else return c2;
  };

  for (int st = s0; st < s1e; ++st) {
    int iyt[NTL];
    if constexpr (S == 2) { iyt[0] = iyt[1] = 2 * st - p.pad_t; iyt[2] = iyt[3] = 2 * st - p.pad_t + 1; }
    else iyt[0] = st;
    float xv[NTL][KQ], xT[NTL][KT][4];
    bool vt[NTL];
#pragma unroll
    for (int t = 0; t < NTL; ++t) {
      const int iy = iyt[t], ix = ixt[t];
      vt[t] = iy >= 0 && iy < H && ix >= 0 && ix < W;
      const int iyc = iy < 0 ? 0 : (iy >= H ? H - 1 : iy), ixc = ix < 0 ? 0 : (ix >= W ? W - 1 : ix);
      irb_load_x<K>(xn_ + ((size_t)iyc * W + ixc) * p.ldx + q * KQ, xv[t]);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        int ic = col_of(t, 4 * q + s);
        ic = ic < 0 ? 0 : (ic >= W ? W - 1 : ic);
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
          const int k = 16 * kt + j;
          xT[t][kt][s] = xn_[((size_t)iyc * W + ic) * p.ldx + (k < K ? k : 0)];
        }
      }
    }
#pragma unroll
    for (int t = 0; t < NTL; ++t) {
      irb_prologue<K>(xv[t], xs, xh, xact);
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int s = 0; s < 4; ++s) xT[t][kt][s] = act_apply(fmaf(xT[t][kt][s], xsT[kt], xhT[kt]), xact);
    }
    irb_f4 dxp[NTL][KT];
#pragma unroll
    for (int t = 0; t < NTL; ++t)
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) dxp[t][kt] = (irb_f4){0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int c = 0; c < CTW; ++c) {
      const int cc = cw + 16 * c + 4 * q;             // this lane's 4 channels
      auto loadD = [&](int oy, int oxx) {
        const bool ok = oy >= 0 && oy < Ho && oxx >= 0 && oxx < Wo;
        const float4 v = ld4(dyn + 16 * c + ((size_t)(ok ? oy : 0) * Wo + (ok ? oxx : 0)) * p.lddy);
        return ok ? v : zero4();
      };
      float4 da[NTL];
      if constexpr (S == 2) {
        const float4 DA = loadD(st, cs), DB = loadD(st, cs - 1), DAp = loadD(st - 1, cs), DBp = loadD(st - 1, cs - 1);
        auto w9 = [&](int t) { return ld4(s_w + (size_t)t * C + cc); };
        da[0] = mul4(w9(0), DA); da[0] = fma4(w9(2), DB, da[0]); da[0] = fma4(w9(6), DAp, da[0]); da[0] = fma4(w9(8), DBp, da[0]);
        da[1] = mul4(w9(1), DA); da[1] = fma4(w9(7), DAp, da[1]);
        da[2] = mul4(w9(3), DA); da[2] = fma4(w9(5), DB, da[2]);
        da[3] = mul4(w9(4), DA);
      } else {
        da[0] = zero4();
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
          for (int c3 = 0; c3 < 3; ++c3)
            da[0] = fma4(ld4(s_w + (size_t)(3 * r + c3) * C + cc), loadD(st + p.pad_t - r, cs + p.pad_l - c3), da[0]);
      }
      const float4 ksc = ld4(s_k + cc), ksh = ld4(s_k + C + cc), kmu = ld4(s_k + 2 * C + cc), kis = ld4(s_k + 3 * C + cc);
      const float4 k0 = ld4(s_k + 4 * C + cc), k1 = ld4(s_k + 5 * C + cc), k2 = ld4(s_k + 6 * C + cc);
#pragma unroll
      for (int t = 0; t < NTL; ++t) {
        const irb_f4 z = irb_expand<K>(wf[c], xv[t]);
        float dz[4];
        const float zz[4] = {z[0], z[1], z[2], z[3]};
        const float dav[4] = {da[t].x, da[t].y, da[t].z, da[t].w};
        const float scv[4] = {ksc.x, ksc.y, ksc.z, ksc.w}, shv[4] = {ksh.x, ksh.y, ksh.z, ksh.w};
        const float muv[4] = {kmu.x, kmu.y, kmu.z, kmu.w}, isv[4] = {kis.x, kis.y, kis.z, kis.w};
        const float c0v[4] = {k0.x, k0.y, k0.z, k0.w}, c1v[4] = {k1.x, k1.y, k1.z, k1.w}, c2v[4] = {k2.x, k2.y, k2.z, k2.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float u = fmaf(zz[i], scv[i], shv[i]);
          const float d = dav[i] * act_grad(u, act1);
          const float v = c0v[i] * (d - c1v[i] - (zz[i] - muv[i]) * isv[i] * c2v[i]);
          dz[i] = vt[t] ? v : 0.f;
        }
        if (want_dx) {
          // dx[k][pixel] += sum_c W1[k][c] dz1[pixel][c]: the result registers are the B operand (k-slot q <-> channel 4q + i)
#pragma unroll
          for (int kt = 0; kt < KT; ++kt) {
            dxp[t][kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[c][kt].x, dz[0], dxp[t][kt], 0, 0, 0);
            dxp[t][kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[c][kt].y, dz[1], dxp[t][kt], 0, 0, 0);
            dxp[t][kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[c][kt].z, dz[2], dxp[t][kt], 0, 0, 0);
            dxp[t][kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[c][kt].w, dz[3], dxp[t][kt], 0, 0, 0);
          }
        }
        // gw[c_local][k] += sum_pixel dz1[pixel][c] x[pixel][k]: dz1 transposed through the wave's LDS tile
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < 4; ++i) tbuf[(4 * q + i) * TP + j] = dz[i];
        __builtin_amdgcn_wave_barrier();
        const float4 at = ld4(tbuf + j * TP + 4 * q);  // channel j of the tile, pixels 4q .. 4q + 3
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
          gw[c][kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(at.x, xT[t][kt][0], gw[c][kt], 0, 0, 0);
          gw[c][kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(at.y, xT[t][kt][1], gw[c][kt], 0, 0, 0);
          gw[c][kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(at.z, xT[t][kt][2], gw[c][kt], 0, 0, 0);
          gw[c][kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(at.w, xT[t][kt][3], gw[c][kt], 0, 0, 0);
        }
      }
    }

    if (want_dx) {
#pragma unroll
      for (int t = 0; t < NTL; ++t)
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
          s_dx[(wave * NTL * KT + t * KT + kt) * 64 + lane] = make_float4(dxp[t][kt][0], dxp[t][kt][1], dxp[t][kt][2], dxp[t][kt][3]);
      __syncthreads();
      // tile (t, kt) is finished by wave (t * KT + kt) % NW: lane = (pixel j, input channels 16kt + 4q ..)
#pragma unroll
      for (int t = 0; t < NTL; ++t)
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
          if ((t * KT + kt) % NW != wave) continue;
          float4 v = s_dx[(t * KT + kt) * 64 + lane];
#pragma unroll
          for (int w = 1; w < NW; ++w) v = add4(v, s_dx[(w * NTL * KT + t * KT + kt) * 64 + lane]);
          const int k = 16 * kt + 4 * q;
          if (vt[t] && k < K) {
            const size_t pix = (size_t)(n * H + iyt[t]) * W + ixt[t];
            float* gp = p.gx + pix * p.ldgx + k;
            if (p.accumulate) v = add4(v, ld4(gp));
            st4(gp, v);
            if (p.partials0) {
              const float4 z0 = ld4(p.z0 + pix * p.ldz0 + k);
              const float4 one = make_float4(1.f, 1.f, 1.f, 1.f);
              const float4 a0 = p.s0 ? ld4(p.s0 + k) : one, b0 = p.h0 ? ld4(p.h0 + k) : zero4();
              const float4 m0 = p.mu0 ? ld4(p.mu0 + k) : zero4(), i0 = p.is0 ? ld4(p.is0 + k) : one;
              const float4 u = fma4(z0, a0, b0);
              const float4 d = make_float4(v.x * act_grad(u.x, p.act0), v.y * act_grad(u.y, p.act0), v.z * act_grad(u.z, p.act0),
                                           v.w * act_grad(u.w, p.act0));
              const float4 xh0 = make_float4((z0.x - m0.x) * i0.x, (z0.y - m0.y) * i0.y, (z0.z - m0.z) * i0.z, (z0.w - m0.w) * i0.w);
              f0a[kt] = add4(f0a[kt], d);
              f0b[kt] = fma4(d, xh0, f0b[kt]);
            }
          }
        }
      __syncthreads();
    }
  }

  // expand-kernel gradient slab of this workgroup: [K][C], this wave's channel columns
  float* slab = p.slabs + (size_t)unit * K * C;
#pragma unroll
  for (int c = 0; c < CTW; ++c)
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      const int k = 16 * kt + j;
      if (k < K) st4(slab + (size_t)k * C + cw + 16 * c + 4 * q, make_float4(gw[c][kt][0], gw[c][kt][1], gw[c][kt][2], gw[c][kt][3]));
    }
  if (p.partials0) {
    // rows of the BatchNorm in front: sum over the pixel lanes, then over the waves (in wave order)
    float* red = reinterpret_cast<float*>(s_dx);      // [NW][2][KT][4 q][4]
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      const float4 a = irb_row_sum4(f0a[kt]), b = irb_row_sum4(f0b[kt]);
      if (j == 0) {
        st4(red + ((wave * 2 + 0) * KT + kt) * 16 + 4 * q, a);
        st4(red + ((wave * 2 + 1) * KT + kt) * 16 + 4 * q, b);
      }
    }
    __syncthreads();
    if (wave == 0 && lane < 2 * KT * 16) {
      const int v = lane / (KT * 16), kk = lane % (KT * 16);
      float acc = 0.f;
      for (int w = 0; w < NW; ++w) acc += red[((w * 2 + v) * KT) * 16 + kk];
      if (kk < K) p.partials0[(size_t)unit * 2 * K + (size_t)v * K + kk] = acc;
    }
  }
}

// ------------------------------------------------------------------------------------------ host side
static int irb_bwd_plan(IrbParams& p, int N, int H, int W, int stride, int pad_t, int pad_l, int ncg, int want_units) {
  const int rsteps = stride == 2 ? (H + pad_t + 1) / 2 : H;
  const int csteps = stride == 2 ? (W + pad_l + 1) / 2 : W;
  p.nseg = ceil_div(csteps, 16);
  p.ncg = ncg;
  const long long per = (long long)N * p.nseg * ncg;
  int nband = (int)ceil_div_ll(want_units, per);
  if (nband < 1) nband = 1;
  const int max_by_rows = DL3P_MAX_STAT_ROWS / (N * p.nseg);
  if (max_by_rows < 1) return -1;
  if (nband > max_by_rows) nband = max_by_rows;
  int band = ceil_div(rsteps, nband);
  if (band < 4) band = rsteps < 4 ? rsteps : 4;
  nband = ceil_div(rsteps, band);
  p.band = band;
  p.nband = nband;
  p.units = (int)(per * nband);
  return N * p.nseg * nband;
}

static int g_irb_bwd_units_a = 0, g_irb_bwd_units_b = 0;
extern "C" int dl3p_irb_set_bwd_plan(int want_waves_sums, int want_workgroups_data) {
  g_irb_bwd_units_a = want_waves_sums;
  g_irb_bwd_units_b = want_workgroups_data;
  return DL3P_OK;
}

extern "C" int dl3p_irb_bwd_supported(int N, int H, int W, int K, int C, int k, int stride, int rate, int pad_t, int pad_l, int Ho,
                                      int Wo) {
  if (!dl3p_irb_supported(N, H, W, K, C, k, stride, rate, pad_t, pad_l, Ho, Wo)) return 0;
  if (!((K == 16 && C == 96) || (K == 24 && C == 144) || (K == 32 && C == 192))) return 0;
  IrbParams p = {};
  if (irb_bwd_plan(p, N, H, W, stride, pad_t, pad_l, C / 16, 8192) < 0) return 0;
  if (irb_bwd_plan(p, N, H, W, stride, pad_t, pad_l, 1, 1024) < 0) return 0;
  return 1;
}

/* rows and bytes of the slab regions: which = 0 depthwise-kernel slabs of dl3p_irb_bwd_sums ([rows][9][C]), 1 expand-kernel
 * slabs of dl3p_irb_bwd_data ([rows][K][C]) */
extern "C" size_t dl3p_irb_bwd_workspace(int which, int N, int H, int W, int K, int C, int stride, int pad_t, int pad_l) {
  IrbParams p = {};
  if (which == 0) {
    const int rows = irb_bwd_plan(p, N, H, W, stride, pad_t, pad_l, C / 16, g_irb_bwd_units_a > 0 ? g_irb_bwd_units_a : 8192);
    return rows < 0 ? 0 : (size_t)rows * 9 * C * sizeof(float);
  }
  const int rows = irb_bwd_plan(p, N, H, W, stride, pad_t, pad_l, 1, g_irb_bwd_units_b > 0 ? g_irb_bwd_units_b : 1024);
  return rows < 0 ? 0 : (size_t)rows * K * C * sizeof(float);
}

static void irb_fill(IrbParams& p, const float* x, int ldx, const float* in_scale, const float* in_shift, int in_act,
                     const float* w1, const float* bn_scale, const float* bn_shift, int bn_act, const float* bn_mean,
                     const float* bn_invstd, const float* wdw, const float* dy, int lddy, int N, int H, int W, int C, int pad_t,
                     int pad_l, int Ho, int Wo) {
  p.x = x; p.ldx = ldx; p.xs = in_scale; p.xh = in_shift; p.xact = in_act; p.w1 = w1; p.s1 = bn_scale; p.h1 = bn_shift;
  p.act1 = bn_act; p.mu1 = bn_mean; p.is1 = bn_invstd; p.wdw = wdw; p.dy = dy; p.lddy = lddy;
  p.N = N; p.H = H; p.W = W; p.C = C; p.Ho = Ho; p.Wo = Wo; p.pad_t = pad_t; p.pad_l = pad_l;
}

extern "C" int dl3p_irb_bwd_sums(const float* x, int ldx, const float* in_scale, const float* in_shift, int in_act,
                                 const float* w1, const float* bn_scale, const float* bn_shift, int bn_act, const float* bn_mean,
                                 const float* bn_invstd, const float* wdw, const float* dy, int lddy, float* gwdw_slabs,
                                 size_t slab_bytes, int* slab_rows_out, float* bn_partials, int N, int H, int W, int K, int C,
                                 int stride, int pad_t, int pad_l, int Ho, int Wo, void* stream) {
  DL3P_CHECK_ARG(x && w1 && bn_scale && bn_shift && bn_mean && bn_invstd && wdw && dy && gwdw_slabs && bn_partials,
                 "dl3p_irb_bwd_sums: null pointer");
  DL3P_CHECK_ARG(dl3p_irb_bwd_supported(N, H, W, K, C, 3, stride, 1, pad_t, pad_l, Ho, Wo),
                 "dl3p_irb_bwd_sums: unsupported shape N=%d H=%d W=%d K=%d C=%d stride=%d", N, H, W, K, C, stride);
  DL3P_CHECK_ARG(ldx >= K && ldx % 4 == 0 && aligned16(x) && lddy >= C && lddy % 4 == 0 && aligned16(dy),
                 "dl3p_irb_bwd_sums: bad layout (ldx=%d, lddy=%d)", ldx, lddy);
  IrbParams p = {};
  irb_fill(p, x, ldx, in_scale, in_shift, in_act, w1, bn_scale, bn_shift, bn_act, bn_mean, bn_invstd, wdw, dy, lddy, N, H, W, C,
           pad_t, pad_l, Ho, Wo);
  p.slabs = gwdw_slabs; p.partials = bn_partials;
  const int rows = irb_bwd_plan(p, N, H, W, stride, pad_t, pad_l, C / 16, g_irb_bwd_units_a > 0 ? g_irb_bwd_units_a : 8192);
  DL3P_CHECK_ARG((size_t)rows * 9 * C * sizeof(float) <= slab_bytes, "dl3p_irb_bwd_sums: slab region too small");
  if (slab_rows_out) *slab_rows_out = rows;
  const int grid = ceil_div(ceil_div(p.units, 4), 8) * 8;
  const size_t shm = (size_t)9 * C * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
#define IRB_A_CASE(KK) \
  if (K == KK) { \
    if (stride == 2) dl3p_launch(irb_bwd_sums_kernel<KK, 2>, dim3(grid), dim3(256), shm, st, p); \
    else dl3p_launch(irb_bwd_sums_kernel<KK, 1>, dim3(grid), dim3(256), shm, st, p); \
  }
  IRB_A_CASE(16) IRB_A_CASE(24) IRB_A_CASE(32)
#undef IRB_A_CASE
  DL3P_CHECK_LAUNCH("dl3p_irb_bwd_sums");
  return DL3P_OK;
}

template <int K, int CTW, int NW>
static void irb_bwd_data_launch(const IrbParams& p, int stride, hipStream_t st) {
  constexpr int KT = (K + 15) / 16;
  const int grid = ceil_div(p.units, 8) * 8;
  const int ntl = stride == 2 ? 4 : 1;
  const size_t shm = ((size_t)16 * p.C + NW * 16 * 20) * sizeof(float) + (size_t)NW * ntl * KT * 64 * sizeof(float4);
  if (stride == 2)
    dl3p_launch(irb_bwd_data_kernel<K, CTW, NW, 2>, dim3(grid), dim3(64 * NW), shm, st, p);
  else
    dl3p_launch(irb_bwd_data_kernel<K, CTW, NW, 1>, dim3(grid), dim3(64 * NW), shm, st, p);
}

extern "C" int dl3p_irb_bwd_data(const float* x, int ldx, const float* in_scale, const float* in_shift, int in_act,
                                 const float* w1, const float* bn_scale, const float* bn_shift, int bn_act, const float* bn_mean,
                                 const float* bn_invstd, const float* bn_coef, const float* wdw, const float* dy, int lddy,
                                 float* gw1_slabs, size_t slab_bytes, int* slab_rows_out, float* gx, int ldgx, int accumulate,
                                 const float* z0, int ldz0, const float* scale0, const float* shift0, int act0,
                                 const float* mean0, const float* invstd0, float* partials0, int N, int H, int W, int K, int C,
                                 int stride, int pad_t, int pad_l, int Ho, int Wo, void* stream) {
  DL3P_CHECK_ARG(x && w1 && bn_scale && bn_shift && bn_mean && bn_invstd && bn_coef && wdw && dy && gw1_slabs,
                 "dl3p_irb_bwd_data: null pointer");
  DL3P_CHECK_ARG(dl3p_irb_bwd_supported(N, H, W, K, C, 3, stride, 1, pad_t, pad_l, Ho, Wo),
                 "dl3p_irb_bwd_data: unsupported shape N=%d H=%d W=%d K=%d C=%d stride=%d", N, H, W, K, C, stride);
  DL3P_CHECK_ARG(ldx >= K && ldx % 4 == 0 && aligned16(x) && lddy >= C && lddy % 4 == 0 && aligned16(dy),
                 "dl3p_irb_bwd_data: bad layout (ldx=%d, lddy=%d)", ldx, lddy);
  DL3P_CHECK_ARG(!gx || (ldgx >= K && ldgx % 4 == 0 && aligned16(gx)), "dl3p_irb_bwd_data: bad gradient layout (ld=%d)", ldgx);
  DL3P_CHECK_ARG(!partials0 || (gx && z0 && ldz0 >= K && ldz0 % 4 == 0 && aligned16(z0)),
                 "dl3p_irb_bwd_data: the BatchNorm in front needs gx and z0");
  IrbParams p = {};
  irb_fill(p, x, ldx, in_scale, in_shift, in_act, w1, bn_scale, bn_shift, bn_act, bn_mean, bn_invstd, wdw, dy, lddy, N, H, W, C,
           pad_t, pad_l, Ho, Wo);
  p.coef1 = bn_coef; p.slabs = gw1_slabs; p.gx = gx; p.ldgx = ldgx; p.accumulate = accumulate;
  p.z0 = z0; p.ldz0 = ldz0; p.s0 = scale0; p.h0 = shift0; p.act0 = act0; p.mu0 = mean0; p.is0 = invstd0; p.partials0 = partials0;
  const int rows = irb_bwd_plan(p, N, H, W, stride, pad_t, pad_l, 1, g_irb_bwd_units_b > 0 ? g_irb_bwd_units_b : 1024);
  DL3P_CHECK_ARG((size_t)rows * K * C * sizeof(float) <= slab_bytes, "dl3p_irb_bwd_data: slab region too small");
  if (slab_rows_out) *slab_rows_out = rows;
  hipStream_t st = (hipStream_t)stream;
  if (K == 16) irb_bwd_data_launch<16, 2, 3>(p, stride, st);
  else if (K == 24) irb_bwd_data_launch<24, 3, 3>(p, stride, st);
  else irb_bwd_data_launch<32, 3, 4>(p, stride, st);
  DL3P_CHECK_LAUNCH("dl3p_irb_bwd_data");
  return DL3P_OK;
}
