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
//
// Both kernels are bound by vector-instruction issue beside their MFMAs (9.7 GFLOP of 16x16x4 products for the 257 x 257 block
// against ~10 vector instructions per element of the virtual 406 MB tensor), so the per-element arithmetic is kept minimal:
// compile-time activations, padding masks only where a tile touches the border (wave-uniform), and in pass B the
// BatchNorm-backward triple folded into per-channel constants: dz1 = [0 < u < 6] (w c0) * dz2 + (A z + B) with the depthwise
// kernel pre-scaled by c0 in LDS, A = -c0 c2 invstd, B = c0 (c2 invstd mean - c1).
#include <climits>
#include "irb_common.h"

// ------------------------------------------------------------------------------------------ pass A
// one wave = (image, 16 column steps, band of row steps, ONE 16-channel tile); nothing is shared between waves.
// With a plain affine in front of the block (XACT = NONE) the expand kernel carries the input scale and the MFMA chain starts from
// (W^T in_shift - mean), so the matrix pipe delivers zc = z1 - mean directly: u = zc * scale + beta', xhat = zc * invstd.
template <int K, int S, int ACT, int XACT>
__global__ __launch_bounds__(256, 2) void irb_bwd_sums_kernel(IrbParams p) {
  constexpr bool FOLD = XACT == DL3P_ACT_NONE;
  constexpr int KQ = K / 4;
  extern __shared__ float sm[];                      // zc0[c] = sum_k W[k][c] in_shift[k] - mean[c] (FOLD), depthwise kernel [9][C]
  const int C = p.C;
  float* s_wd = sm + C;
  for (int c = threadIdx.x; c < C; c += 256) {
    float acc = 0.f;
    if (FOLD && p.xh)
      for (int k = 0; k < K; ++k) acc = fmaf(p.w1[(size_t)k * C + c], p.xh[k], acc);
    sm[c] = acc - p.mu1[c];
  }
  for (int i = threadIdx.x; i < 9 * C; i += 256) s_wd[i] = p.wdw[i];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 15, q = lane >> 4;
  const int unit = irb_wg_index(blockIdx.x, gridDim.x) * 4 + wave;
  if (unit >= p.units) return;
  int rr = unit;
  const int ct = rr % p.ncg; rr /= p.ncg;
  const int band = rr % p.nband; rr /= p.nband;
  const int seg = rr % p.nseg;
  const int n = rr / p.nseg;
  const int row_id = unit / p.ncg;
  const int H = p.H, W = p.W, Ho = p.Ho, Wo = p.Wo;
  const int cl = 16 * ct + 4 * q;

  float wf[KQ], xs[KQ], xh[KQ];
#pragma unroll
  for (int s = 0; s < KQ; ++s) {
    xs[s] = p.xs ? p.xs[q * KQ + s] : 1.f;
    xh[s] = p.xh ? p.xh[q * KQ + s] : 0.f;
    const float w = p.w1[(size_t)(q * KQ + s) * C + 16 * ct + j];
    wf[s] = FOLD ? w * xs[s] : w;
  }
  const float4 ksc = ld4(p.s1 + cl), kis = ld4(p.is1 + cl), kmu = ld4(p.mu1 + cl);
  const float4 zc0 = FOLD ? ld4(sm + cl) : make_float4(-kmu.x, -kmu.y, -kmu.z, -kmu.w);
  const float4 kbe = fma4(ksc, kmu, ld4(p.h1 + cl));     // u = (z - mean) * scale + (shift + scale * mean)
  const int xact = p.xact, act1 = p.act1;
  float4 gw[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) gw[t] = zero4();
  // the nine taps are read from LDS where they are used (36 registers that the two block-input register sets need more)
  const float* wtp = s_wd + cl;
  auto wtap = [&](int t) { return ld4(wtp + t * C); };
  float4 s1 = zero4(), s2 = zero4();                  // sum g', sum g' (z - mean)
  const float* xn_ = p.x + (size_t)n * H * W * p.ldx;
  const float* dyn = p.dy + (size_t)n * Ho * Wo * p.lddy + cl;
  const size_t rowpitch = (size_t)W * p.ldx;
  const int s0 = band * p.band;

  // a = act(u), g' = da * act'(u) into the sums; returns a
  auto point = [&](float4 zc, float4 da, bool masked, bool valid) {
    const float4 u = fma4(zc, ksc, kbe);
    float4 a = irb_act4<ACT>(u, act1);
    float4 d = irb_act_grad_mul4<ACT>(da, u, act1);
    if (masked) { a = irb_sel4(valid, a); d = irb_sel4(valid, d); }
    s1 = add4(s1, d);
    s2 = fma4(d, zc, s2);
    return a;
  };
  const irb_f4 zinit = {zc0.x, zc0.y, zc0.z, zc0.w};

  if constexpr (S == 2) {
    const int NO = (H + p.pad_t + 1) / 2;
    const int s1e = s0 + p.band < NO ? s0 + p.band : NO;
    const int cs = 16 * seg + j;
    const int ixa = 2 * cs - p.pad_l, ixb = ixa + 1;
    const bool va = ixa >= 0 && ixa < W, vb = ixb < W;
    const bool edge = __any(!va) || __any(!vb);
    const int ixac = ixa < 0 ? 0 : (ixa >= W ? W - 1 : ixa), ixbc = ixb >= W ? W - 1 : ixb;
    const int offa = ixac * p.ldx + q * KQ, offb = ixbc * p.ldx + q * KQ;
    const bool oka = cs < Wo, okb = cs >= 1 && cs - 1 < Wo;
    const int doa = (oka ? cs : 0) * p.lddy, dob = (okb ? cs - 1 : 0) * p.lddy;
    auto loadD = [&](int o, float4& A, float4& B) {
      const bool okr = o >= 0 && o < Ho;
      const float* rp = dyn + (size_t)(okr ? o : 0) * Wo * p.lddy;
      A = irb_sel4(okr && oka, ld4(rp + doa));
      B = irb_sel4(okr && okb, ld4(rp + dob));
    };
    auto load_x2 = [&](int iy, float (&ra)[KQ], float (&rb)[KQ]) {
      const int iyc = iy < 0 ? 0 : (iy >= H ? H - 1 : iy);
      const float* rp = xn_ + (size_t)iyc * rowpitch;
      irb_load_x<K>(rp + offa, ra);
      irb_load_x<K>(rp + offb, rb);
    };
    // one row step; the next step's operands are requested at its start and handed over at its end (two resident waves per SIMD
    // measure faster than one wave with alternating register sets: 117 against 140 us on the 257 x 257 block)
    float4 DAp, DBp, DA, DB, nA, nB;
    float x0[KQ], x1[KQ], x2[KQ], x3[KQ], n0[KQ], n1[KQ], n2[KQ], n3[KQ];
    loadD(s0 - 1, DAp, DBp);
    load_x2(2 * s0 - p.pad_t, n0, n1);
    load_x2(2 * s0 - p.pad_t + 1, n2, n3);
    loadD(s0, nA, nB);
    for (int o = s0; o < s1e; ++o) {
#pragma unroll
      for (int s = 0; s < KQ; ++s) { x0[s] = n0[s]; x1[s] = n1[s]; x2[s] = n2[s]; x3[s] = n3[s]; }
      DA = nA; DB = nB;
      const int iyE = 2 * o - p.pad_t, iyO = iyE + 1;
      load_x2(iyE + 2, n0, n1);
      load_x2(iyO + 2, n2, n3);
      loadD(o + 1, nA, nB);
      if constexpr (!FOLD) {
        irb_prologue<K, XACT>(x0, xs, xh, xact); irb_prologue<K, XACT>(x1, xs, xh, xact);
        irb_prologue<K, XACT>(x2, xs, xh, xact); irb_prologue<K, XACT>(x3, xs, xh, xact);
      }
      irb_f4 z0 = zinit, z1 = zinit, z2 = zinit, z3 = zinit;
#pragma unroll
      for (int s = 0; s < KQ; ++s) {
        z0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[s], x0[s], z0, 0, 0, 0);
        z1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[s], x1[s], z1, 0, 0, 0);
        z2 = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[s], x2[s], z2, 0, 0, 0);
        z3 = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[s], x3[s], z3, 0, 0, 0);
      }
      const bool vE = iyE >= 0, vO = iyO < H;          // (wave-uniform)
      const bool me = edge || !vE, mo = edge || !vO;
      {   // (even row, even column): taps (0,0) (0,2) (2,0) (2,2)
        float4 da = mul4(wtap(0), DA);
        da = fma4(wtap(2), DB, da); da = fma4(wtap(6), DAp, da); da = fma4(wtap(8), DBp, da);
        const float4 a = point(irb_f4_to_float4(z0), da, me, vE && va);
        gw[0] = fma4(a, DA, gw[0]); gw[2] = fma4(a, DB, gw[2]); gw[6] = fma4(a, DAp, gw[6]); gw[8] = fma4(a, DBp, gw[8]);
      }
      {   // (even row, odd column): taps (0,1) (2,1)
        float4 da = mul4(wtap(1), DA);
        da = fma4(wtap(7), DAp, da);
        const float4 a = point(irb_f4_to_float4(z1), da, me, vE && vb);
        gw[1] = fma4(a, DA, gw[1]); gw[7] = fma4(a, DAp, gw[7]);
      }
      {   // (odd row, even column): taps (1,0) (1,2)
        float4 da = mul4(wtap(3), DA);
        da = fma4(wtap(5), DB, da);
        const float4 a = point(irb_f4_to_float4(z2), da, mo, vO && va);
        gw[3] = fma4(a, DA, gw[3]); gw[5] = fma4(a, DB, gw[5]);
      }
      {   // (odd row, odd column): tap (1,1)
        const float4 a = point(irb_f4_to_float4(z3), mul4(wtap(4), DA), mo, vO && vb);
        gw[4] = fma4(a, DA, gw[4]);
      }
      DAp = DA; DBp = DB;
    }
  } else {
    const int s1e = s0 + p.band < H ? s0 + p.band : H;
    const int ix = 16 * seg + j;
    const bool vx = ix < W;
    const bool edge = __any(!vx);
    const int offx = (vx ? ix : W - 1) * p.ldx + q * KQ;
    bool okc[3];
    int doc[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int oxx = ix + p.pad_l - c;
      okc[c] = oxx >= 0 && oxx < Wo;
      doc[c] = (okc[c] ? oxx : 0) * p.lddy;
    }
    auto loadD = [&](int oy, float4 (&D)[3]) {
      const bool okr = oy >= 0 && oy < Ho;
      const float* rp = dyn + (size_t)(okr ? oy : 0) * Wo * p.lddy;
#pragma unroll
      for (int c = 0; c < 3; ++c) D[c] = irb_sel4(okr && okc[c], ld4(rp + doc[c]));
    };
    float4 D1[3], D2[3];                               // Dr[c] = dz2[iy + pad_t - r][ix + pad_l - c]
    auto step = [&](int iy, float (&xv)[KQ], float4 (&D0)[3], float (&xn)[KQ], float4 (&Dn)[3]) {
      irb_load_x<K>(xn_ + (size_t)(iy + 1 < H ? iy + 1 : H - 1) * rowpitch + offx, xn);
      loadD(iy + 1 + p.pad_t, Dn);
      if constexpr (!FOLD) irb_prologue<K, XACT>(xv, xs, xh, xact);
      irb_f4 z = zinit;
#pragma unroll
      for (int s = 0; s < KQ; ++s) z = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[s], xv[s], z, 0, 0, 0);
      float4 da = zero4();
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        da = fma4(wtap(c), D0[c], da); da = fma4(wtap(3 + c), D1[c], da); da = fma4(wtap(6 + c), D2[c], da);
      }
      const float4 a = point(irb_f4_to_float4(z), da, edge, vx);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        gw[c] = fma4(a, D0[c], gw[c]); gw[3 + c] = fma4(a, D1[c], gw[3 + c]); gw[6 + c] = fma4(a, D2[c], gw[6 + c]);
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) { D2[c] = D1[c]; D1[c] = D0[c]; }
    };
    loadD(s0 + p.pad_t - 1, D1);
    loadD(s0 + p.pad_t - 2, D2);
    float XA[KQ], XB[KQ];
    float4 DA[3], DB[3];
    irb_load_x<K>(xn_ + (size_t)s0 * rowpitch + offx, XA);
    loadD(s0 + p.pad_t, DA);
    int iy = s0;
    for (; iy + 1 < s1e; iy += 2) {
      step(iy, XA, DA, XB, DB);
      step(iy + 1, XB, DB, XA, DA);
    }
    if (iy < s1e) step(iy, XA, DA, XB, DB);
  }

  // one slab / partial row per (image, segment, band); this wave's 16 channels of it
  float* slab = p.slabs + (size_t)row_id * 9 * C + cl;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const float4 v = irb_row_sum4(gw[t]);
    if (j == 0) st4(slab + (size_t)t * C, v);
  }
  const float4 a = irb_row_sum4(s1), b = mul4(irb_row_sum4(s2), kis);
  if (j == 0) {
    float* prow = p.partials + (size_t)row_id * 2 * C + cl;
    st4(prow, a);
    st4(prow + C, b);
  }
}

// ------------------------------------------------------------------------------------------ pass B
// one WAVE = (image, 16 column steps, band of row steps), ALL channels: the channel-tile loop is unrolled inside the step, the
// expand-kernel gradient (NCT x KT accumulator tiles) and the dx tiles of the step stay in registers, nothing is shared
// between waves until the end (no barrier in the loop: the 12 waves of a CU drift apart and cover each other's MFMA, vector and
// memory phases).  dz1 in the expand layout IS the B operand of dx = W1 dz1 (sum over channels = the register index); for
// gw = dz1^T x (sum over pixels = the lane index) it goes through a wave-private LDS transpose, the four tiles of a step at once.
// The workgroup's four waves add their slabs and front-BatchNorm rows in LDS at the end: one row per workgroup.
template <int K, int NCT, int S, int ACT, int XACT>
__global__ __launch_bounds__(256, (K == 16 ? 2 : 1)) void irb_bwd_data_kernel(IrbParams p) {
  constexpr int KQ = K / 4, KT = (K + 15) / 16, NTL = S == 2 ? 4 : 1, TP = 20, C = 16 * NCT, CP = C + 4;
  constexpr bool FOLD = XACT == DL3P_ACT_NONE;
  extern __shared__ float sm[];
  float* s_w = sm;                                     // depthwise kernel * c0 [9][C]
  float* s_k = s_w + 9 * C;                            // scale, beta', A, B', zc0 [5][C]  (zc = z1 - mean: see pass A)
  float* s_W = s_k + 5 * C;                            // expand kernel [K][C + 4]
  float* s_t = s_W + K * CP;                           // transpose tiles [4 waves][NTL][16][TP]
  for (int i = threadIdx.x; i < C; i += 256) {
    const float c0 = p.coef1[i], c1 = p.coef1[C + i], c2 = p.coef1[2 * C + i], is = p.is1[i], mu = p.mu1[i], sc = p.s1[i];
    s_k[i] = sc;
    s_k[C + i] = fmaf(sc, mu, p.h1[i]);
    s_k[2 * C + i] = -c0 * c2 * is;                    // dz1 = [act'] (w c0) dz2 + A zc + B'
    s_k[3 * C + i] = -c0 * c1;
    float acc = 0.f;
    if (FOLD && p.xh)
      for (int k = 0; k < K; ++k) acc = fmaf(p.w1[(size_t)k * C + i], p.xh[k], acc);
    s_k[4 * C + i] = acc - mu;
#pragma unroll
    for (int t = 0; t < 9; ++t) s_w[t * C + i] = p.wdw[t * C + i] * c0;
  }
  for (int i = threadIdx.x; i < K * C; i += 256) s_W[(i / C) * CP + i % C] = p.w1[i];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 15, q = lane >> 4;
  const int wg = irb_wg_index(blockIdx.x, gridDim.x);
  const int unit = wg * 4 + wave;
  const bool active = unit < p.units;
  int rr = active ? unit : 0;
  const int band = rr % p.nband; rr /= p.nband;
  const int seg = rr % p.nseg;
  const int n = rr / p.nseg;
  const int H = p.H, W = p.W, Ho = p.Ho, Wo = p.Wo;
  const bool want_dx = p.gx != nullptr;
  const int xact = p.xact, act1 = p.act1;

  float xs[KQ], xh[KQ], xsT[KT], xhT[KT];
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
  irb_f4 gw[NCT][KT];
#pragma unroll
  for (int c = 0; c < NCT; ++c)
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) gw[c][kt] = (irb_f4){0.f, 0.f, 0.f, 0.f};
  float4 f0a[KT], f0b[KT];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) { f0a[kt] = zero4(); f0b[kt] = zero4(); }
  float* tbuf = s_t + wave * NTL * 16 * TP;

  const float* xn_ = p.x + (size_t)n * H * W * p.ldx;
  const float* dyn = p.dy + (size_t)n * Ho * Wo * p.lddy + 4 * q;
  const int s0 = band * p.band;
  const int cs = 16 * seg + j;
  int s1e;
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
  if (!active) s1e = s0;
  bool vcol[NTL];
  int ixc[NTL];
  bool edge_c = false;
#pragma unroll
  for (int t = 0; t < NTL; ++t) {
    vcol[t] = ixt[t] >= 0 && ixt[t] < W;
    ixc[t] = ixt[t] < 0 ? 0 : (ixt[t] >= W ? W - 1 : ixt[t]);
    edge_c = edge_c || __any(!vcol[t]);
  }
  // columns of the pixels 4q + s of tile t (transposed operand of the kernel-gradient product)
  int icT[NTL][4];
#pragma unroll
  for (int t = 0; t < NTL; ++t)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int c2 = 16 * seg + 4 * q + s;
      int ic = S == 2 ? 2 * c2 - p.pad_l + (t & 1) : c2;
      icT[t][s] = ic < 0 ? 0 : (ic >= W ? W - 1 : ic);
    }
  // dz2 column offsets (stride 2: cs, cs - 1; stride 1: cs + pad_l - c)
  constexpr int ND = S == 2 ? 2 : 3;
  bool okd[ND];
  size_t offd[ND];
#pragma unroll
  for (int c = 0; c < ND; ++c) {
    const int oxx = S == 2 ? cs - c : cs + p.pad_l - c;
    okd[c] = oxx >= 0 && oxx < Wo;
    offd[c] = (size_t)(okd[c] ? oxx : 0) * p.lddy;
  }

  // the block input of a step (expand layout xv, transposed layout xT): fetched one step ahead -- a step is NCT * 12 * NTL MFMAs long,
  // so what is requested at its start has landed when the registers are handed over at its end
  auto fetch = [&](int st, auto& xv, auto& xT) {
#pragma unroll
    for (int t = 0; t < NTL; ++t) {
      const int iy = S == 2 ? 2 * st - p.pad_t + (t >> 1) : st;
      const int iyc = iy < 0 ? 0 : (iy >= H ? H - 1 : iy);
      const float* rp = xn_ + (size_t)iyc * W * p.ldx;
      irb_load_x<K>(rp + (size_t)ixc[t] * p.ldx + q * KQ, xv[t]);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
          const int k = 16 * kt + j;
          xT[t][kt][s] = rp[(size_t)icT[t][s] * p.ldx + (k < K ? k : 0)];
        }
    }
  };
  // (K = 16: two resident waves per SIMD with the fetch inside the step measure faster than one wave fetching a step ahead: 234 against
  // 256 us on the 257 x 257 block; the wider blocks need the registers of the second wave for their kernel-gradient accumulators)
  constexpr bool AHEAD = K != 16;
  float xv[NTL][KQ], xT[NTL][KT][4], nxv[AHEAD ? NTL : 1][KQ], nxT[AHEAD ? NTL : 1][KT][4];
  if constexpr (AHEAD) { if (s0 < s1e) fetch(s0, nxv, nxT); }
  for (int st = s0; st < s1e; ++st) {
    int iyt[NTL];
    if constexpr (S == 2) { iyt[0] = iyt[1] = 2 * st - p.pad_t; iyt[2] = iyt[3] = 2 * st - p.pad_t + 1; }
    else iyt[0] = st;
    if constexpr (AHEAD) {
#pragma unroll
      for (int t = 0; t < NTL; ++t) {
#pragma unroll
        for (int s = 0; s < KQ; ++s) xv[t][s] = nxv[t][s];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
          for (int s = 0; s < 4; ++s) xT[t][kt][s] = nxT[t][kt][s];
      }
      fetch(st + 1 < s1e ? st + 1 : st, nxv, nxT);
    } else {
      fetch(st, xv, xT);
    }
    bool vt[NTL];
    bool edge = edge_c;
#pragma unroll
    for (int t = 0; t < NTL; ++t) {
      const int iy = iyt[t];
      const bool vr = iy >= 0 && iy < H;
      edge = edge || !vr;
      vt[t] = vr && vcol[t];
    }
    // dz2 rows of this step: stride 2: rows st, st - 1; stride 1: rows st + pad_t - {0, 1, 2}
    constexpr int NR = S == 2 ? 2 : 3;
    size_t rod[NR];
    bool okr[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const int oy = S == 2 ? st - r : st + p.pad_t - r;
      okr[r] = oy >= 0 && oy < Ho;
      rod[r] = (size_t)(okr[r] ? oy : 0) * Wo * p.lddy;
    }
#pragma unroll
    for (int t = 0; t < NTL; ++t) {
      if constexpr (!FOLD) irb_prologue<K, XACT>(xv[t], xs, xh, xact);
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int s = 0; s < 4; ++s) xT[t][kt][s] = irb_act<XACT>(fmaf(xT[t][kt][s], xsT[kt], xhT[kt]), xact);
    }
    irb_f4 dxp[NTL][KT];
#pragma unroll
    for (int t = 0; t < NTL; ++t)
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) dxp[t][kt] = (irb_f4){0.f, 0.f, 0.f, 0.f};

    float4 Dc[NR][ND];
    auto loadDs = [&](int c, float4 (&D)[NR][ND]) {
#pragma unroll
      for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int d = 0; d < ND; ++d) D[r][d] = irb_sel4(okr[r] && okd[d], ld4(dyn + 16 * c + rod[r] + offd[d]));
    };
    loadDs(0, Dc);
#pragma unroll
    for (int c = 0; c < NCT; ++c) {
      const int cc = 16 * c + 4 * q;                  // this lane's 4 channels
      float4 Dn[NR][ND];
      if (c + 1 < NCT) loadDs(c + 1, Dn);             // the next channel tile's dz2, one tile ahead
      float wf[KQ];
      float4 wa[KT];
#pragma unroll
      for (int s = 0; s < KQ; ++s) {
        const float w = s_W[(q * KQ + s) * CP + 16 * c + j];
        wf[s] = FOLD ? w * xs[s] : w;
      }
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) wa[kt] = (16 * kt + j < K) ? ld4(s_W + (16 * kt + j) * CP + cc) : zero4();
      auto w9 = [&](int t) { return ld4(s_w + t * C + cc); };
      float4 da[NTL];
      if constexpr (S == 2) {
        da[0] = mul4(w9(0), Dc[0][0]); da[0] = fma4(w9(2), Dc[0][1], da[0]);
        da[0] = fma4(w9(6), Dc[1][0], da[0]); da[0] = fma4(w9(8), Dc[1][1], da[0]);
        da[1] = mul4(w9(1), Dc[0][0]); da[1] = fma4(w9(7), Dc[1][0], da[1]);
        da[2] = mul4(w9(3), Dc[0][0]); da[2] = fma4(w9(5), Dc[0][1], da[2]);
        da[3] = mul4(w9(4), Dc[0][0]);
      } else {
        da[0] = zero4();
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
          for (int d = 0; d < 3; ++d) da[0] = fma4(w9(3 * r + d), Dc[r][d], da[0]);
      }
      const float4 ksc = ld4(s_k + cc), kbe = ld4(s_k + C + cc), kA = ld4(s_k + 2 * C + cc), kB = ld4(s_k + 3 * C + cc);
      const float4 kz0 = FOLD ? ld4(s_k + 4 * C + cc) : zero4();
      // the expand tiles of this channel tile: MFMA chains interleaved over the step's pixel tiles
      irb_f4 zt[NTL];
#pragma unroll
      for (int t = 0; t < NTL; ++t) zt[t] = (irb_f4){kz0.x, kz0.y, kz0.z, kz0.w};
#pragma unroll
      for (int s = 0; s < KQ; ++s)
#pragma unroll
        for (int t = 0; t < NTL; ++t) zt[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[s], xv[t][s], zt[t], 0, 0, 0);
      float4 dz[NTL];
#pragma unroll
      for (int t = 0; t < NTL; ++t) {
        float4 zc = irb_f4_to_float4(zt[t]);
        if constexpr (!FOLD) {
          const float4 m = ld4(p.mu1 + cc);
          zc = make_float4(zc.x - m.x, zc.y - m.y, zc.z - m.z, zc.w - m.w);
        }
        const float4 u = fma4(zc, ksc, kbe);
        const float4 g = irb_act_grad_mul4<ACT>(da[t], u, act1);       // (the depthwise kernel carries c0)
        dz[t] = add4(g, fma4(kA, zc, kB));
        if (edge) dz[t] = irb_sel4(vt[t], dz[t]);
      }
      if (want_dx) {
        // dx[k][pixel] += sum_c W1[k][c] dz1[pixel][c]: the result registers are the B operand (k-slot q <-> channel 4q + i)
#pragma unroll
        for (int t = 0; t < NTL; ++t)
#pragma unroll
          for (int kt = 0; kt < KT; ++kt) {
            dxp[t][kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[kt].x, dz[t].x, dxp[t][kt], 0, 0, 0);
            dxp[t][kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[kt].y, dz[t].y, dxp[t][kt], 0, 0, 0);
            dxp[t][kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[kt].z, dz[t].z, dxp[t][kt], 0, 0, 0);
            dxp[t][kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[kt].w, dz[t].w, dxp[t][kt], 0, 0, 0);
          }
      }
      // gw[c_local][k] += sum_pixel dz1[pixel][c] x[pixel][k]: dz1 transposed through the wave's LDS tiles
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int t = 0; t < NTL; ++t) {
        float* tb = tbuf + t * 16 * TP + 4 * q * TP + j;
        tb[0] = dz[t].x; tb[TP] = dz[t].y; tb[2 * TP] = dz[t].z; tb[3 * TP] = dz[t].w;
      }
      __builtin_amdgcn_wave_barrier();
      float4 at[NTL];
#pragma unroll
      for (int t = 0; t < NTL; ++t) at[t] = ld4(tbuf + t * 16 * TP + j * TP + 4 * q);   // channel j, pixels 4q .. 4q + 3
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int t = 0; t < NTL; ++t)
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
          gw[c][kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(at[t].x, xT[t][kt][0], gw[c][kt], 0, 0, 0);
          gw[c][kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(at[t].y, xT[t][kt][1], gw[c][kt], 0, 0, 0);
          gw[c][kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(at[t].z, xT[t][kt][2], gw[c][kt], 0, 0, 0);
          gw[c][kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(at[t].w, xT[t][kt][3], gw[c][kt], 0, 0, 0);
        }
      if (c + 1 < NCT) {
#pragma unroll
        for (int r = 0; r < NR; ++r)
#pragma unroll
          for (int d = 0; d < ND; ++d) Dc[r][d] = Dn[r][d];
      }
    }

    if (want_dx) {
      // lane = (pixel j, input channels 16kt + 4q ..)
#pragma unroll
      for (int t = 0; t < NTL; ++t)
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
          const int k = 16 * kt + 4 * q;
          if (vt[t] && k < K) {
            float4 v = irb_f4_to_float4(dxp[t][kt]);
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
    }
  }

  // ---- the workgroup's slab: waves 1 and 3 hand theirs to waves 0 and 2 through LDS, then 2 to 0 (fixed order)
  __syncthreads();                                     // (every wave is done with the constants and its transpose tiles)
  float4* red = reinterpret_cast<float4*>(sm);        // [2][NCT * KT][64]
  constexpr int NG = NCT * KT;
  for (int round = 0; round < 2; ++round) {
    const bool sender = round == 0 ? (wave & 1) : (wave == 2);
    const bool receiver = round == 0 ? !(wave & 1) : (wave == 0);
    const int slot = round == 0 ? (wave >> 1) : 0;
    if (sender) {
#pragma unroll
      for (int c = 0; c < NCT; ++c)
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) red[(slot * NG + c * KT + kt) * 64 + lane] = irb_f4_to_float4(gw[c][kt]);
    }
    __syncthreads();
    if (receiver) {
#pragma unroll
      for (int c = 0; c < NCT; ++c)
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
          const float4 o = red[(slot * NG + c * KT + kt) * 64 + lane];
          gw[c][kt][0] += o.x; gw[c][kt][1] += o.y; gw[c][kt][2] += o.z; gw[c][kt][3] += o.w;
        }
    }
    __syncthreads();
  }
  if (wave == 0) {
    // expand-kernel gradient slab of this workgroup [K][C]: lane = (k = 16kt + j, channels 16c + 4q ..)
    float* slab = p.slabs + (size_t)wg * K * C;
#pragma unroll
    for (int c = 0; c < NCT; ++c)
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        const int k = 16 * kt + j;
        if (k < K) st4(slab + (size_t)k * C + 16 * c + 4 * q, irb_f4_to_float4(gw[c][kt]));
      }
  }
  if (p.partials0) {
    // rows of the BatchNorm in front: sum over the pixel lanes, then over the waves (in wave order)
    float* redf = sm;                                  // [4 waves][2][KT][16]
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      const float4 a = irb_row_sum4(f0a[kt]), b = irb_row_sum4(f0b[kt]);
      if (j == 0) {
        st4(redf + ((wave * 2 + 0) * KT + kt) * 16 + 4 * q, a);
        st4(redf + ((wave * 2 + 1) * KT + kt) * 16 + 4 * q, b);
      }
    }
    __syncthreads();
    if (wave == 0 && lane < 2 * KT * 16) {
      const int v = lane / (KT * 16), kk = lane % (KT * 16);
      float acc = 0.f;
      for (int w = 0; w < 4; ++w) acc += redf[((w * 2 + v) * KT) * 16 + kk];
      if (kk < K) p.partials0[(size_t)wg * 2 * K + (size_t)v * K + kk] = acc;
    }
  }
}

// ------------------------------------------------------------------------------------------ host side
static int irb_bwd_plan(IrbParams& p, int N, int H, int W, int stride, int pad_t, int pad_l, int ncg, int wave_group,
                        int want_units) {
  const int rsteps = stride == 2 ? (H + pad_t + 1) / 2 : H;
  const int csteps = stride == 2 ? (W + pad_l + 1) / 2 : W;
  p.nseg = ceil_div(csteps, 16);
  p.ncg = ncg;
  const long long per = (long long)N * p.nseg * ncg;
  int nband = (int)ceil_div_ll(want_units, per);
  if (nband < 1) nband = 1;
  // rows of the launch: pass A one per (image, segment, band); pass B one per workgroup of `wave_group` units
  const long long max_by_rows = (long long)DL3P_MAX_STAT_ROWS * wave_group / ((long long)N * p.nseg);
  if (max_by_rows < 1) return -1;
  if (nband > max_by_rows) nband = (int)max_by_rows;
  int band = ceil_div(rsteps, nband);
  if (band < 2) band = rsteps < 2 ? rsteps : 2;
  nband = ceil_div(rsteps, band);
  p.band = band;
  p.nband = nband;
  p.units = (int)(per * nband);
  const int rows = wave_group > 1 ? ceil_div(ceil_div(p.units, wave_group), 8) * 8 : N * p.nseg * nband;
  return rows <= DL3P_MAX_STAT_ROWS ? rows : -1;
}

static int g_irb_bwd_units_a = 0, g_irb_bwd_units_b = 0;
extern "C" int dl3p_irb_set_bwd_plan(int want_waves_sums, int want_waves_data) {
  g_irb_bwd_units_a = want_waves_sums;
  g_irb_bwd_units_b = want_waves_data;
  return DL3P_OK;
}
// the four plan knobs as they stand (0 ct, 1 forward waves, 2 pass-A waves, 3 pass-B workgroups): they decide how many partial / slab
// rows a traced launch writes, so an executor records them with its plan and pins them before an eager replay
int dl3p_irb_fwd_plan_knob(int which);
extern "C" int dl3p_irb_get_plan(int which) {
  if (which == 0 || which == 1) return dl3p_irb_fwd_plan_knob(which);
  return which == 2 ? g_irb_bwd_units_a : which == 3 ? g_irb_bwd_units_b : INT_MIN;
}
static int irb_want_a() { return g_irb_bwd_units_a > 0 ? g_irb_bwd_units_a : 8192; }
static int irb_want_b() { return g_irb_bwd_units_b > 0 ? g_irb_bwd_units_b : 6144; }

extern "C" int dl3p_irb_bwd_supported(int N, int H, int W, int K, int C, int k, int stride, int rate, int pad_t, int pad_l, int Ho,
                                      int Wo) {
  if (!dl3p_irb_supported(N, H, W, K, C, k, stride, rate, pad_t, pad_l, Ho, Wo)) return 0;
  if (!((K == 16 && C == 96) || (K == 24 && C == 144) || (K == 32 && C == 192))) return 0;
  IrbParams p = {};
  if (irb_bwd_plan(p, N, H, W, stride, pad_t, pad_l, C / 16, 1, irb_want_a()) < 0) return 0;
  if (irb_bwd_plan(p, N, H, W, stride, pad_t, pad_l, 1, 4, irb_want_b()) < 0) return 0;
  return 1;
}

/* bytes of the slab regions: which = 0 depthwise-kernel slabs of dl3p_irb_bwd_sums ([rows][9][C]), 1 expand-kernel slabs of
 * dl3p_irb_bwd_data ([rows][K][C]) */
extern "C" size_t dl3p_irb_bwd_workspace(int which, int N, int H, int W, int K, int C, int stride, int pad_t, int pad_l) {
  IrbParams p = {};
  if (which == 0) {
    const int rows = irb_bwd_plan(p, N, H, W, stride, pad_t, pad_l, C / 16, 1, irb_want_a());
    return rows < 0 ? 0 : (size_t)rows * 9 * C * sizeof(float);
  }
  const int rows = irb_bwd_plan(p, N, H, W, stride, pad_t, pad_l, 1, 4, irb_want_b());
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

template <int K>
static void irb_bwd_sums_launch(const IrbParams& p, int stride, hipStream_t st) {
  const int grid = ceil_div(ceil_div(p.units, 4), 8) * 8;
  const size_t shm = (size_t)10 * p.C * sizeof(float);
  const bool fast = p.act1 == DL3P_ACT_RELU6 && p.xact == DL3P_ACT_NONE;
  if (stride == 2) {
    if (fast) dl3p_launch(irb_bwd_sums_kernel<K, 2, DL3P_ACT_RELU6, DL3P_ACT_NONE>, dim3(grid), dim3(256), shm, st, p);
    else dl3p_launch(irb_bwd_sums_kernel<K, 2, -1, -1>, dim3(grid), dim3(256), shm, st, p);
  } else {
    if (fast) dl3p_launch(irb_bwd_sums_kernel<K, 1, DL3P_ACT_RELU6, DL3P_ACT_NONE>, dim3(grid), dim3(256), shm, st, p);
    else dl3p_launch(irb_bwd_sums_kernel<K, 1, -1, -1>, dim3(grid), dim3(256), shm, st, p);
  }
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
  const int rows = irb_bwd_plan(p, N, H, W, stride, pad_t, pad_l, C / 16, 1, irb_want_a());
  DL3P_CHECK_ARG((size_t)rows * 9 * C * sizeof(float) <= slab_bytes, "dl3p_irb_bwd_sums: slab region too small");
  if (slab_rows_out) *slab_rows_out = rows;
  hipStream_t st = (hipStream_t)stream;
  if (K == 16) irb_bwd_sums_launch<16>(p, stride, st);
  else if (K == 24) irb_bwd_sums_launch<24>(p, stride, st);
  else irb_bwd_sums_launch<32>(p, stride, st);
  DL3P_CHECK_LAUNCH("dl3p_irb_bwd_sums");
  return DL3P_OK;
}

template <int K, int NCT>
static void irb_bwd_data_launch(const IrbParams& p, int stride, hipStream_t st) {
  const int grid = ceil_div(ceil_div(p.units, 4), 8) * 8;
  const int ntl = stride == 2 ? 4 : 1;
  const int C = 16 * NCT;
  const size_t shm = ((size_t)14 * C + (size_t)K * (C + 4) + 4 * ntl * 16 * 20) * sizeof(float);
  const size_t red = (size_t)2 * NCT * ((K + 15) / 16) * 64 * sizeof(float4);
  const size_t lds = shm > red ? shm : red;
  const bool fast = p.act1 == DL3P_ACT_RELU6 && p.xact == DL3P_ACT_NONE;
  if (stride == 2) {
    if (fast) dl3p_launch(irb_bwd_data_kernel<K, NCT, 2, DL3P_ACT_RELU6, DL3P_ACT_NONE>, dim3(grid), dim3(256), lds, st, p);
    else dl3p_launch(irb_bwd_data_kernel<K, NCT, 2, -1, -1>, dim3(grid), dim3(256), lds, st, p);
  } else {
    if (fast) dl3p_launch(irb_bwd_data_kernel<K, NCT, 1, DL3P_ACT_RELU6, DL3P_ACT_NONE>, dim3(grid), dim3(256), lds, st, p);
    else dl3p_launch(irb_bwd_data_kernel<K, NCT, 1, -1, -1>, dim3(grid), dim3(256), lds, st, p);
  }
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
  const int rows = irb_bwd_plan(p, N, H, W, stride, pad_t, pad_l, 1, 4, irb_want_b());
  DL3P_CHECK_ARG((size_t)rows * K * C * sizeof(float) <= slab_bytes, "dl3p_irb_bwd_data: slab region too small");
  if (slab_rows_out) *slab_rows_out = rows;
  hipStream_t st = (hipStream_t)stream;
  if (K == 16) irb_bwd_data_launch<16, 6>(p, stride, st);
  else if (K == 24) irb_bwd_data_launch<24, 9>(p, stride, st);
  else irb_bwd_data_launch<32, 12>(p, stride, st);
  DL3P_CHECK_LAUNCH("dl3p_irb_bwd_data");
  return DL3P_OK;
}
