// Stride-1 depthwise kernels of the bf16 path: strips along a row (included by dw_bf16.hip inside its namespace).
//
// The gather kernels re-load and re-activate every input once per tap (9 or 25 times).  For stride 1 a thread owns a
// strip of TW outputs of one row that are `rate` pixels apart -- an atrous conv is a dense conv on each of the `rate`
// residue classes of the column index, so those TW outputs share their taps: per kernel row the thread loads
// TW + k - 1 inputs (one residue class), applies the BatchNorm + activation prologue to each ONCE and feeds the k x TW
// products from registers.  Loads per output fall from k*k to k (TW + k - 1) / TW (3x3: 9 -> 4.5, 5x5: 25 -> 10) and
// so does the prologue arithmetic, which is what bounds these kernels on the small maps.  Weights stay packed as bf16
// in registers.  Strips are numbered (residue, position) along a row; out-of-range positions are masked.
#pragma once

template <int V> struct bvec_t;
template <> struct bvec_t<4> { typedef bf16x4 type; };
template <> struct bvec_t<8> { typedef bf16x8 type; };

template <int V, bool HS>
__device__ __forceinline__ fvec<V> pro_fast(const fvec<V>& z, const fvec<V>& sc, const fvec<V>& sh, float lo, float hi, int act) {
  fvec<V> o;
#pragma unroll
  for (int i = 0; i < V; ++i) {
    const float u = fmaf(z.v[i], sc.v[i], sh.v[i]);
    float a;
    if (HS) {
      // the arithmetic of act_apply for the hard-swish family, so that both kernels round the same value
      const float t = fminf(fmaxf(u + 3.f, 0.f), 6.f) * (1.f / 6.f);
      a = act == DL3P_ACT_HSWISH ? u * t : t;
    } else {
      a = fminf(fmaxf(u, lo), hi);
    }
    o.v[i] = bf16_round(a);
  }
  return o;
}

struct StripGeo { int spr, nsr; };      // strips per residue class, strips per row
static inline StripGeo strip_geo(int Wo, int rate, int TW) {
  StripGeo g;
  const int nres = rate < Wo ? rate : Wo;
  const int tmax = ceil_div(Wo, rate);
  g.spr = ceil_div(tmax, TW);
  g.nsr = g.spr * nres;
  return g;
}

template <int V, int KS, int TW, bool HS>
__global__ __launch_bounds__(256) void dwb_fwd_strip(DwB p, StripGeo geo) {
  typedef typename bvec_t<V>::type bvec;
  const int slab = blockIdx.x / p.nbx, bx = blockIdx.x - slab * p.nbx;
  const int pl = threadIdx.x / p.cs, cl = threadIdx.x - pl * p.cs;
  const bool active = pl < p.px;
  const int cbase = slab * p.cs * V, c = cbase + cl * V;
  fvec<V> st[2] = {fzero<V>(), fzero<V>()};
  if (active) {
    const fvec<V> sc = ldv_f32_or<V>(p.scale, c, 1.f), sh = ldv_f32_or<V>(p.shift, c, 0.f);
    const float lo = p.act == DL3P_ACT_NONE ? -DL3P_INF : 0.f;
    const float hi = (p.act == DL3P_ACT_NONE || p.act == DL3P_ACT_RELU) ? DL3P_INF : 6.f;
    bvec wq[KS * KS];
#pragma unroll
    for (int i = 0; i < KS * KS; ++i) wq[i] = *reinterpret_cast<const bvec*>(p.w + (size_t)i * p.C + c);
    const int r = p.rate;
    const XcdRange rg = xcd_range(p.total, bx, p.nbx, p.px, pl);
    for (int s = rg.begin; s < rg.end; s += rg.step) {
      const int sidx = s % geo.nsr;
      const int row = s / geo.nsr;
      const int oy = row % p.Ho, n = row / p.Ho;
      const int rho = sidx / geo.spr;
      const int ox0 = rho + r * (sidx - rho * geo.spr) * TW;
      fvec<V> acc[TW];
#pragma unroll
      for (int j = 0; j < TW; ++j) acc[j] = fzero<V>();
      const bf16* img = p.x + (size_t)n * p.H * p.W * p.ldx + c;
#pragma unroll
      for (int ky = 0; ky < KS; ++ky) {
        const int iy = oy - p.pad_t + ky * r;
        if (iy < 0 || iy >= p.H) continue;                 // a row of zero padding contributes nothing
        const bf16* rowp = img + (size_t)iy * p.W * p.ldx;
        fvec<V> a[TW + KS - 1];
#pragma unroll
        for (int i = 0; i < TW + KS - 1; ++i) {
          const int ix = ox0 - p.pad_l + i * r;
          const bool ok = ix >= 0 && ix < p.W;
          const fvec<V> zin = ldv<V>(rowp + (size_t)min(max(ix, 0), p.W - 1) * p.ldx);
          a[i] = pro_fast<V, HS>(zin, sc, sh, lo, hi, p.act);
          if (!ok) a[i] = fzero<V>();
        }
#pragma unroll
        for (int kx = 0; kx < KS; ++kx) {
          const bvec wk = wq[ky * KS + kx];
#pragma unroll
          for (int j = 0; j < TW; ++j)
#pragma unroll
            for (int e = 0; e < V; ++e) acc[j].v[e] = fmaf(a[j + kx].v[e], (float)wk[e], acc[j].v[e]);
        }
      }
#pragma unroll
      for (int j = 0; j < TW; ++j) {
        const int ox = ox0 + j * r;
        if (ox < p.Wo) {
          stv<V>(p.y + (((size_t)n * p.Ho + oy) * p.Wo + ox) * p.ldy + c, acc[j]);
          if (p.partials) {
#pragma unroll
            for (int e = 0; e < V; ++e) {
              const float q = bf16_round(acc[j].v[e]);
              st[0].v[e] += q;
              st[1].v[e] = fmaf(q, q, st[1].v[e]);
            }
          }
        }
      }
    }
  }
  if (p.partials) dw_block_reduce<V, 2>(st, active, pl, cl, p.cs, p.px, cbase, p.C, p.partials + (size_t)bx * 2 * p.C);
}

// data gradient, stride 1: gx[iy, ix] = sum_taps dy[iy + pad_t - ky r, ix + pad_l - kx r] w[ky, kx]: the same strip walk
// over dy with the kernel mirrored, no prologue
template <int V, int KS, int TW>
__global__ __launch_bounds__(256) void dwb_bwd_data_strip(DwB p, StripGeo geo) {
  typedef typename bvec_t<V>::type bvec;
  const int slab = blockIdx.x / p.nbx, bx = blockIdx.x - slab * p.nbx;
  const int pl = threadIdx.x / p.cs, cl = threadIdx.x - pl * p.cs;
  if (pl >= p.px) return;
  const int c = (slab * p.cs + cl) * V;
  bvec wq[KS * KS];
#pragma unroll
  for (int i = 0; i < KS * KS; ++i) wq[i] = *reinterpret_cast<const bvec*>(p.w + (size_t)i * p.C + c);
  const int r = p.rate;
  const XcdRange rg = xcd_range(p.total, bx, p.nbx, p.px, pl);
  for (int s = rg.begin; s < rg.end; s += rg.step) {
    const int sidx = s % geo.nsr;
    const int row = s / geo.nsr;
    const int iy = row % p.H, n = row / p.H;
    const int rho = sidx / geo.spr;
    const int ix0 = rho + r * (sidx - rho * geo.spr) * TW;
    fvec<V> acc[TW];
#pragma unroll
    for (int j = 0; j < TW; ++j) acc[j] = fzero<V>();
    const bf16* gimg = p.dy + (size_t)n * p.Ho * p.Wo * p.lddy + c;
#pragma unroll
    for (int ky = 0; ky < KS; ++ky) {
      const int oy = iy + p.pad_t - ky * r;
      if (oy < 0 || oy >= p.Ho) continue;
      const bf16* rowp = gimg + (size_t)oy * p.Wo * p.lddy;
      // gx[ix0 + j r] needs dy[ix0 + j r + pad_l - kx r] = d[j + (KS - 1 - kx)], d[i] = dy[ix0 + pad_l - (KS - 1) r + i r]
      fvec<V> d[TW + KS - 1];
#pragma unroll
      for (int i = 0; i < TW + KS - 1; ++i) {
        const int ox = ix0 + p.pad_l - (KS - 1) * r + i * r;
        const bool ok = ox >= 0 && ox < p.Wo;
        d[i] = ldv<V>(rowp + (size_t)min(max(ox, 0), p.Wo - 1) * p.lddy);
        if (!ok) d[i] = fzero<V>();
      }
#pragma unroll
      for (int kx = 0; kx < KS; ++kx) {
        const bvec wk = wq[ky * KS + kx];
#pragma unroll
        for (int j = 0; j < TW; ++j)
#pragma unroll
          for (int e = 0; e < V; ++e) acc[j].v[e] = fmaf(d[j + KS - 1 - kx].v[e], (float)wk[e], acc[j].v[e]);
      }
    }
#pragma unroll
    for (int j = 0; j < TW; ++j) {
      const int ix = ix0 + j * r;
      if (ix < p.W) {
        bf16* o = p.y + (((size_t)n * p.H + iy) * p.W + ix) * p.ldy + c;
        if (p.accumulate) {
          const fvec<V> old = ldv<V>(o);
#pragma unroll
          for (int e = 0; e < V; ++e) acc[j].v[e] += old.v[e];
        }
        stv<V>(o, acc[j]);
      }
    }
  }
}

// weight gradient, stride 1: gw[ky, kx] += sum_j a[ox0 + (j + kx) r - pad_l] * dy[ox0 + j r] over the strip
template <int KS, int TW, bool HS>
__global__ __launch_bounds__(256) void dwb_bwd_weight_strip(DwB p, StripGeo geo) {
  constexpr int V = 4;
  const int slab = blockIdx.x / p.nbx, bx = blockIdx.x - slab * p.nbx;
  const int pl = threadIdx.x / p.cs, cl = threadIdx.x - pl * p.cs;
  const bool active = pl < p.px;
  const int cbase = slab * p.cs * V, c = cbase + cl * V;
  fvec<V> acc[KS * KS];
#pragma unroll
  for (int i = 0; i < KS * KS; ++i) acc[i] = fzero<V>();
  if (active) {
    const fvec<V> sc = ldv_f32_or<V>(p.scale, c, 1.f), sh = ldv_f32_or<V>(p.shift, c, 0.f);
    const float lo = p.act == DL3P_ACT_NONE ? -DL3P_INF : 0.f;
    const float hi = (p.act == DL3P_ACT_NONE || p.act == DL3P_ACT_RELU) ? DL3P_INF : 6.f;
    const int r = p.rate;
    const XcdRange rg = xcd_range(p.total, bx, p.nbx, p.px, pl);
    for (int s = rg.begin; s < rg.end; s += rg.step) {
      const int sidx = s % geo.nsr;
      const int row = s / geo.nsr;
      const int oy = row % p.Ho, n = row / p.Ho;
      const int rho = sidx / geo.spr;
      const int ox0 = rho + r * (sidx - rho * geo.spr) * TW;
      fvec<V> g[TW];
#pragma unroll
      for (int j = 0; j < TW; ++j) {
        const int ox = ox0 + j * r;
        g[j] = ldv<V>(p.dy + (((size_t)n * p.Ho + oy) * p.Wo + min(ox, p.Wo - 1)) * p.lddy + c);
        if (ox >= p.Wo) g[j] = fzero<V>();
      }
      const bf16* img = p.x + (size_t)n * p.H * p.W * p.ldx + c;
#pragma unroll
      for (int ky = 0; ky < KS; ++ky) {
        const int iy = oy - p.pad_t + ky * r;
        if (iy < 0 || iy >= p.H) continue;
        const bf16* rowp = img + (size_t)iy * p.W * p.ldx;
        fvec<V> a[TW + KS - 1];
#pragma unroll
        for (int i = 0; i < TW + KS - 1; ++i) {
          const int ix = ox0 - p.pad_l + i * r;
          const bool ok = ix >= 0 && ix < p.W;
          a[i] = pro_fast<V, HS>(ldv<V>(rowp + (size_t)min(max(ix, 0), p.W - 1) * p.ldx), sc, sh, lo, hi, p.act);
          if (!ok) a[i] = fzero<V>();
        }
#pragma unroll
        for (int kx = 0; kx < KS; ++kx)
#pragma unroll
          for (int j = 0; j < TW; ++j)
#pragma unroll
            for (int e = 0; e < V; ++e) acc[ky * KS + kx].v[e] = fmaf(a[j + kx].v[e], g[j].v[e], acc[ky * KS + kx].v[e]);
      }
    }
  }
  dw_block_reduce<V, KS * KS>(acc, active, pl, cl, p.cs, p.px, cbase, p.C, p.partials + (size_t)bx * KS * KS * p.C);
}
