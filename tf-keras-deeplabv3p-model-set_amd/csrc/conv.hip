// Small dense convolution (the RGB stem and Xception's entry_flow_conv1_2) for gfx950.
//
// Replaces Conv2D(32,3,strides=2) at /root/reference deeplabv3p_mobilenetv2.py:101,
// deeplabv3p_xception.py:119-123, deeplabv3p_mobilenetv3.py:345.  K = k*k*Cin is tiny for the stem
// (27), so this is a direct convolution: a thread owns 4 output channels of one output pixel; the
// 8..16 channel lanes of a pixel share the input taps through L1.  Persistent workgroups emit the BN
// statistics as one partial row each (see dwconv.hip); the XCD-aware split keeps an image's taps
// in one L2.  Weight gradients use the same decomposition with register accumulators per (tap,ci)
// chunk and the deterministic slab reduce.
#include "common.h"

struct ConvParams {
  const float* x; int ldx; const float* scale; const float* shift; int act;
  const float* w; float* y; int ldy; const float* dy; int lddy;
  float* partials;
  int N, H, W, Cin, Cout, Ho, Wo, k, stride, rate, pad_t, pad_l;
  int c4s, px, nslab, nbx;
  long long total;
  int accumulate;
  int chunk0, chunk_n;  // (tap,ci) range handled by this wgrad launch
};

__global__ __launch_bounds__(256) void conv_fwd_kernel(ConvParams p) {
  const int b = blockIdx.x;
  const int slab = b / p.nbx;
  const int bx = b - slab * p.nbx;
  const int pl = threadIdx.x / p.c4s;
  const int cl = threadIdx.x - pl * p.c4s;
  const bool active = pl < p.px;
  const int cbase4 = slab * p.c4s;
  const int co = (cbase4 + cl) * 4;
  float4 st[2] = {zero4(), zero4()};
  if (active) {
    XcdRange r = xcd_range(p.total, bx, p.nbx, p.px, pl);
    for (int s = r.begin; s < r.end; s += r.step) {
      const int ox = s % p.Wo;
      const int row = s / p.Wo;
      const int oy = row % p.Ho;
      const int n = row / p.Ho;
      float4 acc = zero4();
      for (int ky = 0; ky < p.k; ++ky) {
        const int iy = oy * p.stride - p.pad_t + ky * p.rate;
        if (iy < 0 || iy >= p.H) continue;
        for (int kx = 0; kx < p.k; ++kx) {
          const int ix = ox * p.stride - p.pad_l + kx * p.rate;
          if (ix < 0 || ix >= p.W) continue;
          const float* xp = p.x + (((size_t)n * p.H + iy) * p.W + ix) * p.ldx;
          const float* wp = p.w + (size_t)(ky * p.k + kx) * p.Cin * p.Cout + co;
          for (int ci = 0; ci < p.Cin; ++ci) {
            float a = xp[ci];
            if (p.scale) a = fmaf(a, p.scale[ci], p.shift[ci]);
            a = act_apply(a, p.act);
            const float4 wv = ld4(wp + (size_t)ci * p.Cout);
            acc.x = fmaf(a, wv.x, acc.x); acc.y = fmaf(a, wv.y, acc.y);
            acc.z = fmaf(a, wv.z, acc.z); acc.w = fmaf(a, wv.w, acc.w);
          }
        }
      }
      st4(p.y + (((size_t)n * p.Ho + oy) * p.Wo + ox) * p.ldy + co, acc);
      st[0] = add4(st[0], acc);
      st[1] = fma4(acc, acc, st[1]);
    }
  }
  if (p.partials) block_reduce_store<2>(st, active, pl, cl, p.c4s, p.px, cbase4, p.Cout, p.partials + (size_t)bx * 2 * p.Cout);
}

#define WCHUNK 32
__global__ __launch_bounds__(256) void conv_bwd_weight_kernel(ConvParams p) {
  const int b = blockIdx.x;
  const int slab = b / p.nbx;
  const int bx = b - slab * p.nbx;
  const int pl = threadIdx.x / p.c4s;
  const int cl = threadIdx.x - pl * p.c4s;
  const bool active = pl < p.px;
  const int cbase4 = slab * p.c4s;
  const int co = (cbase4 + cl) * 4;
  float4 acc[WCHUNK];
#pragma unroll
  for (int i = 0; i < WCHUNK; ++i) acc[i] = zero4();
  if (active) {
    XcdRange r = xcd_range(p.total, bx, p.nbx, p.px, pl);
    for (int s = r.begin; s < r.end; s += r.step) {
      const int ox = s % p.Wo;
      const int row = s / p.Wo;
      const int oy = row % p.Ho;
      const int n = row / p.Ho;
      const float4 g = ld4(p.dy + (((size_t)n * p.Ho + oy) * p.Wo + ox) * p.lddy + co);
#pragma unroll
      for (int i = 0; i < WCHUNK; ++i) {
        const int e = p.chunk0 + i;          // flat (tap, ci)
        if (i < p.chunk_n) {
          const int tap = e / p.Cin, ci = e - tap * p.Cin;
          const int ky = tap / p.k, kx = tap - ky * p.k;
          const int iy = oy * p.stride - p.pad_t + ky * p.rate;
          const int ix = ox * p.stride - p.pad_l + kx * p.rate;
          if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) {
            float a = p.x[(((size_t)n * p.H + iy) * p.W + ix) * p.ldx + ci];
            if (p.scale) a = fmaf(a, p.scale[ci], p.shift[ci]);
            a = act_apply(a, p.act);
            acc[i].x = fmaf(a, g.x, acc[i].x); acc[i].y = fmaf(a, g.y, acc[i].y);
            acc[i].z = fmaf(a, g.z, acc[i].z); acc[i].w = fmaf(a, g.w, acc[i].w);
          }
        }
      }
    }
  }
  // partial row layout [WCHUNK][Cout]
  block_reduce_store<WCHUNK>(acc, active, pl, cl, p.c4s, p.px, cbase4, p.Cout,
                             p.partials + (size_t)bx * WCHUNK * p.Cout);
}

// gx[n,iy,ix,ci] (+)= sum_{taps,co} dy[n,oy,ox,co] * w[ky,kx,ci,co]; one thread per (pixel, ci)
__global__ __launch_bounds__(256) void conv_bwd_data_kernel(ConvParams p) {
  const long long total = (long long)p.N * p.H * p.W * p.Cin;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int ci = (int)(i % p.Cin);
    long long pix = i / p.Cin;
    const int ix = (int)(pix % p.W);
    pix /= p.W;
    const int iy = (int)(pix % p.H);
    const int n = (int)(pix / p.H);
    float acc = 0.f;
    for (int ky = 0; ky < p.k; ++ky) {
      const int ty = iy + p.pad_t - ky * p.rate;
      if (ty < 0 || ty % p.stride) continue;
      const int oy = ty / p.stride;
      if (oy >= p.Ho) continue;
      for (int kx = 0; kx < p.k; ++kx) {
        const int tx = ix + p.pad_l - kx * p.rate;
        if (tx < 0 || tx % p.stride) continue;
        const int ox = tx / p.stride;
        if (ox >= p.Wo) continue;
        const float* g = p.dy + (((size_t)n * p.Ho + oy) * p.Wo + ox) * p.lddy;
        const float* wv = p.w + ((size_t)(ky * p.k + kx) * p.Cin + ci) * p.Cout;
        for (int co = 0; co < p.Cout; co += 4) {
          const float4 gg = ld4(g + co), ww = ld4(wv + co);
          acc = fmaf(gg.x, ww.x, acc); acc = fmaf(gg.y, ww.y, acc);
          acc = fmaf(gg.z, ww.z, acc); acc = fmaf(gg.w, ww.w, acc);
        }
      }
    }
    float* o = p.y + (((size_t)n * p.H + iy) * p.W + ix) * p.ldy + ci;
    *o = p.accumulate ? *o + acc : acc;
  }
}

static int conv_check(const char* fn, int Cin, int Cout, int k, const void* w) {
  DL3P_CHECK_ARG(Cin > 0 && Cout > 0 && Cout % 4 == 0, "%s: Cout=%d must be a positive multiple of 4", fn, Cout);
  DL3P_CHECK_ARG(k >= 1 && k <= 7, "%s: kernel size %d unsupported", fn, k);
  DL3P_CHECK_ARG(w && aligned16(w), "%s: weight pointer must be 16-byte aligned", fn);
  return DL3P_OK;
}

extern "C" int dl3p_conv2d_fwd(const float* x, int ldx, const float* in_scale, const float* in_shift, int in_act,
                               const float* w, float* y, int ldy, float* stat_partials, int* rows_out, int N, int H,
                               int W, int Cin, int Cout, int k, int stride, int rate, int pad_t, int pad_l, int Ho,
                               int Wo, void* stream) {
  int rc = conv_check("dl3p_conv2d_fwd", Cin, Cout, k, w);
  if (rc) return rc;
  DL3P_CHECK_ARG(x && y && ldx >= Cin && ldy >= Cout && ldy % 4 == 0 && aligned16(y), "dl3p_conv2d_fwd: bad layout");
  ConvParams p = {};
  p.x = x; p.ldx = ldx; p.scale = in_scale; p.shift = in_shift; p.act = in_act; p.w = w; p.y = y; p.ldy = ldy;
  p.partials = stat_partials;
  p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.Ho = Ho; p.Wo = Wo; p.k = k; p.stride = stride;
  p.rate = rate; p.pad_t = pad_t; p.pad_l = pad_l;
  pick_lanes(Cout, &p.c4s, &p.px, &p.nslab);
  p.total = (long long)N * Ho * Wo;
  p.nbx = pick_nbx(p.total, p.px, p.nslab);
  if (rows_out) *rows_out = p.nbx;
  hipLaunchKernelGGL(conv_fwd_kernel, dim3(p.nbx * p.nslab), dim3(256), 0, (hipStream_t)stream, p);
  DL3P_CHECK_LAUNCH("dl3p_conv2d_fwd");
  return DL3P_OK;
}

static int conv_bwdw_rows(int N, int Ho, int Wo, int Cout) {
  int c4s, px, nslab;
  pick_lanes(Cout, &c4s, &px, &nslab);
  return pick_nbx((long long)N * Ho * Wo, px, nslab);
}

extern "C" size_t dl3p_conv2d_bwd_weight_workspace(int N, int Ho, int Wo, int Cin, int Cout, int k) {
  if (Cout <= 0 || Cout % 4) return 0;
  (void)Cin; (void)k;
  return (size_t)conv_bwdw_rows(N, Ho, Wo, Cout) * WCHUNK * Cout * sizeof(float);
}

extern "C" int dl3p_conv2d_bwd_weight(const float* x, int ldx, const float* in_scale, const float* in_shift,
                                      int in_act, const float* dy, int lddy, float* gw, float* workspace,
                                      size_t workspace_bytes, int N, int H, int W, int Cin, int Cout, int k,
                                      int stride, int rate, int pad_t, int pad_l, int Ho, int Wo, void* stream) {
  int rc = conv_check("dl3p_conv2d_bwd_weight", Cin, Cout, k, gw);
  if (rc) return rc;
  DL3P_CHECK_ARG(x && dy && workspace && lddy % 4 == 0 && lddy >= Cout && aligned16(dy) && aligned16(workspace),
                 "dl3p_conv2d_bwd_weight: bad layout");
  const size_t need = dl3p_conv2d_bwd_weight_workspace(N, Ho, Wo, Cin, Cout, k);
  if (workspace_bytes < need) {
    dl3p_set_error("dl3p_conv2d_bwd_weight: workspace %zu < %zu bytes", workspace_bytes, need);
    return DL3P_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  ConvParams p = {};
  p.x = x; p.ldx = ldx; p.scale = in_scale; p.shift = in_shift; p.act = in_act; p.dy = dy; p.lddy = lddy;
  p.partials = workspace;
  p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.Ho = Ho; p.Wo = Wo; p.k = k; p.stride = stride;
  p.rate = rate; p.pad_t = pad_t; p.pad_l = pad_l;
  pick_lanes(Cout, &p.c4s, &p.px, &p.nslab);
  p.total = (long long)N * Ho * Wo;
  p.nbx = pick_nbx(p.total, p.px, p.nslab);
  const int total_e = k * k * Cin;
  for (int e0 = 0; e0 < total_e; e0 += WCHUNK) {
    p.chunk0 = e0;
    p.chunk_n = total_e - e0 < WCHUNK ? total_e - e0 : WCHUNK;
    hipLaunchKernelGGL(conv_bwd_weight_kernel, dim3(p.nbx * p.nslab), dim3(256), 0, st, p);
    DL3P_CHECK_LAUNCH("dl3p_conv2d_bwd_weight");
    // gw is [k*k*Cin][Cout]: rows e0..e0+chunk_n
    rc = dl3p_reduce_rows_strided_impl(workspace, p.nbx, (size_t)WCHUNK * Cout, (size_t)p.chunk_n * Cout,
                                       gw + (size_t)e0 * Cout, 0, st);
    if (rc) return rc;
  }
  return DL3P_OK;
}

extern "C" int dl3p_conv2d_bwd_data(const float* dy, int lddy, const float* w, float* gx, int ldgx, int accumulate,
                                    int N, int H, int W, int Cin, int Cout, int k, int stride, int rate, int pad_t,
                                    int pad_l, int Ho, int Wo, void* stream) {
  int rc = conv_check("dl3p_conv2d_bwd_data", Cin, Cout, k, w);
  if (rc) return rc;
  DL3P_CHECK_ARG(dy && gx && lddy % 4 == 0 && lddy >= Cout && aligned16(dy) && ldgx >= Cin,
                 "dl3p_conv2d_bwd_data: bad layout");
  ConvParams p = {};
  p.dy = dy; p.lddy = lddy; p.w = w; p.y = gx; p.ldy = ldgx; p.accumulate = accumulate;
  p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.Ho = Ho; p.Wo = Wo; p.k = k; p.stride = stride;
  p.rate = rate; p.pad_t = pad_t; p.pad_l = pad_l;
  long long total = (long long)N * H * W * Cin;
  long long blocks = ceil_div_ll(total, 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(conv_bwd_data_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p);
  DL3P_CHECK_LAUNCH("dl3p_conv2d_bwd_data");
  return DL3P_OK;
}

// ------------------------------------------------------------------------------ im2col
// one thread per (output pixel, group of 4 consecutive k): gathers 4 taps, writes one 16-B vector
__global__ __launch_bounds__(256) void im2col_kernel(ConvParams p, int kp) {
  const int k4n = kp / 4;
  const long long total = (long long)p.N * p.Ho * p.Wo * k4n;
  const int kk = p.k * p.k * p.Cin;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int k4 = (int)(i % k4n);
    long long m = i / k4n;
    const int ox = (int)(m % p.Wo);
    long long row = m / p.Wo;
    const int oy = (int)(row % p.Ho);
    const int n = (int)(row / p.Ho);
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int e = k4 * 4 + j;
      float a = 0.f;
      if (e < kk) {
        const int tap = e / p.Cin, ci = e - tap * p.Cin;
        const int ky = tap / p.k, kx = tap - ky * p.k;
        const int iy = oy * p.stride - p.pad_t + ky * p.rate;
        const int ix = ox * p.stride - p.pad_l + kx * p.rate;
        if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) {
          a = p.x[(((size_t)n * p.H + iy) * p.W + ix) * p.ldx + ci];
          if (p.scale) a = fmaf(a, p.scale[ci], p.shift[ci]);
          a = act_apply(a, p.act);
        }
      }
      v[j] = a;
    }
    st4(p.y + (size_t)m * p.ldy + k4 * 4, make_float4(v[0], v[1], v[2], v[3]));
  }
}

extern "C" int dl3p_im2col(const float* x, int ldx, const float* in_scale, const float* in_shift, int in_act,
                           float* col, int ld_col, int N, int H, int W, int Cin, int k, int stride, int rate,
                           int pad_t, int pad_l, int Ho, int Wo, void* stream) {
  DL3P_CHECK_ARG(x && col && aligned16(col) && ld_col % 4 == 0 && ld_col >= k * k * Cin && ldx >= Cin,
                 "dl3p_im2col: bad layout (ld_col=%d)", ld_col);
  ConvParams p = {};
  p.x = x; p.ldx = ldx; p.scale = in_scale; p.shift = in_shift; p.act = in_act; p.y = col; p.ldy = ld_col;
  p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Ho = Ho; p.Wo = Wo; p.k = k; p.stride = stride; p.rate = rate;
  p.pad_t = pad_t; p.pad_l = pad_l;
  long long total = (long long)N * Ho * Wo * (ld_col / 4);
  long long blocks = ceil_div_ll(total, 256);
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(im2col_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, ld_col);
  DL3P_CHECK_LAUNCH("dl3p_im2col");
  return DL3P_OK;
}

// ------------------------------------------------------------------------------ col2im (gather form)
__global__ __launch_bounds__(256) void col2im_kernel(ConvParams p, int kp) {
  const int c4n = p.Cin / 4;
  const long long total = (long long)p.N * p.H * p.W * c4n;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c4 = (int)(i % c4n);
    long long pix = i / c4n;
    const int ix = (int)(pix % p.W);
    pix /= p.W;
    const int iy = (int)(pix % p.H);
    const int n = (int)(pix / p.H);
    float4 acc = zero4();
    for (int ky = 0; ky < p.k; ++ky) {
      const int ty = iy + p.pad_t - ky * p.rate;
      if (ty < 0 || ty % p.stride) continue;
      const int oy = ty / p.stride;
      if (oy >= p.Ho) continue;
      for (int kx = 0; kx < p.k; ++kx) {
        const int tx = ix + p.pad_l - kx * p.rate;
        if (tx < 0 || tx % p.stride) continue;
        const int ox = tx / p.stride;
        if (ox >= p.Wo) continue;
        const size_t m = ((size_t)n * p.Ho + oy) * p.Wo + ox;
        acc = add4(acc, ld4(p.dy + m * kp + (size_t)(ky * p.k + kx) * p.Cin + c4 * 4));
      }
    }
    float* o = p.y + (((size_t)n * p.H + iy) * p.W + ix) * p.ldy + c4 * 4;
    if (p.accumulate) acc = add4(acc, ld4(o));
    st4(o, acc);
  }
}

extern "C" int dl3p_col2im(const float* gcol, int ld_col, float* gx, int ldgx, int accumulate, int N, int H, int W,
                           int Cin, int k, int stride, int rate, int pad_t, int pad_l, int Ho, int Wo, void* stream) {
  DL3P_CHECK_ARG(gcol && gx && aligned16(gcol) && aligned16(gx) && Cin % 4 == 0 && ld_col % 4 == 0 &&
                 ld_col >= k * k * Cin && ldgx % 4 == 0 && ldgx >= Cin, "dl3p_col2im: bad layout");
  ConvParams p = {};
  p.dy = gcol; p.y = gx; p.ldy = ldgx; p.accumulate = accumulate;
  p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Ho = Ho; p.Wo = Wo; p.k = k; p.stride = stride; p.rate = rate;
  p.pad_t = pad_t; p.pad_l = pad_l;
  long long total = (long long)N * H * W * (Cin / 4);
  long long blocks = ceil_div_ll(total, 256);
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(col2im_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, ld_col);
  DL3P_CHECK_LAUNCH("dl3p_col2im");
  return DL3P_OK;
}
