// MaxPooling2D behind an explicit ZeroPadding2D (deeplabv3p_resnet50.py:266-267: pool1_pad + MaxPooling2D((3,3),
// strides 2)): the padded taps are real zeros that take part in the maximum, as they do in Keras.  The input may carry
// its producer's BatchNorm + activation as a prologue (bn_conv1 + ReLU), like every other consumer here.
#include "common.h"

namespace {

struct PoolParams {
  const float* x; int ldx;
  const float* scale; const float* shift; int act;
  float* y; int ldy;              // forward output / backward: gradient w.r.t. the (activated) input
  const float* dy; int lddy;
  unsigned char* arg;             // [N][Ho][Wo][C] tap index of each window's first maximum (forward writes, backward reads)
  int accumulate;
  int N, H, W, C, k, stride, pad_t, pad_l, Ho, Wo;
  long long total;
};

__device__ __forceinline__ float4 pool_in(const PoolParams& p, const float* img, int iy, int ix, int c, float4 sc, float4 sh) {
  if (iy < 0 || iy >= p.H || ix < 0 || ix >= p.W) return zero4();            // ZeroPadding2D
  float4 v = ld4(img + ((size_t)iy * p.W + ix) * p.ldx + c);
  if (p.scale) v = fma4(v, sc, sh);
  return act_apply4(v, p.act);
}

// one thread per (output pixel, 4 channels)
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(PoolParams p) {
  const int c4s = p.C / 4;
  for (long long s = (long long)blockIdx.x * 256 + threadIdx.x; s < p.total; s += (long long)gridDim.x * 256) {
    const int c = (int)(s % c4s) * 4;
    long long r = s / c4s;
    const int ox = (int)(r % p.Wo); r /= p.Wo;
    const int oy = (int)(r % p.Ho);
    const int n = (int)(r / p.Ho);
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = zero4();
    if (p.scale) { sc = ld4(p.scale + c); sh = ld4(p.shift + c); }
    const float* img = p.x + (size_t)n * p.H * p.W * p.ldx;
    float4 m = make_float4(-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f);
    uchar4 a = make_uchar4(0, 0, 0, 0);
    for (int ky = 0; ky < p.k; ++ky)
      for (int kx = 0; kx < p.k; ++kx) {
        const float4 v = pool_in(p, img, oy * p.stride - p.pad_t + ky, ox * p.stride - p.pad_l + kx, c, sc, sh);
        const unsigned char t = (unsigned char)(ky * p.k + kx);
        if (v.x > m.x) { m.x = v.x; a.x = t; }             // strict: the first maximum in (ky, kx) order wins
        if (v.y > m.y) { m.y = v.y; a.y = t; }
        if (v.z > m.z) { m.z = v.z; a.z = t; }
        if (v.w > m.w) { m.w = v.w; a.w = t; }
      }
    st4(p.y + (((size_t)n * p.Ho + oy) * p.Wo + ox) * p.ldy + c, m);
    if (p.arg) *reinterpret_cast<uchar4*>(p.arg + (((size_t)n * p.Ho + oy) * p.Wo + ox) * p.C + c) = a;
  }
}

// gather form (deterministic): an input pixel collects dy of every window in which it is the FIRST maximum in
// (ky, kx) scan order
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(PoolParams p) {
  const int c4s = p.C / 4;
  for (long long s = (long long)blockIdx.x * 256 + threadIdx.x; s < p.total; s += (long long)gridDim.x * 256) {
    const int c = (int)(s % c4s) * 4;
    long long r = s / c4s;
    const int ix = (int)(r % p.W); r /= p.W;
    const int iy = (int)(r % p.H);
    const int n = (int)(r / p.H);
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = zero4();
    if (p.scale) { sc = ld4(p.scale + c); sh = ld4(p.shift + c); }
    const float* img = p.x + (size_t)n * p.H * p.W * p.ldx;
    float4 g = zero4();
    if (p.arg) {
      // the forward pass recorded which tap won each window: 2 small loads per window instead of recomputing it
      const int oy_hi = (iy + p.pad_t) / p.stride, ox_hi = (ix + p.pad_l) / p.stride;
      for (int oy = oy_hi; oy >= 0 && oy * p.stride - p.pad_t + p.k > iy; --oy) {
        if (oy >= p.Ho) continue;
        for (int ox = ox_hi; ox >= 0 && ox * p.stride - p.pad_l + p.k > ix; --ox) {
          if (ox >= p.Wo) continue;
          const unsigned char t = (unsigned char)((iy - (oy * p.stride - p.pad_t)) * p.k + ix - (ox * p.stride - p.pad_l));
          const size_t o = ((size_t)n * p.Ho + oy) * p.Wo + ox;
          const uchar4 a = *reinterpret_cast<const uchar4*>(p.arg + o * p.C + c);
          const float4 d = ld4(p.dy + o * p.lddy + c);
          g = make_float4(g.x + (a.x == t ? d.x : 0.f), g.y + (a.y == t ? d.y : 0.f), g.z + (a.z == t ? d.z : 0.f),
                          g.w + (a.w == t ? d.w : 0.f));
        }
      }
      float* o = p.y + (((size_t)n * p.H + iy) * p.W + ix) * p.ldy + c;
      if (p.accumulate) g = add4(g, ld4(o));
      st4(o, g);
      continue;
    }
    const float4 mine = pool_in(p, img, iy, ix, c, sc, sh);
    // windows (oy, ox) with oy*stride - pad_t <= iy < oy*stride - pad_t + k
    const int oy_hi = (iy + p.pad_t) / p.stride, ox_hi = (ix + p.pad_l) / p.stride;
    for (int oy = oy_hi; oy >= 0 && oy * p.stride - p.pad_t + p.k > iy; --oy) {
      if (oy >= p.Ho) continue;
      for (int ox = ox_hi; ox >= 0 && ox * p.stride - p.pad_l + p.k > ix; --ox) {
        if (ox >= p.Wo) continue;
        const int my_ky = iy - (oy * p.stride - p.pad_t), my_kx = ix - (ox * p.stride - p.pad_l);
        // is `mine` the first maximum of this window?  earlier taps must be strictly smaller, later ones not larger
        bool wx = true, wy = true, wz = true, ww = true;
        for (int ky = 0; ky < p.k; ++ky)
          for (int kx = 0; kx < p.k; ++kx) {
            if (ky == my_ky && kx == my_kx) continue;
            const float4 v = pool_in(p, img, oy * p.stride - p.pad_t + ky, ox * p.stride - p.pad_l + kx, c, sc, sh);
            const bool before = ky < my_ky || (ky == my_ky && kx < my_kx);
            wx = wx && (before ? v.x < mine.x : v.x <= mine.x);
            wy = wy && (before ? v.y < mine.y : v.y <= mine.y);
            wz = wz && (before ? v.z < mine.z : v.z <= mine.z);
            ww = ww && (before ? v.w < mine.w : v.w <= mine.w);
          }
        const float4 d = ld4(p.dy + (((size_t)n * p.Ho + oy) * p.Wo + ox) * p.lddy + c);
        g = make_float4(g.x + (wx ? d.x : 0.f), g.y + (wy ? d.y : 0.f), g.z + (wz ? d.z : 0.f), g.w + (ww ? d.w : 0.f));
      }
    }
    float* o = p.y + (((size_t)n * p.H + iy) * p.W + ix) * p.ldy + c;
    if (p.accumulate) g = add4(g, ld4(o));
    st4(o, g);
  }
}

int check(const char* fn, const void* a, int ld, int C) {
  DL3P_CHECK_ARG(a && aligned16(a) && ld % 4 == 0 && ld >= C, "%s: bad tensor layout (ld=%d, C=%d)", fn, ld, C);
  return DL3P_OK;
}

unsigned grid_for(long long total) {
  long long b = (total + 255) / 256;
  const long long cap = (long long)DL3P_NUM_CUS * 16;
  return (unsigned)(b < 1 ? 1 : (b > cap ? cap : b));
}

}  // namespace

extern "C" int dl3p_maxpool2d_fwd(const float* x, int ldx, const float* in_scale, const float* in_shift, int in_act,
                                  float* y, int ldy, uint8_t* argmax, int N, int H, int W, int C, int k, int stride,
                                  int pad_t, int pad_l, int Ho, int Wo, void* stream) {
  DL3P_CHECK_ARG(C > 0 && C % 4 == 0 && N > 0 && k >= 1 && k <= 15 && stride >= 1 && Ho > 0 && Wo > 0, "dl3p_maxpool2d_fwd: bad dims");
  DL3P_CHECK_ARG(!argmax || (uintptr_t)argmax % 4 == 0, "dl3p_maxpool2d_fwd: argmax must be 4-byte aligned");
  int rc = check("dl3p_maxpool2d_fwd", x, ldx, C);
  if (rc) return rc;
  rc = check("dl3p_maxpool2d_fwd", y, ldy, C);
  if (rc) return rc;
  PoolParams p = {};
  p.x = x; p.ldx = ldx; p.scale = in_scale; p.shift = in_shift; p.act = in_act; p.y = y; p.ldy = ldy; p.arg = argmax;
  p.N = N; p.H = H; p.W = W; p.C = C; p.k = k; p.stride = stride; p.pad_t = pad_t; p.pad_l = pad_l; p.Ho = Ho; p.Wo = Wo;
  p.total = (long long)N * Ho * Wo * (C / 4);
  hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(grid_for(p.total)), dim3(256), 0, (hipStream_t)stream, p);
  DL3P_CHECK_LAUNCH("dl3p_maxpool2d_fwd");
  return DL3P_OK;
}

extern "C" int dl3p_maxpool2d_bwd(const float* x, int ldx, const float* in_scale, const float* in_shift, int in_act,
                                  const float* dy, int lddy, const uint8_t* argmax, float* gx, int ldgx, int accumulate,
                                  int N, int H, int W, int C, int k, int stride, int pad_t, int pad_l, int Ho, int Wo,
                                  void* stream) {
  DL3P_CHECK_ARG(C > 0 && C % 4 == 0 && N > 0 && k >= 1 && stride >= 1 && Ho > 0 && Wo > 0, "dl3p_maxpool2d_bwd: bad dims");
  int rc = check("dl3p_maxpool2d_bwd", x, ldx, C);
  if (rc) return rc;
  rc = check("dl3p_maxpool2d_bwd", dy, lddy, C);
  if (rc) return rc;
  rc = check("dl3p_maxpool2d_bwd", gx, ldgx, C);
  if (rc) return rc;
  PoolParams p = {};
  p.x = x; p.ldx = ldx; p.scale = in_scale; p.shift = in_shift; p.act = in_act; p.y = gx; p.ldy = ldgx;
  p.dy = dy; p.lddy = lddy; p.accumulate = accumulate; p.arg = const_cast<uint8_t*>(argmax);
  p.N = N; p.H = H; p.W = W; p.C = C; p.k = k; p.stride = stride; p.pad_t = pad_t; p.pad_l = pad_l; p.Ho = Ho; p.Wo = Wo;
  p.total = (long long)N * H * W * (C / 4);
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(grid_for(p.total)), dim3(256), 0, (hipStream_t)stream, p);
  DL3P_CHECK_LAUNCH("dl3p_maxpool2d_bwd");
  return DL3P_OK;
}
