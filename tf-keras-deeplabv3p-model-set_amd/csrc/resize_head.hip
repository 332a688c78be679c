// Bilinear resize (forward / deterministic gather backward) and the fused prediction head
// (pred_resize + Softmax + sparse categorical cross-entropy) for gfx950.
//
// Replaces tf.image.resize(method='bilinear') at /root/reference deeplabv3p/models/layers.py:48-60
// (aspp_resize layers.py:138, decoder_resize layers.py:207, pred_resize model.py:76), the Softmax
// 'pred_mask' layer (model.py:86) and SparseCategoricalCrossEntropy (loss.py:121-156).
// Source coordinates follow TF's half-pixel rule in float32, op for op:
//   scale = in/out; src = (o + 0.5f)*scale - 0.5f; lo = max(floor(src),0); hi = min(ceil(src),in-1);
//   t = src - floor(src).
// HBM-bound; NHWC with 16-B channel vectors where C % 4 == 0.  The backward is the transposed
// operator in gather form (each input pixel sums the output pixels that read it) -> no atomics.
#include "common.h"

struct Lerp { int lo, hi; float t; };
__device__ __forceinline__ Lerp lerp_coeff(int o, float scale, int in_size) {
  const float src = ((float)o + 0.5f) * scale - 0.5f;
  const float fl = floorf(src);
  Lerp r;
  r.lo = max((int)fl, 0);
  r.hi = min((int)ceilf(src), in_size - 1);
  r.t = src - fl;
  return r;
}

struct ResizeParams {
  const float* x; int ldx; float* y; int ldy;
  int N, h, w, C, H, W;
  int c4s, px, nslab, nbx;
  long long total;
  int accumulate;
};

__global__ __launch_bounds__(256) void resize_fwd_kernel(ResizeParams p) {
  const int b = blockIdx.x;
  const int slab = b / p.nbx;
  const int bx = b - slab * p.nbx;
  const int pl = threadIdx.x / p.c4s;
  const int cl = threadIdx.x - pl * p.c4s;
  if (pl >= p.px) return;
  const int c = (slab * p.c4s + cl) * 4;
  const float sy = (float)p.h / (float)p.H, sx = (float)p.w / (float)p.W;
  XcdRange r = xcd_range(p.total, bx, p.nbx, p.px, pl);
  for (int s = r.begin; s < r.end; s += r.step) {
    const int ox = s % p.W;
    const int row = s / p.W;
    const int oy = row % p.H;
    const int n = row / p.H;
    const Lerp ly = lerp_coeff(oy, sy, p.h), lx = lerp_coeff(ox, sx, p.w);
    const float* img = p.x + (size_t)n * p.h * p.w * p.ldx + c;
    const float4 tl = ld4(img + ((size_t)ly.lo * p.w + lx.lo) * p.ldx);
    const float4 tr = ld4(img + ((size_t)ly.lo * p.w + lx.hi) * p.ldx);
    const float4 bl = ld4(img + ((size_t)ly.hi * p.w + lx.lo) * p.ldx);
    const float4 br = ld4(img + ((size_t)ly.hi * p.w + lx.hi) * p.ldx);
    // TF: top = tl + (tr-tl)*tx; bottom = bl + (br-bl)*tx; out = top + (bottom-top)*ty
    float4 o;
#define LERP2(f) { float top = tl.f + (tr.f - tl.f) * lx.t; float bot = bl.f + (br.f - bl.f) * lx.t; o.f = top + (bot - top) * ly.t; }
    LERP2(x) LERP2(y) LERP2(z) LERP2(w)
#undef LERP2
    st4(p.y + (((size_t)n * p.H + oy) * p.W + ox) * p.ldy + c, o);
  }
}

// output-index range that can read input index i
__device__ __forceinline__ void touch_range(int i, float inv_scale, int out_size, int& o0, int& o1) {
  // src(o) in (i-1, i+1)  <=>  o in ((i-0.5)/scale - 0.5, (i+1.5)/scale - 0.5); widen by one for rounding
  const float a = ((float)i - 0.5f) * inv_scale - 0.5f;
  const float b = ((float)i + 1.5f) * inv_scale - 0.5f;
  o0 = max((int)floorf(a) - 1, 0);
  o1 = min((int)ceilf(b) + 1, out_size - 1);
}

__global__ __launch_bounds__(256) void resize_bwd_kernel(ResizeParams p) {
  const int b = blockIdx.x;
  const int slab = b / p.nbx;
  const int bx = b - slab * p.nbx;
  const int pl = threadIdx.x / p.c4s;
  const int cl = threadIdx.x - pl * p.c4s;
  if (pl >= p.px) return;
  const int c = (slab * p.c4s + cl) * 4;
  const float sy = (float)p.h / (float)p.H, sx = (float)p.w / (float)p.W;
  const float isy = (float)p.H / (float)p.h, isx = (float)p.W / (float)p.w;
  XcdRange r = xcd_range(p.total, bx, p.nbx, p.px, pl);
  for (int s = r.begin; s < r.end; s += r.step) {
    const int ix = s % p.w;
    const int row = s / p.w;
    const int iy = row % p.h;
    const int n = row / p.h;
    int y0, y1, x0, x1;
    touch_range(iy, isy, p.H, y0, y1);
    touch_range(ix, isx, p.W, x0, x1);
    // the first/last input index also collects every clamped output (src < 0 or src > in-1)
    if (iy == 0) y0 = 0;
    if (iy == p.h - 1) y1 = p.H - 1;
    if (ix == 0) x0 = 0;
    if (ix == p.w - 1) x1 = p.W - 1;
    float4 acc = zero4();
    const float* gimg = p.x + (size_t)n * p.H * p.W * p.ldx + c;
    for (int oy = y0; oy <= y1; ++oy) {
      const Lerp ly = lerp_coeff(oy, sy, p.h);
      const float wy = (ly.lo == iy ? 1.f - ly.t : 0.f) + (ly.hi == iy ? ly.t : 0.f);
      if (wy == 0.f) continue;
      for (int ox = x0; ox <= x1; ++ox) {
        const Lerp lx = lerp_coeff(ox, sx, p.w);
        const float wx = (lx.lo == ix ? 1.f - lx.t : 0.f) + (lx.hi == ix ? lx.t : 0.f);
        if (wx == 0.f) continue;
        const float wgt = wy * wx;
        const float4 g = ld4(gimg + ((size_t)oy * p.W + ox) * p.ldx);
        acc.x = fmaf(g.x, wgt, acc.x); acc.y = fmaf(g.y, wgt, acc.y);
        acc.z = fmaf(g.z, wgt, acc.z); acc.w = fmaf(g.w, wgt, acc.w);
      }
    }
    float* o = p.y + (((size_t)n * p.h + iy) * p.w + ix) * p.ldy + c;
    if (p.accumulate) acc = add4(acc, ld4(o));
    st4(o, acc);
  }
}

static int resize_check(const char* fn, const float* x, int ldx, const float* y, int ldy, int C) {
  DL3P_CHECK_ARG(x && y, "%s: null pointer", fn);
  DL3P_CHECK_ARG(C > 0 && C % 4 == 0, "%s: C=%d must be a positive multiple of 4", fn, C);
  DL3P_CHECK_ARG(ldx % 4 == 0 && ldx >= C && ldy % 4 == 0 && ldy >= C && aligned16(x) && aligned16(y),
                 "%s: bad layout", fn);
  return DL3P_OK;
}

extern "C" int dl3p_resize_bilinear_fwd(const float* x, int ldx, float* y, int ldy, int N, int h, int w, int C, int H,
                                        int W, void* stream) {
  int rc = resize_check("dl3p_resize_bilinear_fwd", x, ldx, y, ldy, C);
  if (rc) return rc;
  ResizeParams p = {};
  p.x = x; p.ldx = ldx; p.y = y; p.ldy = ldy; p.N = N; p.h = h; p.w = w; p.C = C; p.H = H; p.W = W;
  pick_lanes(C, &p.c4s, &p.px, &p.nslab);
  p.total = (long long)N * H * W;
  p.nbx = pick_nbx(p.total, p.px, p.nslab);
  hipLaunchKernelGGL(resize_fwd_kernel, dim3(p.nbx * p.nslab), dim3(256), 0, (hipStream_t)stream, p);
  DL3P_CHECK_LAUNCH("dl3p_resize_bilinear_fwd");
  return DL3P_OK;
}

extern "C" int dl3p_resize_bilinear_bwd(const float* gy, int ldgy, float* gx, int ldgx, int accumulate, int N, int h,
                                        int w, int C, int H, int W, void* stream) {
  int rc = resize_check("dl3p_resize_bilinear_bwd", gy, ldgy, gx, ldgx, C);
  if (rc) return rc;
  ResizeParams p = {};
  p.x = gy; p.ldx = ldgy; p.y = gx; p.ldy = ldgx; p.N = N; p.h = h; p.w = w; p.C = C; p.H = H; p.W = W;
  p.accumulate = accumulate;
  pick_lanes(C, &p.c4s, &p.px, &p.nslab);
  p.total = (long long)N * h * w;
  p.nbx = pick_nbx(p.total, p.px, p.nslab);
  hipLaunchKernelGGL(resize_bwd_kernel, dim3(p.nbx * p.nslab), dim3(256), 0, (hipStream_t)stream, p);
  DL3P_CHECK_LAUNCH("dl3p_resize_bilinear_bwd");
  return DL3P_OK;
}

// ------------------------------------------------------------------------------ prediction head
// one thread per output pixel: gather the 4 neighbour logit rows (ld = padded C, 16-B loads),
// interpolate, softmax in registers, loss + gradient.  The (N,H,W,C) probability tensor is only
// written when asked for (predict); training never materialises it.
struct HeadParams {
  const float* z; int ldz; const float* labels; int ignore_index; float inv_count;
  float* logits_big; float* probs; float* dlogits; float* loss_partials;
  int N, h, w, C, H, W, ld_big;
  long long total;
};

template <int CP>  // padded channel count held in registers (multiple of 4, >= C)
__global__ __launch_bounds__(256) void head_kernel(HeadParams p) {
  __shared__ float wsum[4];
  const float sy = (float)p.h / (float)p.H, sx = (float)p.w / (float)p.W;
  float loss = 0.f;
  for (long long s = (long long)blockIdx.x * 256 + threadIdx.x; s < p.total; s += (long long)gridDim.x * 256) {
    const int ox = s % p.W;
    const int row = s / p.W;
    const int oy = row % p.H;
    const int n = row / p.H;
    const Lerp ly = lerp_coeff(oy, sy, p.h), lx = lerp_coeff(ox, sx, p.w);
    const float* img = p.z + (size_t)n * p.h * p.w * p.ldz;
    const float* ptl = img + ((size_t)ly.lo * p.w + lx.lo) * p.ldz;
    const float* ptr = img + ((size_t)ly.lo * p.w + lx.hi) * p.ldz;
    const float* pbl = img + ((size_t)ly.hi * p.w + lx.lo) * p.ldz;
    const float* pbr = img + ((size_t)ly.hi * p.w + lx.hi) * p.ldz;
    float v[CP];
#pragma unroll
    for (int c4 = 0; c4 < CP / 4; ++c4) {
      const float4 tl = ld4(ptl + c4 * 4), tr = ld4(ptr + c4 * 4), bl = ld4(pbl + c4 * 4), br = ld4(pbr + c4 * 4);
#define LERP2(f, i) { float top = tl.f + (tr.f - tl.f) * lx.t; float bot = bl.f + (br.f - bl.f) * lx.t; v[c4 * 4 + i] = top + (bot - top) * ly.t; }
      LERP2(x, 0) LERP2(y, 1) LERP2(z, 2) LERP2(w, 3)
#undef LERP2
    }
    float mx = -3.0e38f;
#pragma unroll
    for (int c = 0; c < CP; ++c) if (c < p.C) mx = fmaxf(mx, v[c]);
    float e[CP];
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < CP; ++c) {
      e[c] = c < p.C ? expf(v[c] - mx) : 0.f;
      sum += e[c];
    }
    const float inv = 1.f / sum;
    const size_t obase = (size_t)s * p.C;
    const size_t bbase = (size_t)s * p.ld_big;
    const bool vec = p.ld_big == CP;   // padded rows: 16-B stores, pad channels written as 0
    if (p.logits_big) {
      if (vec) {
#pragma unroll
        for (int c4 = 0; c4 < CP / 4; ++c4)
          st4(p.logits_big + bbase + c4 * 4, make_float4(c4 * 4 + 0 < p.C ? v[c4 * 4 + 0] : 0.f, c4 * 4 + 1 < p.C ? v[c4 * 4 + 1] : 0.f,
                                                         c4 * 4 + 2 < p.C ? v[c4 * 4 + 2] : 0.f, c4 * 4 + 3 < p.C ? v[c4 * 4 + 3] : 0.f));
      } else {
#pragma unroll
        for (int c = 0; c < CP; ++c) if (c < p.C) p.logits_big[bbase + c] = v[c];
      }
    }
    if (p.probs) {
#pragma unroll
      for (int c = 0; c < CP; ++c) if (c < p.C) p.probs[obase + c] = e[c] * inv;
    }
    if (p.labels) {
      const int lab = (int)p.labels[s];
      const bool masked = p.ignore_index != 0 && lab == p.ignore_index;   // loss.py:139 truthiness
      const bool valid = !masked && lab >= 0 && lab < p.C;
      float pt = 0.f;
#pragma unroll
      for (int c = 0; c < CP; ++c) if (c == lab) pt = e[c] * inv;
      const bool unclipped = pt > 1e-7f && pt < 1.f - 1e-7f;
      if (valid) loss += -logf(fminf(fmaxf(pt, 1e-7f), 1.f - 1e-7f));
      if (p.dlogits) {
        const float gs = (valid && unclipped) ? p.inv_count : 0.f;
        float d[CP];
#pragma unroll
        for (int c = 0; c < CP; ++c) d[c] = c < p.C ? gs * (e[c] * inv - (c == lab ? 1.f : 0.f)) : 0.f;
        if (vec) {
#pragma unroll
          for (int c4 = 0; c4 < CP / 4; ++c4)
            st4(p.dlogits + bbase + c4 * 4, make_float4(d[c4 * 4], d[c4 * 4 + 1], d[c4 * 4 + 2], d[c4 * 4 + 3]));
        } else {
#pragma unroll
          for (int c = 0; c < CP; ++c) if (c < p.C) p.dlogits[bbase + c] = d[c];
        }
      }
    }
  }
  if (p.loss_partials) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) loss += __shfl_xor(loss, off);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = loss;
    __syncthreads();
    if (threadIdx.x == 0) p.loss_partials[blockIdx.x] = (wsum[0] + wsum[1] + wsum[2] + wsum[3]) * p.inv_count;
  }
}

extern "C" int dl3p_upsample_softmax_ce(const float* z, int ldz, const float* labels, int ignore_index,
                                        float inv_count, float* logits_big, float* probs, float* dlogits_big,
                                        int ld_big, float* loss_partials, int* rows_out, int N, int h, int w, int C,
                                        int H, int W, void* stream) {
  DL3P_CHECK_ARG(z && aligned16(z) && ldz % 4 == 0, "dl3p_upsample_softmax_ce: logits must be 16-byte aligned, ld %% 4 == 0");
  DL3P_CHECK_ARG(C > 0 && C <= 32 && ldz >= ((C + 3) / 4) * 4, "dl3p_upsample_softmax_ce: C=%d (ld=%d) unsupported", C, ldz);
  DL3P_CHECK_ARG(!labels || loss_partials, "dl3p_upsample_softmax_ce: loss_partials required with labels");
  DL3P_CHECK_ARG((!logits_big && !dlogits_big) || (ld_big >= C && (ld_big % 4 || (aligned16(logits_big) && aligned16(dlogits_big)))),
                 "dl3p_upsample_softmax_ce: bad ld_big=%d", ld_big);
  HeadParams p = {};
  p.z = z; p.ldz = ldz; p.labels = labels; p.ignore_index = ignore_index; p.inv_count = inv_count;
  p.logits_big = logits_big; p.probs = probs; p.dlogits = dlogits_big; p.loss_partials = labels ? loss_partials : nullptr;
  p.N = N; p.h = h; p.w = w; p.C = C; p.H = H; p.W = W; p.ld_big = ld_big;
  p.total = (long long)N * H * W;
  long long blocks = ceil_div_ll(p.total, 256);
  if (blocks > DL3P_MAX_STAT_ROWS) blocks = DL3P_MAX_STAT_ROWS;
  if (rows_out) *rows_out = (int)blocks;
  hipStream_t st = (hipStream_t)stream;
  const int cp = ((C + 3) / 4) * 4;
  const int cpv = cp <= 20 ? 20 : (cp <= 24 ? 24 : 32);
  DL3P_CHECK_ARG(ldz >= cpv, "dl3p_upsample_softmax_ce: ld=%d must be >= %d for C=%d", ldz, cpv, C);
  if (cp <= 20) hipLaunchKernelGGL((head_kernel<20>), dim3((unsigned)blocks), dim3(256), 0, st, p);
  else if (cp <= 24) hipLaunchKernelGGL((head_kernel<24>), dim3((unsigned)blocks), dim3(256), 0, st, p);
  else hipLaunchKernelGGL((head_kernel<32>), dim3((unsigned)blocks), dim3(256), 0, st, p);
  DL3P_CHECK_LAUNCH("dl3p_upsample_softmax_ce");
  return DL3P_OK;
}
