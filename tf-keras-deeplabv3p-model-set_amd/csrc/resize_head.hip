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
#include <stdlib.h>

// exp of the max-shifted logits is the hardware v_exp_f32 (1 ulp on the [-87, 0] arguments a softmax sees); the
// full-range expf costs ~25 instructions x 21 classes per pixel and made the head kernels VALU-bound
struct Lerp { int lo, hi; float t; };
__host__ __device__ __forceinline__ Lerp lerp_coeff(int o, float scale, int in_size) {
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
  int tight;           // resize_bwd_kernel: the eight-column window where it applies (DL3P_RESIZE_TIGHT, default 1)
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

// Upsampling with 256-channel slabs (decoder_resize: 33 x 33 -> 129 x 129 x 256, layers.py:207): one WAVE per (output row, segment of SEG
// output columns, 256-channel slab), a lane per float4 of the slab.  What bounds the pixel-at-a-time kernel above is not its ~150 vector
// instructions per stored float4 and not its four gathers per output (1.1 GB of L2 reads for 272 MB of stores) but that every store
// is followed by loads the next output waits for: vmcnt retires in order, so each wave has ONE 1 KB store in flight per memory round
// trip -- 3.5 TB/s at batch 16 (82.7 us) where a fill kernel writes 7 TB/s.  Three rewrites that kept a load between stores (a strip
// walker, the same with the next strip's loads hoisted, a wave-per-row kernel with the column pair cached) all landed on 3.5-3.6 TB/s.
// Here a segment's loads all come first: the at most six input columns (SEG = 16 above 3.9x, 8 from 2x up) of both input rows go
// through a wave-private LDS tile, and the SEG outputs are then formed from LDS reads only -- SEG stores in flight per wave.
// Column coefficients: lane j computes lerp_coeff of column ox0 + j, the unrolled loop reads them back with v_readlane.
// Same arithmetic per output, same bits.
#define RS_NCOL 6
template <int SEG>
__global__ __launch_bounds__(256) void resize_fwd_seg_kernel(ResizeParams p) {
  extern __shared__ __attribute__((aligned(16))) float rs_lds[];          // [4 waves][2 rows][RS_NCOL][256 floats]
  const float sy = (float)p.h / (float)p.H, sx = (float)p.w / (float)p.W;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int wave = blockIdx.x * 4 + wv;
  const int nwaves = gridDim.x * 4;
  float* tile = rs_lds + (size_t)wv * 2 * RS_NCOL * 256 + lane * 4;
  const int nseg = (p.W + SEG - 1) / SEG, nslab = p.C / 256;
  const int items = p.N * p.H * nslab * nseg;          // (the host rejects >= 2^31)
  for (int item = wave; item < items; item += nwaves) {
    const int seg = item % nseg;
    int rr = item / nseg;
    const int slab = rr % nslab; rr /= nslab;
    const int oy = rr % p.H, n = rr / p.H;
    const Lerp ly = lerp_coeff(oy, sy, p.h);
    const int c = slab * 256 + lane * 4;
    const int ox0 = seg * SEG;
    const Lerp mine = lerp_coeff(min(ox0 + (lane & (SEG - 1)), p.W - 1), sx, p.w);      // lane j < SEG: column ox0 + j
    const int base = __builtin_amdgcn_readlane(mine.lo, 0);
    const float* top = p.x + (((size_t)n * p.h + ly.lo) * p.w) * p.ldx + c;
    const float* bot = p.x + (((size_t)n * p.h + ly.hi) * p.w) * p.ldx + c;
    float4 vt[RS_NCOL], vb[RS_NCOL];
#pragma unroll
    for (int k = 0; k < RS_NCOL; ++k) {
      const size_t col = (size_t)min(base + k, p.w - 1) * p.ldx;
      vt[k] = ld4(top + col);
      vb[k] = ld4(bot + col);
    }
    __builtin_amdgcn_sched_barrier(0);          // all twelve loads in flight together (left alone the compiler pairs each with its LDS store)
#pragma unroll
    for (int k = 0; k < RS_NCOL; ++k) {
      st4(tile + k * 256, vt[k]);
      st4(tile + (RS_NCOL + k) * 256, vb[k]);
    }
    // (wave-private tile: the LDS writes above and the reads below are ordered by lgkmcnt within the wave; no barrier)
    float* yrow = p.y + (((size_t)n * p.H + oy) * p.W) * p.ldy + c;
#pragma unroll
    for (int j = 0; j < SEG; ++j) {
      const int lo = __builtin_amdgcn_readlane(mine.lo, j) - base, hi = __builtin_amdgcn_readlane(mine.hi, j) - base;
      const float t = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mine.t), j));
      const float4 tl = ld4(tile + lo * 256), tr = ld4(tile + hi * 256);
      const float4 bl = ld4(tile + (RS_NCOL + lo) * 256), br = ld4(tile + (RS_NCOL + hi) * 256);
      float4 o;
#define LERP2(f) { float tp = tl.f + (tr.f - tl.f) * t; float bt = bl.f + (br.f - bl.f) * t; o.f = tp + (bt - tp) * ly.t; }
      LERP2(x) LERP2(y) LERP2(z) LERP2(w)
#undef LERP2
      // a column past the row's end repeats the last pixel (same value, same address): the stores stay unconditional
      st4(yrow + (size_t)min(ox0 + j, p.W - 1) * p.ldy, o);
    }
  }
}

// output-index range that can read input index i
__host__ __device__ __forceinline__ void touch_range(int i, float inv_scale, int out_size, int& o0, int& o1) {
  // src(o) in (i-1, i+1)  <=>  o in ((i-0.5)/scale - 0.5, (i+1.5)/scale - 0.5); widen by one for rounding
  const float a = ((float)i - 0.5f) * inv_scale - 0.5f;
  const float b = ((float)i + 1.5f) * inv_scale - 0.5f;
  o0 = max((int)floorf(a) - 1, 0);
  o1 = min((int)ceilf(b) + 1, out_size - 1);
}

#define RB_MAXW 12
#define RB_TIGHT 8
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
    // the output columns that read input column ix lie in an open interval 2 / sx wide: at most RB_TIGHT of them when 2 / sx < RB_TIGHT
    // (33 -> 129: 7.8).  touch_range() is widened for rounding, so the first column with a non-zero weight is found by looking at up
    // to four candidates; every row then costs RB_TIGHT loads instead of RB_MAXW (the skipped terms were exact zeros)
    bool tight = p.tight && 2.f * isx < (float)RB_TIGHT && x1 - x0 < RB_MAXW;
    int xs = x0;
    if (tight) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const Lerp lx = lerp_coeff(min(xs, p.W - 1), sx, p.w);
        if (xs < x1 && lx.lo != ix && lx.hi != ix) ++xs;
      }
      const Lerp lend = lerp_coeff(min(xs + RB_TIGHT, p.W - 1), sx, p.w);
      if (xs + RB_TIGHT <= x1 && (lend.lo == ix || lend.hi == ix)) tight = false;      // (never at these scales: the general path is right anyway)
    }
    if (tight) {
      float wxs[RB_TIGHT];
#pragma unroll
      for (int j = 0; j < RB_TIGHT; ++j) {
        const int ox = min(xs + j, p.W - 1);
        const Lerp lx = lerp_coeff(ox, sx, p.w);
        const float wx = (lx.lo == ix ? 1.f - lx.t : 0.f) + (lx.hi == ix ? lx.t : 0.f);
        wxs[j] = (xs + j <= x1) ? wx : 0.f;
      }
      for (int oy = y0; oy <= y1; ++oy) {
        const Lerp ly = lerp_coeff(oy, sy, p.h);
        const float wy = (ly.lo == iy ? 1.f - ly.t : 0.f) + (ly.hi == iy ? ly.t : 0.f);
        if (wy == 0.f) continue;
        const float* grow = gimg + (size_t)oy * p.W * p.ldx;
        float4 g[RB_TIGHT];
#pragma unroll
        for (int j = 0; j < RB_TIGHT; ++j) g[j] = ld4(grow + (size_t)min(xs + j, x1) * p.ldx);
#pragma unroll
        for (int j = 0; j < RB_TIGHT; ++j) {
          const float wgt = wy * wxs[j];
          acc.x = fmaf(g[j].x, wgt, acc.x); acc.y = fmaf(g[j].y, wgt, acc.y);
          acc.z = fmaf(g[j].z, wgt, acc.z); acc.w = fmaf(g[j].w, wgt, acc.w);
        }
      }
    } else if (x1 - x0 < RB_MAXW) {
      // column weights once per pixel (not once per row): the window is at most RB_MAXW wide for scales <= ~4
      float wxs[RB_MAXW];
#pragma unroll
      for (int j = 0; j < RB_MAXW; ++j) {
        const int ox = min(x0 + j, p.W - 1);
        const Lerp lx = lerp_coeff(ox, sx, p.w);
        const float wx = (lx.lo == ix ? 1.f - lx.t : 0.f) + (lx.hi == ix ? lx.t : 0.f);
        wxs[j] = (x0 + j <= x1) ? wx : 0.f;
      }
      for (int oy = y0; oy <= y1; ++oy) {
        const Lerp ly = lerp_coeff(oy, sy, p.h);
        const float wy = (ly.lo == iy ? 1.f - ly.t : 0.f) + (ly.hi == iy ? ly.t : 0.f);
        if (wy == 0.f) continue;
        // all loads of the row first (columns past x1 re-read the last one with weight 0: fma(g, 0, acc) == acc)
        const float* grow = gimg + (size_t)oy * p.W * p.ldx;
        float4 g[RB_MAXW];
#pragma unroll
        for (int j = 0; j < RB_MAXW; ++j) g[j] = ld4(grow + (size_t)min(x0 + j, x1) * p.ldx);
#pragma unroll
        for (int j = 0; j < RB_MAXW; ++j) {
          const float wgt = wy * wxs[j];
          acc.x = fmaf(g[j].x, wgt, acc.x); acc.y = fmaf(g[j].y, wgt, acc.y);
          acc.z = fmaf(g[j].z, wgt, acc.z); acc.w = fmaf(g[j].w, wgt, acc.w);
        }
      }
    } else {
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
  static const int strip_ok = getenv("DL3P_RESIZE_STRIP") ? atoi(getenv("DL3P_RESIZE_STRIP")) : 1;      // (A/B switch)
  if (strip_ok && W >= 2 * w && C % 256 == 0) {
    // SEG outputs lie between at most RS_NCOL = 6 input columns: floor((SEG - 1) w / W) + 3 <= 6
    const int seg = (15ll * w) / W <= 3 ? 16 : 8;
    const long long items = (long long)N * H * (C / 256) * ((W + seg - 1) / seg);
    if (items < (1ll << 31)) {
      long long wgs = (items + 3) / 4;
      if (wgs > DL3P_NUM_CUS * 3) wgs = DL3P_NUM_CUS * 3;          // 48 KB of LDS per workgroup: three per CU
      const size_t lds = (size_t)4 * 2 * RS_NCOL * 256 * sizeof(float);
      if (seg == 16) hipLaunchKernelGGL(resize_fwd_seg_kernel<16>, dim3((unsigned)wgs), dim3(256), lds, (hipStream_t)stream, p);
      else hipLaunchKernelGGL(resize_fwd_seg_kernel<8>, dim3((unsigned)wgs), dim3(256), lds, (hipStream_t)stream, p);
      DL3P_CHECK_LAUNCH("dl3p_resize_bilinear_fwd");
      return DL3P_OK;
    }
  }
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
  static const int tight = getenv("DL3P_RESIZE_TIGHT") ? atoi(getenv("DL3P_RESIZE_TIGHT")) : 1;      // (A/B switch)
  p.tight = tight;
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
  int loss_kind; const float* class_w; float focal_gamma, focal_alpha;   // DL3P_LOSS_*
  const float* pixel_w;           // Keras sample_weight_mode='temporal': one weight per (image, pixel), or null
  float* logits_big; float* probs; float* dlogits; float* loss_partials;
  int N, h, w, C, H, W, ld_big;
  long long total;
};

template <int CP>  // padded channel count held in registers (multiple of 4, >= C)
__global__ __launch_bounds__(256) void head_kernel(HeadParams p) {
  __shared__ float wsum[4];
  // a wave's 64 pixels are 64*CP contiguous floats of the padded (N,H,W,CP) tensors: rows written by the
  // pixel's thread go through a wave-private LDS slice and leave as 1 KB-contiguous store instructions
  __shared__ __attribute__((aligned(16))) float tr_s[4][64 * (CP + 4)];
  float* trw = tr_s[threadIdx.x >> 6];
  const int lane = threadIdx.x & 63;
  const float sy = (float)p.h / (float)p.H, sx = (float)p.w / (float)p.W;
  float loss = 0.f;
  // whole waves stay in the loop (lanes past the end recompute the last pixel and store nothing): the
  // coalesced gradient store below needs every lane of the wave
  for (long long sw = (long long)blockIdx.x * 256 + threadIdx.x; sw - lane < p.total; sw += (long long)gridDim.x * 256) {
    const bool live = sw < p.total;
    const long long s = live ? sw : p.total - 1;
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
      e[c] = c < p.C ? __expf(v[c] - mx) : 0.f;
      sum += e[c];
    }
    const float inv = 1.f / sum;
    const size_t obase = (size_t)s * p.C;
    const size_t bbase = (size_t)s * p.ld_big;
    const bool vec = p.ld_big == CP;   // padded rows: 16-B stores, pad channels written as 0
    if (p.logits_big && live) {
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
    if (p.probs && live) {
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
      // per-pixel loss l(p_t) and the factor f with d l / d z_c = f * (p_c - [c == y]):
      //   cross entropy (loss.py:121-156)   l = -log(clip(p_t, 1e-7, 1 - 1e-7)),  f = 1 inside the clip range, else 0
      //   class-weighted  (loss.py:159-191) l = -w_y log(p_t)  (no clipping),      f = w_y
      //   focal           (loss.py:63-118)  l = -alpha (1 - p_t)^gamma log(p_t),   p_t clipped to [1e-15, 1 - 1e-15];
      //                                     f = alpha ((1 - p_t)^gamma - gamma (1 - p_t)^(gamma - 1) p_t log p_t)
      float li, f;
      if (p.loss_kind == DL3P_LOSS_WEIGHTED_CE) {
        float wy = 0.f;
        if (valid) wy = p.class_w[lab];
        li = -wy * logf(pt);
        f = wy;
      } else if (p.loss_kind == DL3P_LOSS_FOCAL) {
        const float pc = fmaxf(pt, 1e-15f);               // 1 - 1e-15 rounds to 1 in fp32: the upper clip is a no-op
        const float om = 1.f - pc, lp = logf(pc);
        const float pw1 = om > 0.f ? powf(om, p.focal_gamma - 1.f) : (p.focal_gamma == 1.f ? 1.f : 0.f);
        const float pw = pw1 * om;
        li = -p.focal_alpha * pw * lp;
        f = pt >= 1e-15f ? p.focal_alpha * (pw - p.focal_gamma * pw1 * pc * lp) : 0.f;
      } else {
        const bool unclipped = pt > 1e-7f && pt < 1.f - 1e-7f;
        li = -logf(fminf(fmaxf(pt, 1e-7f), 1.f - 1e-7f));
        f = unclipped ? 1.f : 0.f;
      }
      if (p.pixel_w) {            // the weighted per-pixel losses are averaged over ALL entries, like the unweighted ones
        const float sw = p.pixel_w[s];
        li *= sw;
        f *= sw;
      }
      if (valid && live) loss += li;
      if (p.dlogits) {
        const float gs = valid ? f * p.inv_count : 0.f;
        float d[CP];
#pragma unroll
        for (int c = 0; c < CP; ++c) d[c] = c < p.C ? gs * (e[c] * inv - (c == lab ? 1.f : 0.f)) : 0.f;
        if (vec) {
#pragma unroll
          for (int c4 = 0; c4 < CP / 4; ++c4)
            *reinterpret_cast<float4*>(&trw[lane * (CP + 4) + c4 * 4]) = make_float4(d[c4 * 4], d[c4 * 4 + 1], d[c4 * 4 + 2], d[c4 * 4 + 3]);
          // other lanes' rows are read next: keep the compiler (single-thread view) from hoisting those reads
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          const long long s0 = sw - lane;                     // first pixel of this wave's group
          const int npx = (int)(p.total - s0 < 64 ? p.total - s0 : 64);
#pragma unroll
          for (int k = 0; k < CP / 4; ++k) {
            const int f = lane + 64 * k;
            const int r = f / (CP / 4), cc = f - r * (CP / 4);
            if (r < npx) st4(p.dlogits + (size_t)s0 * CP + (size_t)f * 4, *reinterpret_cast<const float4*>(&trw[r * (CP + 4) + cc * 4]));
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // ... and the next pixel's writes from passing them
        } else {
#pragma unroll
          for (int c = 0; c < CP; ++c) if (c < p.C && live) p.dlogits[bbase + c] = d[c];
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

// More than 32 classes (the reference allows up to 253: train.py:34): a pixel's class vector no longer fits registers.
// One thread per output pixel walks the classes three times (maximum, sum of exponentials, outputs), re-interpolating
// each logit from its 4 neighbours (L1 / L2 hits); same arithmetic as head_kernel, class by class.
__global__ __launch_bounds__(256) void head_kernel_big(HeadParams p) {
  __shared__ float wsum[4];
  const float sy = (float)p.h / (float)p.H, sx = (float)p.w / (float)p.W;
  float loss = 0.f;
  for (long long s = (long long)blockIdx.x * 256 + threadIdx.x; s < p.total; s += (long long)gridDim.x * 256) {
    const int ox = s % p.W;
    const long long row = s / p.W;
    const int oy = (int)(row % p.H), n = (int)(row / p.H);
    const Lerp ly = lerp_coeff(oy, sy, p.h), lx = lerp_coeff(ox, sx, p.w);
    const float* img = p.z + (size_t)n * p.h * p.w * p.ldz;
    const float* ptl = img + ((size_t)ly.lo * p.w + lx.lo) * p.ldz;
    const float* ptr = img + ((size_t)ly.lo * p.w + lx.hi) * p.ldz;
    const float* pbl = img + ((size_t)ly.hi * p.w + lx.lo) * p.ldz;
    const float* pbr = img + ((size_t)ly.hi * p.w + lx.hi) * p.ldz;
    auto logit = [&](int c) {
      const float top = ptl[c] + (ptr[c] - ptl[c]) * lx.t, bot = pbl[c] + (pbr[c] - pbl[c]) * lx.t;
      return top + (bot - top) * ly.t;
    };
    float mx = -3.0e38f;
    for (int c = 0; c < p.C; ++c) mx = fmaxf(mx, logit(c));
    float sum = 0.f;
    for (int c = 0; c < p.C; ++c) sum += __expf(logit(c) - mx);
    const float inv = 1.f / sum;
    const int lab = p.labels ? (int)p.labels[s] : -1;
    const bool masked = p.ignore_index != 0 && lab == p.ignore_index;
    const bool valid = p.labels && !masked && lab >= 0 && lab < p.C;
    float li = 0.f, f = 0.f;
    if (p.labels) {
      const float pt = valid ? __expf(logit(lab) - mx) * inv : 0.f;
      if (p.loss_kind == DL3P_LOSS_WEIGHTED_CE) {
        const float wy = valid ? p.class_w[lab] : 0.f;
        li = -wy * logf(pt);
        f = wy;
      } else if (p.loss_kind == DL3P_LOSS_FOCAL) {
        const float pc = fmaxf(pt, 1e-15f);
        const float om = 1.f - pc, lp = logf(pc);
        const float pw1 = om > 0.f ? powf(om, p.focal_gamma - 1.f) : (p.focal_gamma == 1.f ? 1.f : 0.f);
        const float pw = pw1 * om;
        li = -p.focal_alpha * pw * lp;
        f = pt >= 1e-15f ? p.focal_alpha * (pw - p.focal_gamma * pw1 * pc * lp) : 0.f;
      } else {
        const bool unclipped = pt > 1e-7f && pt < 1.f - 1e-7f;
        li = -logf(fminf(fmaxf(pt, 1e-7f), 1.f - 1e-7f));
        f = unclipped ? 1.f : 0.f;
      }
      if (p.pixel_w) { const float sw = p.pixel_w[s]; li *= sw; f *= sw; }
      if (valid) loss += li;
    }
    const float gs = valid ? f * p.inv_count : 0.f;
    for (int c = 0; c < p.ld_big; ++c) {
      const bool in = c < p.C;
      const float v = in ? logit(c) : 0.f;
      const float pr = in ? __expf(v - mx) * inv : 0.f;
      if (p.logits_big) p.logits_big[(size_t)s * p.ld_big + c] = v;
      if (p.dlogits) p.dlogits[(size_t)s * p.ld_big + c] = in ? gs * (pr - (c == lab ? 1.f : 0.f)) : 0.f;
      if (p.probs && in) p.probs[(size_t)s * p.C + c] = pr;
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
  return dl3p_upsample_softmax_loss(z, ldz, labels, ignore_index, inv_count, DL3P_LOSS_CE, nullptr, 0.f, 0.f, nullptr,
                                    logits_big, probs, dlogits_big, ld_big, loss_partials, rows_out, N, h, w, C, H, W,
                                    stream);
}

extern "C" int dl3p_upsample_softmax_loss(const float* z, int ldz, const float* labels, int ignore_index,
                                          float inv_count, int loss_kind, const float* class_weights, float focal_gamma,
                                          float focal_alpha, const float* pixel_weights, float* logits_big, float* probs,
                                          float* dlogits_big,
                                          int ld_big, float* loss_partials, int* rows_out, int N, int h, int w, int C,
                                          int H, int W, void* stream) {
  DL3P_CHECK_ARG(z && aligned16(z) && ldz % 4 == 0, "dl3p_upsample_softmax_ce: logits must be 16-byte aligned, ld %% 4 == 0");
  DL3P_CHECK_ARG(loss_kind == DL3P_LOSS_CE || loss_kind == DL3P_LOSS_FOCAL || (loss_kind == DL3P_LOSS_WEIGHTED_CE && class_weights),
                 "dl3p_upsample_softmax_loss: bad loss kind %d (weighted CE needs class_weights[C])", loss_kind);
  DL3P_CHECK_ARG(C > 0 && C <= 256 && ldz >= ((C + 3) / 4) * 4, "dl3p_upsample_softmax_ce: C=%d (ld=%d) unsupported", C, ldz);
  DL3P_CHECK_ARG(!labels || loss_partials, "dl3p_upsample_softmax_ce: loss_partials required with labels");
  DL3P_CHECK_ARG((!logits_big && !dlogits_big) || (ld_big >= C && (ld_big % 4 || (aligned16(logits_big) && aligned16(dlogits_big)))),
                 "dl3p_upsample_softmax_ce: bad ld_big=%d", ld_big);
  HeadParams p = {};
  p.z = z; p.ldz = ldz; p.labels = labels; p.ignore_index = ignore_index; p.inv_count = inv_count;
  p.loss_kind = loss_kind; p.class_w = class_weights; p.focal_gamma = focal_gamma; p.focal_alpha = focal_alpha;
  p.pixel_w = pixel_weights;
  p.logits_big = logits_big; p.probs = probs; p.dlogits = dlogits_big; p.loss_partials = labels ? loss_partials : nullptr;
  p.N = N; p.h = h; p.w = w; p.C = C; p.H = H; p.W = W; p.ld_big = ld_big;
  p.total = (long long)N * H * W;
  long long blocks = ceil_div_ll(p.total, 256);
  if (blocks > DL3P_MAX_STAT_ROWS) blocks = DL3P_MAX_STAT_ROWS;
  if (rows_out) *rows_out = (int)blocks;
  hipStream_t st = (hipStream_t)stream;
  if (C > 32) {
    hipLaunchKernelGGL(head_kernel_big, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p);
    DL3P_CHECK_LAUNCH("dl3p_upsample_softmax_ce");
    return DL3P_OK;
  }
  int cp = ((C + 3) / 4) * 4;
  // rows padded further than that (the bf16 graphs pad the class dimension to a multiple of 8: 19 classes -> 24): run the
  // instantiation that matches the row, so that the gradient still leaves as whole 16-byte vectors
  if ((logits_big || dlogits_big) && ld_big % 4 == 0 && ld_big > cp && ld_big <= 32 && ldz >= ld_big) cp = ld_big;
  // one instantiation per padded class count (a multiple of 4 up to 32): the kernel reads exactly the cp floats a
  // pixel of the (N,h,w,cp) logits holds, so 2-class models work like 21-class ones
  switch (cp) {
#define DL3P_HEAD_CASE(CP) case CP: hipLaunchKernelGGL((head_kernel<CP>), dim3((unsigned)blocks), dim3(256), 0, st, p); break;
    DL3P_HEAD_CASE(4) DL3P_HEAD_CASE(8) DL3P_HEAD_CASE(12) DL3P_HEAD_CASE(16) DL3P_HEAD_CASE(20) DL3P_HEAD_CASE(24)
    DL3P_HEAD_CASE(28) DL3P_HEAD_CASE(32)
#undef DL3P_HEAD_CASE
  }
  DL3P_CHECK_LAUNCH("dl3p_upsample_softmax_ce");
  return DL3P_OK;
}

// ------------------------------------------------------------------------------ evaluation head: argmax + confusion matrix
// eval.py:33-36 (`np.argmax(prediction, -1)`) and :368-373 (`generate_matrix`): the class of a pixel is the argmax of
// the upsampled logits (= the argmax of their softmax, except that logits too close for fp32 softmax to tell apart are
// still told apart here; first index on ties, like np.argmax); pixels whose label lies
// in [0, C) add 1 to confusion[label][prediction].  Counts are integers: a workgroup-private LDS histogram flushed
// with 64-bit atomics gives the same matrix in any order.  The (N,H,W,C) probability tensor is never written.
struct EvalParams {
  const float* z; int ldz; const float* labels; int* pred; unsigned long long* cm;
  int N, h, w, C, H, W;
  long long total;
};

template <int CP>
__global__ __launch_bounds__(256) void argmax_confusion_kernel(EvalParams p) {
  extern __shared__ unsigned int hist[];                 // [C][C], only with labels and C <= 64
  const bool use_lds = p.labels && p.cm && p.C <= 64;
  if (use_lds) {
    for (int i = threadIdx.x; i < p.C * p.C; i += 256) hist[i] = 0u;
    __syncthreads();
  }
  const float sy = (float)p.h / (float)p.H, sx = (float)p.w / (float)p.W;
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
    float best = -3.0e38f;
    int arg = 0;
#pragma unroll
    for (int c4 = 0; c4 < CP / 4; ++c4) {
      const float4 tl = ld4(ptl + c4 * 4), tr = ld4(ptr + c4 * 4), bl = ld4(pbl + c4 * 4), br = ld4(pbr + c4 * 4);
      // the same arithmetic as head_kernel, so predict() followed by a host argmax gives the same class
#define LERP2(f, i) { float top = tl.f + (tr.f - tl.f) * lx.t; float bot = bl.f + (br.f - bl.f) * lx.t; \
                      const float v = top + (bot - top) * ly.t; \
                      if (c4 * 4 + i < p.C && v > best) { best = v; arg = c4 * 4 + i; } }
      LERP2(x, 0) LERP2(y, 1) LERP2(z, 2) LERP2(w, 3)
#undef LERP2
    }
    if (p.pred) p.pred[s] = arg;
    if (p.labels && p.cm) {
      const int lab = (int)p.labels[s];
      if (lab >= 0 && lab < p.C) {
        if (use_lds) atomicAdd(&hist[lab * p.C + arg], 1u);
        else atomicAdd(&p.cm[(size_t)lab * p.C + arg], 1ull);
      }
    }
  }
  if (use_lds) {
    __syncthreads();
    for (int i = threadIdx.x; i < p.C * p.C; i += 256)
      if (hist[i]) atomicAdd(&p.cm[i], (unsigned long long)hist[i]);
  }
}

__global__ __launch_bounds__(256) void argmax_confusion_big(EvalParams p) {
  const float sy = (float)p.h / (float)p.H, sx = (float)p.w / (float)p.W;
  for (long long s = (long long)blockIdx.x * 256 + threadIdx.x; s < p.total; s += (long long)gridDim.x * 256) {
    const int ox = s % p.W;
    const long long row = s / p.W;
    const int oy = (int)(row % p.H), n = (int)(row / p.H);
    const Lerp ly = lerp_coeff(oy, sy, p.h), lx = lerp_coeff(ox, sx, p.w);
    const float* img = p.z + (size_t)n * p.h * p.w * p.ldz;
    const float* ptl = img + ((size_t)ly.lo * p.w + lx.lo) * p.ldz;
    const float* ptr = img + ((size_t)ly.lo * p.w + lx.hi) * p.ldz;
    const float* pbl = img + ((size_t)ly.hi * p.w + lx.lo) * p.ldz;
    const float* pbr = img + ((size_t)ly.hi * p.w + lx.hi) * p.ldz;
    float best = -3.0e38f;
    int arg = 0;
    for (int c = 0; c < p.C; ++c) {
      const float top = ptl[c] + (ptr[c] - ptl[c]) * lx.t, bot = pbl[c] + (pbr[c] - pbl[c]) * lx.t;
      const float v = top + (bot - top) * ly.t;
      if (v > best) { best = v; arg = c; }
    }
    if (p.pred) p.pred[s] = arg;
    if (p.labels && p.cm) {
      const int lab = (int)p.labels[s];
      if (lab >= 0 && lab < p.C) atomicAdd(&p.cm[(size_t)lab * p.C + arg], 1ull);
    }
  }
}

extern "C" int dl3p_argmax_confusion(const float* z, int ldz, const float* labels, int32_t* pred_mask,
                                     unsigned long long* confusion, int N, int h, int w, int C, int H, int W,
                                     void* stream) {
  DL3P_CHECK_ARG(z && aligned16(z) && ldz % 4 == 0, "dl3p_argmax_confusion: logits must be 16-byte aligned, ld %% 4 == 0");
  DL3P_CHECK_ARG(C > 0 && C <= 256 && N > 0 && h > 0 && w > 0 && H > 0 && W > 0, "dl3p_argmax_confusion: bad dims (C=%d)", C);
  DL3P_CHECK_ARG(pred_mask || (labels && confusion), "dl3p_argmax_confusion: nothing to produce");
  DL3P_CHECK_ARG(!confusion || labels, "dl3p_argmax_confusion: a confusion matrix needs labels");
  EvalParams p = {};
  p.z = z; p.ldz = ldz; p.labels = labels; p.pred = pred_mask; p.cm = confusion;
  p.N = N; p.h = h; p.w = w; p.C = C; p.H = H; p.W = W;
  p.total = (long long)N * H * W;
  long long blocks = ceil_div_ll(p.total, 256 * 8);      // 8 pixels per thread: few histogram flushes
  if (blocks < 1) blocks = 1;
  if (blocks > 4096) blocks = 4096;
  const int cp = ((C + 3) / 4) * 4;
  DL3P_CHECK_ARG(ldz >= cp, "dl3p_argmax_confusion: ld=%d must be >= %d for C=%d", ldz, cp, C);
  hipStream_t st = (hipStream_t)stream;
  if (C > 32) {
    hipLaunchKernelGGL(argmax_confusion_big, dim3((unsigned)blocks), dim3(256), 0, st, p);
    DL3P_CHECK_LAUNCH("dl3p_argmax_confusion");
    return DL3P_OK;
  }
  const size_t lds = (labels && confusion) ? (size_t)C * C * sizeof(unsigned int) : 0;
  switch (cp) {
#define DL3P_AC_CASE(CP) case CP: hipLaunchKernelGGL((argmax_confusion_kernel<CP>), dim3((unsigned)blocks), dim3(256), lds, st, p); break;
    DL3P_AC_CASE(4) DL3P_AC_CASE(8) DL3P_AC_CASE(12) DL3P_AC_CASE(16) DL3P_AC_CASE(20) DL3P_AC_CASE(24) DL3P_AC_CASE(28)
    DL3P_AC_CASE(32)
#undef DL3P_AC_CASE
  }
  DL3P_CHECK_LAUNCH("dl3p_argmax_confusion");
  return DL3P_OK;
}

// ------------------------------------------------------------------------------ training metrics: per-image class counts
// deeplabv3p/metrics.py:29-46 Jaccard (train.py:140 metrics={'pred_mask': Jaccard}) and :20-26
// sparse_accuracy_ignoring_last_label need, per image n and class c: |label == c & pred == c|, |label == c| and
// |pred == c| over ALL pixels (a pixel with an ignored label still counts towards the union of its predicted class).
// counts[n][3][C] (int32, zeroed by the caller): workgroups stay inside one image and keep an LDS histogram.
__global__ __launch_bounds__(256) void class_counts_kernel(EvalParams p, int* counts, int blocks_per_image) {
  extern __shared__ unsigned int hist[];                 // [3][C]
  for (int i = threadIdx.x; i < 3 * p.C; i += 256) hist[i] = 0u;
  __syncthreads();
  const int n = blockIdx.x / blocks_per_image, b = blockIdx.x - n * blocks_per_image;
  const int HW = p.H * p.W;
  const float sy = (float)p.h / (float)p.H, sx = (float)p.w / (float)p.W;
  const float* img = p.z + (size_t)n * p.h * p.w * p.ldz;
  for (int s = b * 256 + threadIdx.x; s < HW; s += blocks_per_image * 256) {
    const int ox = s % p.W, oy = s / p.W;
    const Lerp ly = lerp_coeff(oy, sy, p.h), lx = lerp_coeff(ox, sx, p.w);
    const float* ptl = img + ((size_t)ly.lo * p.w + lx.lo) * p.ldz;
    const float* ptr = img + ((size_t)ly.lo * p.w + lx.hi) * p.ldz;
    const float* pbl = img + ((size_t)ly.hi * p.w + lx.lo) * p.ldz;
    const float* pbr = img + ((size_t)ly.hi * p.w + lx.hi) * p.ldz;
    float best = -3.0e38f;
    int arg = 0;
    for (int c = 0; c < p.C; ++c) {                      // same arithmetic as head_kernel / argmax_confusion_kernel
      const float top = ptl[c] + (ptr[c] - ptl[c]) * lx.t, bot = pbl[c] + (pbr[c] - pbl[c]) * lx.t;
      const float v = top + (bot - top) * ly.t;
      if (v > best) { best = v; arg = c; }
    }
    const int lab = (int)p.labels[(size_t)n * HW + s];
    atomicAdd(&hist[2 * p.C + arg], 1u);
    if (lab >= 0 && lab < p.C) {
      atomicAdd(&hist[p.C + lab], 1u);
      if (lab == arg) atomicAdd(&hist[lab], 1u);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 3 * p.C; i += 256)
    if (hist[i]) atomicAdd(&counts[(size_t)n * 3 * p.C + i], (int)hist[i]);
}

extern "C" int dl3p_class_counts(const float* z, int ldz, const float* labels, int32_t* counts, int N, int h, int w,
                                 int C, int H, int W, void* stream) {
  DL3P_CHECK_ARG(z && labels && counts && ldz >= C, "dl3p_class_counts: bad arguments");
  DL3P_CHECK_ARG(C > 0 && C <= 1024 && N > 0 && h > 0 && w > 0 && H > 0 && W > 0, "dl3p_class_counts: bad dims");
  EvalParams p = {};
  p.z = z; p.ldz = ldz; p.labels = labels; p.N = N; p.h = h; p.w = w; p.C = C; p.H = H; p.W = W;
  int bpi = (H * W + 256 * 8 - 1) / (256 * 8);
  if (bpi < 1) bpi = 1;
  if (bpi > 256) bpi = 256;
  hipLaunchKernelGGL(class_counts_kernel, dim3((unsigned)(N * bpi)), dim3(256), (size_t)3 * C * sizeof(unsigned int),
                     (hipStream_t)stream, p, counts, bpi);
  DL3P_CHECK_LAUNCH("dl3p_class_counts");
  return DL3P_OK;
}

// ------------------------------------------------------------------------------ fused training head
// pred_resize + Softmax + loss + the transposed resize of the gradient in ONE kernel: the (N,H,W,C) gradient
// (404 MB at batch 16, written by head_kernel and read back by resize_bwd_kernel) never exists.  A workgroup
// owns a tile of TH x TW logit pixels: it stages their neighbourhood of z in LDS, evaluates softmax / loss /
// gradient for every full-resolution pixel that reads the tile (halo pixels are evaluated by both neighbours:
// 2x the minimum work, all of it out of LDS), keeps that gradient tile in LDS and gathers it back onto the
// logit pixels with the bilinear weights, in the same (oy, ox) order as resize_bwd_kernel -> same bits.
struct HeadTrainParams {
  const float* z; int ldz; const float* labels; int ignore_index; float inv_count;
  float* gz; int ldgz; int accumulate; float* loss_partials;
  int N, h, w, C, H, W, tiles_y, tiles_x;
  long long tiles;
};

template <int CP, int TH, int TW, int EH, int EW>
__global__ __launch_bounds__(512) void head_train_kernel(HeadTrainParams p) {
  constexpr int NTHR = 512;
  constexpr int ZH = TH + 4, ZW = TW + 4;                // z tile with a 2-pixel halo
  constexpr int C4 = CP / 4;
  extern __shared__ __attribute__((aligned(16))) float ht_lds[];
  float* zt = ht_lds;                                    // [ZH*ZW][CP]
  float* dt = ht_lds + ZH * ZW * CP;                     // [EH*EW][CP]
  float* wy_s = dt + EH * EW * CP;                       // [TH][EH] row weights of the transposed resize
  float* wx_s = wy_s + TH * EH;                          // [TW][EW]
  int* rng_s = reinterpret_cast<int*>(wx_s + TW * EW);   // [TH][2] + [TW][2]: first / last non-zero weight
  __shared__ float wsum[8];
  const int t = threadIdx.x;
  const float sy = (float)p.h / (float)p.H, sx = (float)p.w / (float)p.W;
  const float isy = (float)p.H / (float)p.h, isx = (float)p.W / (float)p.w;
  float loss = 0.f;
  for (long long tile = blockIdx.x; tile < p.tiles; tile += gridDim.x) {
    const int tx = (int)(tile % p.tiles_x);
    const int trow = (int)(tile / p.tiles_x);
    const int ty = trow % p.tiles_y;
    const int n = trow / p.tiles_y;
    const int iy0 = ty * TH, ix0 = tx * TW;
    const int iy1 = min(iy0 + TH, p.h) - 1, ix1 = min(ix0 + TW, p.w) - 1;
    int Y0, Y1, X0, X1, tmp;
    touch_range(iy0, isy, p.H, Y0, tmp);
    touch_range(iy1, isy, p.H, tmp, Y1);
    touch_range(ix0, isx, p.W, X0, tmp);
    touch_range(ix1, isx, p.W, tmp, X1);
    if (iy0 == 0) Y0 = 0;
    if (iy1 == p.h - 1) Y1 = p.H - 1;
    if (ix0 == 0) X0 = 0;
    if (ix1 == p.w - 1) X1 = p.W - 1;
    const int eh = Y1 - Y0 + 1, ew = X1 - X0 + 1;        // <= EH, EW (host check)
    const int zy0 = iy0 - 2, zx0 = ix0 - 2;
    __syncthreads();                                     // previous tile's gather is done with dt / zt
    // 1. z neighbourhood -> LDS (rows / columns outside the map are clamped copies; never read with weight)
    const float* zimg = p.z + (size_t)n * p.h * p.w * p.ldz;
    for (int i = t; i < ZH * ZW * C4; i += NTHR) {
      const int c4 = i % C4, px = i / C4;
      const int zy = min(max(zy0 + px / ZW, 0), p.h - 1), zx = min(max(zx0 + px % ZW, 0), p.w - 1);
      *reinterpret_cast<float4*>(&zt[px * CP + c4 * 4]) = ld4(zimg + ((size_t)zy * p.w + zx) * p.ldz + c4 * 4);
    }
    // weights of the transposed resize for this tile (one table entry per thread)
    for (int i = t; i < TH * EH + TW * EW; i += NTHR) {
      if (i < TH * EH) {
        const int r = i / EH, oy = Y0 + i - r * EH;
        float wgt = 0.f;
        if (oy <= Y1 && iy0 + r <= iy1) {
          const Lerp ly = lerp_coeff(oy, sy, p.h);
          wgt = (ly.lo == iy0 + r ? 1.f - ly.t : 0.f) + (ly.hi == iy0 + r ? ly.t : 0.f);
        }
        wy_s[i] = wgt;
      } else {
        const int j = i - TH * EH;
        const int r = j / EW, ox = X0 + j - r * EW;
        float wgt = 0.f;
        if (ox <= X1 && ix0 + r <= ix1) {
          const Lerp lx = lerp_coeff(ox, sx, p.w);
          wgt = (lx.lo == ix0 + r ? 1.f - lx.t : 0.f) + (lx.hi == ix0 + r ? lx.t : 0.f);
        }
        wx_s[j] = wgt;
      }
    }
    __syncthreads();
    if (t < TH + TW) {
      const float* tab = t < TH ? wy_s + t * EH : wx_s + (t - TH) * EW;
      const int len = t < TH ? EH : EW;
      int first = len, last = -1;
      for (int i = 0; i < len; ++i)
        if (tab[i] != 0.f) { if (first == len) first = i; last = i; }
      rng_s[2 * t] = first;
      rng_s[2 * t + 1] = last;
    }
    // 2. every full-resolution pixel of the extent: logits, softmax, loss (owner tile only), gradient -> LDS
    for (int e = t; e < eh * ew; e += NTHR) {
      const int ey = e / ew, ex = e - ey * ew;
      const int oy = Y0 + ey, ox = X0 + ex;
      const Lerp ly = lerp_coeff(oy, sy, p.h), lx = lerp_coeff(ox, sx, p.w);
      const float* ptl = zt + ((ly.lo - zy0) * ZW + (lx.lo - zx0)) * CP;
      const float* ptr = zt + ((ly.lo - zy0) * ZW + (lx.hi - zx0)) * CP;
      const float* pbl = zt + ((ly.hi - zy0) * ZW + (lx.lo - zx0)) * CP;
      const float* pbr = zt + ((ly.hi - zy0) * ZW + (lx.hi - zx0)) * CP;
      float v[CP];
#pragma unroll
      for (int c4 = 0; c4 < C4; ++c4) {
        const float4 tl = *reinterpret_cast<const float4*>(ptl + c4 * 4), tr = *reinterpret_cast<const float4*>(ptr + c4 * 4);
        const float4 bl = *reinterpret_cast<const float4*>(pbl + c4 * 4), br = *reinterpret_cast<const float4*>(pbr + c4 * 4);
#define LERP2(f, i) { float top = tl.f + (tr.f - tl.f) * lx.t; float bot = bl.f + (br.f - bl.f) * lx.t; v[c4 * 4 + i] = top + (bot - top) * ly.t; }
        LERP2(x, 0) LERP2(y, 1) LERP2(z, 2) LERP2(w, 3)
#undef LERP2
      }
      float mx = -3.0e38f;
#pragma unroll
      for (int c = 0; c < CP; ++c) if (c < p.C) mx = fmaxf(mx, v[c]);
      float sum = 0.f;
#pragma unroll
      for (int c = 0; c < CP; ++c) {
        v[c] = c < p.C ? __expf(v[c] - mx) : 0.f;
        sum += v[c];
      }
      const float inv = 1.f / sum;
      const int lab = (int)p.labels[((size_t)n * p.H + oy) * p.W + ox];
      const bool masked = p.ignore_index != 0 && lab == p.ignore_index;
      const bool valid = !masked && lab >= 0 && lab < p.C;
      float pt = 0.f;
#pragma unroll
      for (int c = 0; c < CP; ++c) if (c == lab) pt = v[c] * inv;
      const bool unclipped = pt > 1e-7f && pt < 1.f - 1e-7f;
      const bool owner = ly.lo >= iy0 && ly.lo <= iy1 && lx.lo >= ix0 && lx.lo <= ix1;
      if (valid && owner) loss += -logf(fminf(fmaxf(pt, 1e-7f), 1.f - 1e-7f));
      const float gs = (valid && unclipped) ? p.inv_count : 0.f;
#pragma unroll
      for (int c4 = 0; c4 < C4; ++c4) {
        float d[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int c = c4 * 4 + i;
          d[i] = c < p.C ? gs * (v[c] * inv - (c == lab ? 1.f : 0.f)) : 0.f;
        }
        *reinterpret_cast<float4*>(&dt[e * CP + c4 * 4]) = make_float4(d[0], d[1], d[2], d[3]);
      }
    }
    __syncthreads();
    // 3. transposed resize: logit pixel (iy, ix) gathers the gradient of every pixel that read it
    if (t < TH * TW * C4) {
      const int c4 = t % C4, lp = t / C4;
      const int iy = iy0 + lp / TW, ix = ix0 + lp % TW;
      if (iy <= iy1 && ix <= ix1) {
        const int ry = lp / TW, rx = lp % TW;
        const int ey0 = rng_s[2 * ry], ey1 = rng_s[2 * ry + 1];
        const int ex0 = rng_s[2 * (TH + rx)], ex1 = rng_s[2 * (TH + rx) + 1];
        float4 acc = zero4();
        for (int ey = ey0; ey <= ey1; ++ey) {
          const float wy = wy_s[ry * EH + ey];
          if (wy == 0.f) continue;
          for (int ex = ex0; ex <= ex1; ++ex) {
            const float wx = wx_s[rx * EW + ex];
            if (wx == 0.f) continue;
            const float wgt = wy * wx;
            const float4 g = *reinterpret_cast<const float4*>(&dt[(ey * ew + ex) * CP + c4 * 4]);
            acc.x = fmaf(g.x, wgt, acc.x); acc.y = fmaf(g.y, wgt, acc.y);
            acc.z = fmaf(g.z, wgt, acc.z); acc.w = fmaf(g.w, wgt, acc.w);
          }
        }
        float* o = p.gz + (((size_t)n * p.h + iy) * p.w + ix) * p.ldgz + c4 * 4;
        if (p.accumulate) acc = add4(acc, ld4(o));
        st4(o, acc);
      }
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) loss += __shfl_xor(loss, off);
  if ((t & 63) == 0) wsum[t >> 6] = loss;
  __syncthreads();
  if (t == 0) p.loss_partials[blockIdx.x] = (((wsum[0] + wsum[1]) + (wsum[2] + wsum[3])) + ((wsum[4] + wsum[5]) + (wsum[6] + wsum[7]))) * p.inv_count;
}

// the fused head handles upsampling factors up to ~4.2 (logits at OS 4); others use the two-kernel path
#define HT_TH 4
#define HT_TW 8
#define HT_EH 26
#define HT_EW 42
static bool head_train_fits(int h, int w, int H, int W) {
  if (h < 1 || w < 1 || H < h || W < w) return false;
  const float isy = (float)H / (float)h, isx = (float)W / (float)w;
  // extent of a tile: touch_range(first).o0 .. touch_range(last).o1
  const float eh = ((float)HT_TH + 1.f) * isy + 5.f, ew = ((float)HT_TW + 1.f) * isx + 5.f;
  return eh <= (float)HT_EH && ew <= (float)HT_EW;
}

extern "C" int dl3p_head_train_supported(int h, int w, int C, int H, int W) {
  const int cp = ((C + 3) / 4) * 4;      // instantiated for 20 / 24 / 32 padded classes (the (N,h,w,cp) logits' row)
  return ((cp == 20 || cp == 24 || cp == 32) && head_train_fits(h, w, H, W)) ? 1 : 0;
}

template <int CP>
static void launch_head_train(const HeadTrainParams& p, unsigned grid, hipStream_t st) {
  constexpr size_t lds = sizeof(float) * ((size_t)CP * ((HT_TH + 4) * (HT_TW + 4) + HT_EH * HT_EW) + HT_TH * HT_EH + HT_TW * HT_EW +
                                          2 * (HT_TH + HT_TW));
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)head_train_kernel<CP, HT_TH, HT_TW, HT_EH, HT_EW>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL((head_train_kernel<CP, HT_TH, HT_TW, HT_EH, HT_EW>), dim3(grid), dim3(512), lds, st, p);
}

extern "C" int dl3p_head_train(const float* z, int ldz, const float* labels, int ignore_index, float inv_count,
                               float* gz, int ldgz, int accumulate, float* loss_partials, int* rows_out,
                               int N, int h, int w, int C, int H, int W, void* stream) {
  DL3P_CHECK_ARG(z && labels && gz && loss_partials && aligned16(z) && aligned16(gz) && ldz % 4 == 0 && ldgz % 4 == 0,
                 "dl3p_head_train: null / misaligned pointer");
  DL3P_CHECK_ARG(dl3p_head_train_supported(h, w, C, H, W), "dl3p_head_train: upsampling %dx%d -> %dx%d not supported "
                 "(use dl3p_upsample_softmax_ce + dl3p_resize_bilinear_bwd)", h, w, H, W);
  const int cp = ((C + 3) / 4) * 4;
  const int cpv = cp <= 20 ? 20 : (cp <= 24 ? 24 : 32);
  DL3P_CHECK_ARG(ldz >= cpv && ldgz >= cpv, "dl3p_head_train: ld=%d/%d must be >= %d for C=%d (use dl3p_upsample_softmax_ce)",
                 ldz, ldgz, cpv, C);
  HeadTrainParams p = {};
  p.z = z; p.ldz = ldz; p.labels = labels; p.ignore_index = ignore_index; p.inv_count = inv_count;
  p.gz = gz; p.ldgz = ldgz; p.accumulate = accumulate; p.loss_partials = loss_partials;
  p.N = N; p.h = h; p.w = w; p.C = C; p.H = H; p.W = W;
  p.tiles_y = ceil_div(h, HT_TH); p.tiles_x = ceil_div(w, HT_TW);
  p.tiles = (long long)N * p.tiles_y * p.tiles_x;
  long long blocks = p.tiles < DL3P_MAX_STAT_ROWS ? p.tiles : DL3P_MAX_STAT_ROWS;
  if (rows_out) *rows_out = (int)blocks;
  hipStream_t st = (hipStream_t)stream;
  if (cpv == 20) launch_head_train<20>(p, (unsigned)blocks, st);
  else if (cpv == 24) launch_head_train<24>(p, (unsigned)blocks, st);
  else launch_head_train<32>(p, (unsigned)blocks, st);
  DL3P_CHECK_LAUNCH("dl3p_head_train");
  return DL3P_OK;
}

// ------------------------------------------------------------------------------ fused training head, separable form (round 5)
// pred_resize + Softmax + cross-entropy + the transposed pred_resize of the gradient with a QUARTER of the (N,H,W,C) gradient's HBM
// traffic (404 MB written and read back at batch 16 / 513x513): the transposed bilinear resize factorises, G_low = Ry^T (G_high Rx),
// and its x half needs one full-resolution row at a time.
//   x pass: one workgroup of 512 threads per CU walks a contiguous chunk of full-resolution rows.  LDS holds the two logit rows a
//     full-resolution row interpolates between ALREADY INTERPOLATED IN x (a ring of two rows [CP/4][W] float4; a logit row is
//     interpolated once per workgroup, when the walk first needs it -- every ~H/h rows) and the gradient of the current row.  Per row:
//     one pixel per thread -> top + (bot - top) ty (the expression of head_kernel: same bits), softmax, loss, gradient -> LDS; then
//     every (logit column, 4 channels) item gathers its <= 9 pixels of that row with weights it computed once per workgroup and stores
//     the row of G_high Rx: (N,H,w,CP), 1 / (W/w) of the gradient.  The kernel is bound by LDS bandwidth (128 B / clk / CU: a 16-byte
//     read per lane is 8 clocks a wave), which is why the x interpolation is shared between the rows and every LDS array is laid out
//     in channel-quad planes (lanes = consecutive pixels -> consecutive 16-byte words, no bank conflicts).
//   y pass: one thread per (logit pixel, 4 channels) adds up its <= ~2 H/h rows of that tensor -- coalesced, 16 bytes per lane.
// Against dl3p_head_train (the tile kernel): no 100 KB gradient tile with 2.1x the pixels evaluated.
// Summation order differs from dl3p_resize_bilinear_bwd (x first, then y): equal to rounding, not bit for bit.
struct HeadRowsParams {
  const float* z; int ldz; const float* labels; int ignore_index; float inv_count;
  float* gxh; float* loss_partials;
  int N, h, w, C, H, W, chunk;
  int nseg, jw, plane;       // column segments of a row (logit columns per segment), LDS plane stride for the widest segment
};

typedef float hr_f2 __attribute__((ext_vector_type(2)));
#define HR_THREADS 1024

// plane stride (floats) of the LDS rows [CP/4][W] float4: = 4 (mod 64), so that the channel quads of a logit column and the columns
// next to it (16 floats apart at W / w = 4) fall on different LDS banks in the gather
__host__ __device__ static inline int head_rows_plane(int W) { return 4 * W + ((4 - (4 * W) % 64) + 64) % 64; }

// first full-resolution column with a non-zero weight for logit column j and the number of columns up to the last one
__host__ __device__ static inline void head_rows_window(int j, int w, int W, int& first, int& count) {
  const float sx = (float)w / (float)W, isx = (float)W / (float)w;
  int x0, x1;
  touch_range(j, isx, W, x0, x1);
  if (j == 0) x0 = 0;
  if (j == w - 1) x1 = W - 1;
  first = -1;
  int last = -1;
  for (int X = x0; X <= x1; ++X) {
    const Lerp lx = lerp_coeff(X, sx, w);
    const float wx = (lx.lo == j ? 1.f - lx.t : 0.f) + (lx.hi == j ? lx.t : 0.f);
    if (wx != 0.f) {
      if (first < 0) first = X;
      last = X;
    }
  }
  if (first < 0) { first = x0; last = x0 - 1; }
  count = last - first + 1;
}

template <int CP, int NI, int MAXW>
__global__ __launch_bounds__(HR_THREADS) void head_xpass_kernel(HeadRowsParams p) {
  constexpr int C4 = CP / 4;
  extern __shared__ __attribute__((aligned(16))) float hr_lds[];
  // A row wider than LDS holds (W > ~530 at 24 channels: 769 x 769, 1024 x 2048) is cut into COLUMN SEGMENTS of p.jw logit columns: a
  // workgroup evaluates the pixels [Xs, Xe] that carry weight for its columns [j0, j1) -- neighbours overlap by the ~4 pixels of a
  // window, evaluated by both -- and counts the loss of the pixels whose left logit column is its own.  LDS indices are segment-local.
  const int seg = blockIdx.x % p.nseg, chunk_id = blockIdx.x / p.nseg;
  const int j0 = seg * p.jw, j1 = min(j0 + p.jw, p.w);
  int Xs, Xe;
  {
    int f, c;
    head_rows_window(j0, p.w, p.W, Xs, c);
    head_rows_window(j1 - 1, p.w, p.W, f, c);
    Xe = f + c - 1;
  }
  const int Wn = Xe - Xs + 1;
  const int PW = p.plane;
  float* xr = hr_lds;                                   // [2][CP/4][plane]: logit rows i (slot i & 1), interpolated in x
  float* gr = hr_lds + 2 * C4 * PW;                     // [CP/4][plane]: gradient of the current full-resolution row
  __shared__ float wsum[HR_THREADS / 64];
  const int t = threadIdx.x;
  const float sy = (float)p.h / (float)p.H, sx = (float)p.w / (float)p.W;
  const int rows_total = p.N * p.H;
  const int r0 = chunk_id * p.chunk, r1 = min(r0 + p.chunk, rows_total);
  // this thread's items of the transposed resize in x: (logit column j, channel quad c4); window start and weights, once
  int x0s[NI];
  float wxs[NI][MAXW];
#pragma unroll
  for (int m = 0; m < NI; ++m) {
    const int it = t + HR_THREADS * m;
    const bool on = it < (j1 - j0) * C4;
    const int j = on ? j0 + it / C4 : j0;
    int first, count;
    head_rows_window(j, p.w, p.W, first, count);
    x0s[m] = first - Xs;
#pragma unroll
    for (int k = 0; k < MAXW; ++k) {
      const Lerp lx = lerp_coeff(min(first + k, p.W - 1), sx, p.w);
      const float wx = (lx.lo == j ? 1.f - lx.t : 0.f) + (lx.hi == j ? lx.t : 0.f);
      wxs[m][k] = (on && k < count) ? wx : 0.f;
    }
  }
  float loss = 0.f;
  int have_n = -1, have0 = -1, have1 = -1;              // the logit row in slot 0 / 1 (of image have_n)
  // gfx9 counts loads and stores in one counter (vmcnt), so a wait for any load also waits for every store in flight.  The walk keeps
  // both out of the way: the gather result of a row stays in registers and is STORED AT THE START OF THE NEXT ROW'S PIXEL PHASE,
  // together with the request for the next row's label; both are waited for once, at the end of that phase -- nothing is in flight
  // across the barriers, the ring refresh or the loop edge, where the compiler's waits would expose the latency.
  // One or two pixels past a multiple of 64 (W = 513, 769, 1025: the crop sizes of the reference are 2^k + 1) would cost a whole
  // wave-iteration of the pixel phase for one lane -- on the SIMD that already holds the most pixel waves.  Those tail pixels are
  // evaluated by ONE wave with a channel per lane instead (32 lanes per pixel, reductions by lane shuffles: ~50 instructions, not 230),
  // a wave that sits on another SIMD.
  const int lane = t & 63, wave = t >> 6;
  const int Wfull = (Wn & 63) <= 2 ? (Wn & ~63) : Wn;
  const int tail_wave = min((Wfull >> 6) + 1, HR_THREADS / 64 - 1);
  const bool tail_lane = wave == tail_wave && Wfull + (lane >> 5) < Wn;
  const int tc = tail_lane ? Wfull + (lane >> 5) : min(t, Wn - 1);          // (segment-local)
  auto owns = [&](int Xl) { const int lo = lerp_coeff(Xs + Xl, sx, p.w).lo; return lo >= j0 && lo < j1; };
  const bool own_first = owns(tc);
  int lab_cur = r0 < r1 ? (int)p.labels[(size_t)r0 * p.W + Xs + tc] : 0;
  float4 held[NI];
#pragma unroll
  for (int m = 0; m < NI; ++m) held[m] = zero4();
  for (int row = r0; row < r1; ++row) {
    const int n = row / p.H, Y = row - n * p.H;
    const Lerp ly = lerp_coeff(Y, sy, p.h);
    // the ring: logit rows this full-resolution row needs and does not have.  (Nobody reads xr any more: the previous row's pixel
    // phase ended at its barrier.)  Every load of a refresh is issued before the first is used.
    if (n != have_n) { have_n = n; have0 = have1 = -1; }
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int i = e ? ly.hi : ly.lo;
      const int have = (i & 1) ? have1 : have0;
      if (have == i) continue;
      const float* zrow = p.z + ((size_t)n * p.h + i) * p.w * p.ldz;
      float* dst = xr + (i & 1) * C4 * PW;
      for (int X = t; X < Wn; X += HR_THREADS) {         // a pixel per thread: its two corners, all channel quads, loads first
        const Lerp lx = lerp_coeff(Xs + X, sx, p.w);
        const float* pa = zrow + (size_t)lx.lo * p.ldz;
        const float* pb = zrow + (size_t)lx.hi * p.ldz;
        float4 a[C4], b[C4];
#pragma unroll
        for (int c4 = 0; c4 < C4; ++c4) { a[c4] = ld4(pa + c4 * 4); b[c4] = ld4(pb + c4 * 4); }
        const hr_f2 tx2 = {lx.t, lx.t};
#pragma unroll
        for (int c4 = 0; c4 < C4; ++c4) {
          const hr_f2 a0 = {a[c4].x, a[c4].y}, a1 = {a[c4].z, a[c4].w}, b0 = {b[c4].x, b[c4].y}, b1 = {b[c4].z, b[c4].w};
          const hr_f2 o0 = a0 + (b0 - a0) * tx2, o1 = a1 + (b1 - a1) * tx2;
          *reinterpret_cast<float4*>(dst + c4 * PW + X * 4) = make_float4(o0[0], o0[1], o1[0], o1[1]);
        }
      }
      if (i & 1) have1 = i; else have0 = i;
    }
    __syncthreads();                                     // xr is in place; the previous row's gather is done with gr
    const float* top = xr + (ly.lo & 1) * C4 * PW;
    const float* bot = xr + (ly.hi & 1) * C4 * PW;
    auto pixel = [&](int X, int lab, bool own) {
      // logits: the expression of head_kernel (top + (bot - top) * ty, unfused), two channels per packed instruction.  From there on
      // the arithmetic is arranged for instruction count (the phase is bound by VALU issue): exp2 of a fused (v - max) * log2(e),
      // pairwise sums, one multiplier gs / sum for the gradient -- equal to head_kernel's softmax to rounding, not bit for bit.
      hr_f2 v[CP / 2];
      const hr_f2 ty2 = {ly.t, ly.t};
#pragma unroll
      for (int c4 = 0; c4 < C4; ++c4) {
        const float4 a = *reinterpret_cast<const float4*>(top + c4 * PW + X * 4);
        const float4 b = *reinterpret_cast<const float4*>(bot + c4 * PW + X * 4);
        const hr_f2 a0 = {a.x, a.y}, a1 = {a.z, a.w}, b0 = {b.x, b.y}, b1 = {b.z, b.w};
        v[c4 * 2] = a0 + (b0 - a0) * ty2;
        v[c4 * 2 + 1] = a1 + (b1 - a1) * ty2;
      }
      // padded channels (Cpad - C <= 3, all in the last quad) never win the maximum and add exp2(-huge) = 0 to the sum
      if (CP - 3 >= p.C) v[CP / 2 - 2][1] = -3.0e38f;
      if (CP - 2 >= p.C) v[CP / 2 - 1][0] = -3.0e38f;
      if (CP - 1 >= p.C) v[CP / 2 - 1][1] = -3.0e38f;
      float mx = fmaxf(v[0][0], v[0][1]);
#pragma unroll
      for (int c2 = 1; c2 < CP / 2; ++c2) mx = fmaxf(fmaxf(mx, v[c2][0]), v[c2][1]);
      constexpr float LOG2E = 1.4426950408889634f;
      const float mneg = -mx * LOG2E;
      const hr_f2 l2 = {LOG2E, LOG2E}, m2 = {mneg, mneg};
      // (v_exp_f32 of a large negative argument is 0: the padded channels, and anything 126 octaves under the maximum)
#pragma unroll
      for (int c2 = 0; c2 < CP / 2; ++c2) v[c2] = __builtin_elementwise_fma(v[c2], l2, m2);
#pragma unroll
      for (int c2 = 0; c2 < CP / 2; ++c2) {
        v[c2][0] = __builtin_amdgcn_exp2f(v[c2][0]);
        v[c2][1] = __builtin_amdgcn_exp2f(v[c2][1]);
      }
      hr_f2 s2 = v[0], s3 = v[1];
#pragma unroll
      for (int c2 = 2; c2 < CP / 2; c2 += 2) { s2 += v[c2]; s3 += v[c2 + 1]; }
      s2 += s3;
      const float inv = 1.f / (s2[0] + s2[1]);
      const bool masked = p.ignore_index != 0 && lab == p.ignore_index;
      const bool valid = !masked && lab >= 0 && lab < p.C;
      // the label's probability: its logit again from LDS (the same expressions -> the same bits) instead of a 24-way select
      float pt = 0.f;
      if (valid) {
        const int o = (lab >> 2) * PW + X * 4 + (lab & 3);
        const float a = top[o], b = bot[o];
        pt = __builtin_amdgcn_exp2f(fmaf(a + (b - a) * ly.t, LOG2E, mneg)) * inv;
      }
      const bool unclipped = pt > 1e-7f && pt < 1.f - 1e-7f;
      if (valid && own) loss += -logf(fminf(fmaxf(pt, 1e-7f), 1.f - 1e-7f));
      const float gs = (valid && unclipped) ? p.inv_count : 0.f;
      const float k = gs * inv;
      const hr_f2 k2 = {k, k};
      float* gpx = gr + X * 4;
#pragma unroll
      for (int c4 = 0; c4 < C4; ++c4) {
        // gs * p; the label's channel is patched below (gs * (p - 1))
        const hr_f2 d0 = v[c4 * 2] * k2, d1 = v[c4 * 2 + 1] * k2;
        *reinterpret_cast<float4*>(gpx + c4 * PW) = make_float4(d0[0], d0[1], d1[0], d1[1]);
      }
      if (valid) gpx[(lab >> 2) * PW + (lab & 3)] = gs * (pt - 1.f);
    };
    if (row > r0) {
      float* orow = p.gxh + (size_t)(row - 1) * p.w * CP;
#pragma unroll
      for (int m = 0; m < NI; ++m)
        if (t + HR_THREADS * m < (j1 - j0) * C4) st4(orow + ((size_t)j0 * C4 + t + HR_THREADS * m) * 4, held[m]);
    }
    const float lab_req = p.labels[(size_t)min(row + 1, r1 - 1) * p.W + Xs + tc];
    if (t < Wfull) pixel(t, lab_cur, own_first);
    for (int X = t + HR_THREADS; X < Wfull; X += HR_THREADS) pixel(X, (int)p.labels[(size_t)row * p.W + Xs + X], owns(X));
    if (wave == tail_wave && Wfull < Wn) {
      const int c = lane & 31, X = min(Wfull + (lane >> 5), Wn - 1), lab = lab_cur;
      const bool on = tail_lane && c < CP;
      const int o = on ? (c >> 2) * PW + X * 4 + (c & 3) : 0;
      const float a = top[o], b = bot[o];
      const float v = (on && c < p.C) ? a + (b - a) * ly.t : -3.0e38f;
      float mx = v;
#pragma unroll
      for (int off = 16; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 32));
      constexpr float LOG2E = 1.4426950408889634f;
      const float e = __builtin_amdgcn_exp2f(fmaf(v, LOG2E, -mx * LOG2E));
      float sum = e;
#pragma unroll
      for (int off = 16; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 32);
      const float inv = 1.f / sum;
      const bool masked = p.ignore_index != 0 && lab == p.ignore_index;
      const bool valid = tail_lane && !masked && lab >= 0 && lab < p.C;
      const float pt = valid ? __shfl(e, (lane & 32) + (lab & 31)) * inv : 0.f;
      const bool unclipped = pt > 1e-7f && pt < 1.f - 1e-7f;
      if (valid && c == 0 && own_first) loss += -logf(fminf(fmaxf(pt, 1e-7f), 1.f - 1e-7f));
      const float gs = (valid && unclipped) ? p.inv_count : 0.f;
      if (on) gr[o] = c == lab ? gs * (pt - 1.f) : e * (gs * inv);
    }
    lab_cur = (int)lab_req;
    asm volatile("" : "+v"(lab_cur));                    // the wait for the label (and the stores above) is HERE
    __syncthreads();
#pragma unroll
    for (int m = 0; m < NI; ++m) {
      const int it = t + HR_THREADS * m;
      if (it >= (j1 - j0) * C4) continue;
      float4 acc = zero4();
      const float* gp = gr + (it % C4) * PW;
#pragma unroll
      for (int k = 0; k < MAXW; ++k) {
        const float4 g = *reinterpret_cast<const float4*>(gp + min(x0s[m] + k, Wn - 1) * 4);
        acc.x = fmaf(g.x, wxs[m][k], acc.x); acc.y = fmaf(g.y, wxs[m][k], acc.y);
        acc.z = fmaf(g.z, wxs[m][k], acc.z); acc.w = fmaf(g.w, wxs[m][k], acc.w);
      }
      held[m] = acc;
    }
  }
  if (r1 > r0) {
    float* orow = p.gxh + (size_t)(r1 - 1) * p.w * CP;
#pragma unroll
    for (int m = 0; m < NI; ++m)
      if (t + HR_THREADS * m < (j1 - j0) * C4) st4(orow + ((size_t)j0 * C4 + t + HR_THREADS * m) * 4, held[m]);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) loss += __shfl_xor(loss, off);
  if ((t & 63) == 0) wsum[t >> 6] = loss;
  __syncthreads();
  if (t == 0) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < HR_THREADS / 64; ++i) s += wsum[i];
    p.loss_partials[blockIdx.x] = s * p.inv_count;
  }
}

struct HeadYParams { const float* gxh; float* gz; int ldgz; int accumulate; int N, h, w, H, c4s, cp; long long total; };

template <int MAXW>
__global__ __launch_bounds__(256) void head_ypass_kernel(HeadYParams p) {
  const long long s = (long long)blockIdx.x * 256 + threadIdx.x;
  if (s >= p.total) return;
  const int c4 = (int)(s % p.c4s);
  const long long px = s / p.c4s;
  const int j = (int)(px % p.w);
  const int row = (int)(px / p.w);
  const int i = row % p.h, n = row / p.h;
  const float sy = (float)p.h / (float)p.H;
  // the rows of the workspace this logit row owns (<= MAXW, host-checked): weights first, then every load before the first add
  // (clamped rows with weight 0: fma(g, 0, acc) == acc -- never a load inside a branch)
  int first, count;
  head_rows_window(i, p.h, p.H, first, count);
  float wy[MAXW];
#pragma unroll
  for (int k = 0; k < MAXW; ++k) {
    const Lerp ly = lerp_coeff(min(first + k, p.H - 1), sy, p.h);
    const float wgt = (ly.lo == i ? 1.f - ly.t : 0.f) + (ly.hi == i ? ly.t : 0.f);
    wy[k] = k < count ? wgt : 0.f;
  }
  const float* src = p.gxh + (((size_t)n * p.H) * p.w + j) * p.cp + c4 * 4;
  float4 g[MAXW];
#pragma unroll
  for (int k = 0; k < MAXW; ++k) g[k] = ld4(src + (size_t)min(first + k, first + max(count, 1) - 1) * p.w * p.cp);
  float4 acc = zero4();
#pragma unroll
  for (int k = 0; k < MAXW; ++k) {
    acc.x = fmaf(g[k].x, wy[k], acc.x); acc.y = fmaf(g[k].y, wy[k], acc.y);
    acc.z = fmaf(g[k].z, wy[k], acc.z); acc.w = fmaf(g[k].w, wy[k], acc.w);
  }
  float* o = p.gz + (size_t)px * p.ldgz + c4 * 4;
  if (p.accumulate) acc = add4(acc, ld4(o));
  st4(o, acc);
}

static int head_rows_maxwin(int w, int W) {
  int m = 0;
  for (int j = 0; j < w; ++j) {
    int first, count;
    head_rows_window(j, w, W, first, count);
    m = std::max(m, count);
  }
  return m;
}
// the fewest column segments whose items fit a workgroup (one (column, channel quad) item per thread) and whose three LDS rows fit 150 KB
struct HeadSeg { int nseg, jw, wmax; };
static bool head_rows_segments(int w, int W, int cp, HeadSeg* out) {
  const int c4 = cp / 4;
  for (int nseg = 1; nseg <= 32 && nseg <= w; ++nseg) {
    const int jw = ceil_div(w, nseg);
    if (jw * c4 > HR_THREADS || ceil_div(w, jw) != nseg) continue;
    int wmax = 0;
    for (int s = 0; s < nseg; ++s) {
      const int j0 = s * jw, j1 = std::min(j0 + jw, w);
      int xs, c, f;
      head_rows_window(j0, w, W, xs, c);
      head_rows_window(j1 - 1, w, W, f, c);
      wmax = std::max(wmax, f + c - xs);
    }
    if (sizeof(float) * 3 * (size_t)c4 * head_rows_plane(wmax) > 150 * 1024) continue;
    out->nseg = nseg; out->jw = jw; out->wmax = wmax;
    return true;
  }
  return false;
}

extern "C" int dl3p_head_train_rows_supported(int h, int w, int C, int H, int W) {
  const int cp = ((C + 3) / 4) * 4;
  if (!(cp == 20 || cp == 24 || cp == 32) || h < 1 || w < 1 || H < h || W < w) return 0;
  if (head_rows_maxwin(w, W) > 12 || head_rows_maxwin(h, H) > 12) return 0;     // the window of a logit column / row fits the weight table
  HeadSeg sg;
  return head_rows_segments(w, W, cp, &sg) ? 1 : 0;
}

extern "C" size_t dl3p_head_train_rows_workspace(int N, int h, int w, int C, int H, int W) {
  (void)h; (void)W;
  const int cp = ((C + 3) / 4) * 4;
  return sizeof(float) * (size_t)N * H * w * cp;
}

template <int CP, int NI, int MAXW>
static void launch_head_xpass(const HeadRowsParams& p, unsigned grid, size_t lds, hipStream_t st) {
  (void)hipFuncSetAttribute((const void*)head_xpass_kernel<CP, NI, MAXW>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
  hipLaunchKernelGGL((head_xpass_kernel<CP, NI, MAXW>), dim3(grid), dim3(HR_THREADS), lds, st, p);
}

extern "C" int dl3p_head_train_rows(const float* z, int ldz, const float* labels, int ignore_index, float inv_count, float* gz,
                                    int ldgz, int accumulate, float* loss_partials, int* rows_out, void* workspace,
                                    size_t workspace_bytes, int N, int h, int w, int C, int H, int W, void* stream) {
  DL3P_CHECK_ARG(z && labels && gz && loss_partials && workspace && aligned16(z) && aligned16(gz) && aligned16(workspace) &&
                 ldz % 4 == 0 && ldgz % 4 == 0, "dl3p_head_train_rows: null / misaligned pointer");
  DL3P_CHECK_ARG(dl3p_head_train_rows_supported(h, w, C, H, W), "dl3p_head_train_rows: %dx%d -> %dx%d, %d classes not supported "
                 "(use dl3p_upsample_softmax_loss + dl3p_resize_bilinear_bwd)", h, w, H, W, C);
  const int cp = ((C + 3) / 4) * 4;
  DL3P_CHECK_ARG(ldz >= cp && ldgz >= cp, "dl3p_head_train_rows: ld=%d/%d must be >= %d", ldz, ldgz, cp);
  DL3P_CHECK_ARG(workspace_bytes >= dl3p_head_train_rows_workspace(N, h, w, C, H, W), "dl3p_head_train_rows: workspace of %zu bytes, "
                 "%zu needed", workspace_bytes, dl3p_head_train_rows_workspace(N, h, w, C, H, W));
  DL3P_CHECK_ARG((long long)N * H < (1ll << 31) / (long long)(w * cp), "dl3p_head_train_rows: batch too large");
  HeadRowsParams p = {};
  p.z = z; p.ldz = ldz; p.labels = labels; p.ignore_index = ignore_index; p.inv_count = inv_count;
  p.gxh = (float*)workspace; p.loss_partials = loss_partials;
  p.N = N; p.h = h; p.w = w; p.C = C; p.H = H; p.W = W;
  // one workgroup per CU, one round; never more workgroups than loss partial rows
  HeadSeg sg;
  (void)head_rows_segments(w, W, cp, &sg);
  p.nseg = sg.nseg; p.jw = sg.jw; p.plane = head_rows_plane(sg.wmax);
  const int rows_total = N * H;
  const int slots = std::max(1, std::min(dl3p_device_cus(), (int)DL3P_MAX_STAT_ROWS) / sg.nseg);
  p.chunk = std::max(1, ceil_div(rows_total, slots));
  const int blocks = ceil_div(rows_total, p.chunk) * sg.nseg;
  if (rows_out) *rows_out = blocks;
  const size_t lds = sizeof(float) * 3 * (size_t)(cp / 4) * p.plane;
  const int mw = head_rows_maxwin(w, W);
  hipStream_t st = (hipStream_t)stream;
#define HR_CASE(CC) \
  if (cp == CC) { \
    if (mw <= 5) launch_head_xpass<CC, 1, 5>(p, (unsigned)blocks, lds, st); \
    else if (mw <= 9) launch_head_xpass<CC, 1, 9>(p, (unsigned)blocks, lds, st); \
    else launch_head_xpass<CC, 1, 12>(p, (unsigned)blocks, lds, st); \
  }
  HR_CASE(20) HR_CASE(24) HR_CASE(32)
#undef HR_CASE
  DL3P_CHECK_LAUNCH("dl3p_head_train_rows (x pass)");
  HeadYParams q = {};
  q.gxh = (const float*)workspace; q.gz = gz; q.ldgz = ldgz; q.accumulate = accumulate;
  q.N = N; q.h = h; q.w = w; q.H = H; q.c4s = cp / 4; q.cp = cp; q.total = (long long)N * h * w * (cp / 4);
  const int mwy = head_rows_maxwin(h, H);
  const dim3 ygrid((unsigned)ceil_div(q.total, 256ll));
  if (mwy <= 5) hipLaunchKernelGGL(head_ypass_kernel<5>, ygrid, dim3(256), 0, st, q);
  else if (mwy <= 9) hipLaunchKernelGGL(head_ypass_kernel<9>, ygrid, dim3(256), 0, st, q);
  else hipLaunchKernelGGL(head_ypass_kernel<12>, ygrid, dim3(256), 0, st, q);
  DL3P_CHECK_LAUNCH("dl3p_head_train_rows (y pass)");
  return DL3P_OK;
}
