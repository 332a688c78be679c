// Depthwise (atrous) convolution kernels for gfx950: forward, backward-data, backward-weight.
//
// Replaces DepthwiseConv2D at /root/reference deeplabv3p/models/layers.py:100 (SepConv_BN; ASPP rates
// 6/12/18 at layers.py:146-153), deeplabv3p_mobilenetv2.py:56, deeplabv3p_mobilenetv3.py:173.
//
// HBM-bound (2.25 flop/B fp32).  Layout: NHWC; a thread owns 4 consecutive channels (16-B vector
// loads, a wave covers >= 128 B contiguous per pixel) and a strip of TW output pixels; channel
// lanes are fastest in the block so every global access is coalesced.  Workgroups are persistent
// (grid-stride) so the fused BatchNorm statistics (sum, sum^2 of the raw output) stay in registers
// and leave as ONE partial row per workgroup -- no atomics, deterministic.  The XCD-aware split
// (common.h xcd_range) keeps all taps of one image in one XCD's L2, so a dilated 3x3 (which touches
// the whole 33x33 map from every tile) reads x from HBM once.
// The BN+activation of the producing layer is applied on load (prologue): zero padding is in
// activation space, so out-of-range taps contribute exactly 0.
#include "bf16.h"
#include <stdlib.h>
#include <type_traits>

struct DwParams {
  const float* x; int ldx;
  const float* scale; const float* shift; int act;
  const float* w;
  float* y; int ldy;
  const float* dy; int lddy;   // bwd_weight only
  float* partials;
  int N, H, W, C, Ho, Wo, stride, rate, pad_t, pad_l;
  int c4s, px, nslab, nbx, spr, th, nbands, ks, ks5;
  int uh;              // rows the bands share: band b covers rows [b*uh/nbands, (b+1)*uh/nbands)
  long long total;
  int flip, accumulate;
  int nt;              // streaming stores for the output (forward role only)
  int fast_rows;       // dw_fwd_seg: interior strips on the counted-wait rows (DL3P_DW_FAST_ROWS, default 1)
  int tw;              // strip width the plan was made for (3x3 window kernels)
  int lat;             // residue-lattice kernels (kind 3): pixels per class and dimension, 2 or 3
  int bf16_io;         // x / w / y / dy are bf16 (the mixed-precision entry at the end of this file; window kernels only)
  // fused BatchNorm-backward statistics (data-gradient role, output = gradient of act(BN(z))): partial rows hold
  // (sum g', sum g' * xhat) as dl3p_bn_bwd_reduce would compute them from the finished gradient
  const float* bb_z; int bb_ldz;
  const float* bb_scale; const float* bb_shift; const float* bb_mean; const float* bb_invstd; int bb_act;
  // weight-gradient role with the BatchNorm-backward apply of the conv's own output folded in (BNA instantiations): dy is
  // the gradient of act(BN(z)); dz = c0 * (dy * act'(z*scale+shift) - c1 - xhat * c2) is formed at the output pixel (each is
  // visited once) and written to fa_dz for the data gradient that follows
  const float* fa_z; int fa_ldz;
  const float* fa_scale; const float* fa_shift; const float* fa_mean; const float* fa_invstd; const float* fa_coef; int fa_act;
  float* fa_dz; int fa_lddz;
  // UP instantiations (window kernels, stride 1, rate 1): channels [0, up_C) of the input are NOT read from x but formed on the fly as
  // the bilinear upsampling of up_x (N, up_h, up_w, up_C) to H x W -- dl3p_resize_bilinear_fwd's arithmetic, expression for
  // expression, so the result is the unfused pair's bit for bit; channels from up_C on come from x as ever (Decoder_block,
  // deeplabv3p/models/layers.py:207-215: resize -> concat with the skip features -> depthwise conv, without the resized tensor in HBM)
  const float* up_x; int up_ld, up_h, up_w, up_C;
};

// dl3p_resize_bilinear_fwd's source coordinates (csrc/resize_head.hip lerp_coeff: TF2 half-pixel centres), restated here
struct DwLerp { int lo, hi; float t; };
__device__ __forceinline__ DwLerp dw_lerp(int o, float scale, int in_size) {
  const float src = ((float)o + 0.5f) * scale - 0.5f;
  const float fl = floorf(src);
  DwLerp r;
  r.lo = max((int)fl, 0);
  r.hi = min((int)ceilf(src), in_size - 1);
  r.t = src - fl;
  return r;
}
// one input vector of an UP launch: the four neighbours of (iy, ix) in the low-resolution map, TF's order of operations
// (top = tl + (tr - tl) tx; bottom = bl + (br - bl) tx; out = top + (bottom - top) ty)
__device__ __forceinline__ float4 dw_up_load(const float* img, int up_w, int up_ld, DwLerp ly, DwLerp lx) {
  const float4 tl = ld4(img + ((size_t)ly.lo * up_w + lx.lo) * up_ld);
  const float4 tr = ld4(img + ((size_t)ly.lo * up_w + lx.hi) * up_ld);
  const float4 bl = ld4(img + ((size_t)ly.hi * up_w + lx.lo) * up_ld);
  const float4 br = ld4(img + ((size_t)ly.hi * up_w + lx.hi) * up_ld);
  float4 o;
#define DW_LERP2(f) { const float top = tl.f + (tr.f - tl.f) * lx.t; const float bot = bl.f + (br.f - bl.f) * lx.t; o.f = top + (bot - top) * ly.t; }
  DW_LERP2(x) DW_LERP2(y) DW_LERP2(z) DW_LERP2(w)
#undef DW_LERP2
  return o;
}

// g' = g * act'(z*scale+shift), xhat = (z - mean) * invstd  ->  s[0] += g', s[1] += g' * xhat
__device__ __forceinline__ void bnb_accumulate(float4 (&s1)[2], float4 g, float4 zv, float4 bsc, float4 bsh, float4 bmu,
                                               float4 bis, int act) {
  const float4 u = fma4(zv, bsc, bsh);
  const float4 d = make_float4(g.x * act_grad(u.x, act), g.y * act_grad(u.y, act), g.z * act_grad(u.z, act),
                               g.w * act_grad(u.w, act));
  const float4 xh = make_float4((zv.x - bmu.x) * bis.x, (zv.y - bmu.y) * bis.y, (zv.z - bmu.z) * bis.z, (zv.w - bmu.w) * bis.w);
  s1[0] = add4(s1[0], d);
  s1[1] = fma4(d, xh, s1[1]);
}

// producer prologue, specialised at compile time: PRO 0 = raw tensor (backward-data), 1 = BatchNorm affine
// only (no activation follows the producer's BN), 2 = affine + activation
template <int PRO>
__device__ __forceinline__ float4 prologue4(float4 v, float4 sc, float4 sh, int act) {
  if (PRO == 0) return v;
  v = fma4(v, sc, sh);
  if (PRO == 2) v = act_apply4(v, act);
  return v;
}

// ------------------------------------------------------------------------------ forward, sliding window
// A thread owns 4 channels x TW output columns and walks a band of `th` output rows, keeping the KS
// input rows of its window in registers (already normalised + activated).  Each new output row loads
// only S new input rows and the loads of the next row are issued before the current row's FMAs.
// Atrous rates are handled on the rate x rate sub-lattices of the image: outputs (py + v*r, px + u*r)
// only touch inputs of the same residue class, on which the dilated conv IS a dense 3x3 -- so the
// window walks (u, v) with pixel stride r and every rate gets the same reuse; taps that fall into the
// zero padding are exec-masked loads that never leave the CU (at rate 18 on a 33x33 map a 2x2 output
// block issues 4 loads instead of 36).  This replaces TF's SpaceToBatchND->conv->BatchToSpaceND
// (two extra tensor passes) by index arithmetic.
template <int KS, int TW, int S, int PRO, bool BNB = false, bool UP = false>
__global__ __launch_bounds__(256) void dw_fwd_seg(DwParams p) {
  constexpr int SEG = (TW - 1) * S + KS;
  const int b = blockIdx.x;
  const int slab = b / p.nbx;
  const int bx = b - slab * p.nbx;
  const int t = threadIdx.x;
  const int pl = t / p.c4s;
  const int cl = t - pl * p.c4s;
  const bool active = pl < p.px;
  const int cbase4 = slab * p.c4s;
  const int c = (cbase4 + cl) * 4;
  float4 s1[2] = {zero4(), zero4()};
  // 5x5: the 25 weight vectors of a thread's channels live in LDS ([tap][channel lane], one ds_read_b128 per tap and
  // row) -- in registers they cost 100 VGPRs on top of the 5-row window and halve the occupancy
  constexpr bool WLDS = KS == 5;
  extern __shared__ __attribute__((aligned(16))) float4 dw_w_lds[];
  if (WLDS) {
    for (int i = t; i < KS * KS * p.c4s; i += 256) {
      const int tap = i / p.c4s, lane = i - tap * p.c4s;
      dw_w_lds[i] = ld4(p.w + (size_t)(p.flip ? KS * KS - 1 - tap : tap) * p.C + (cbase4 + lane) * 4);
    }
    __syncthreads();
  }
  if (active) {
    float4 wreg[WLDS ? 1 : KS * KS];
    if (!WLDS) {
#pragma unroll
      for (int i = 0; i < KS * KS; ++i) wreg[i] = ld4(p.w + (size_t)(p.flip ? KS * KS - 1 - i : i) * p.C + c);
    }
    auto wtap = [&](int i) { return WLDS ? dw_w_lds[i * p.c4s + cl] : wreg[WLDS ? 0 : i]; };
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = zero4();
    if (p.scale) { sc = ld4(p.scale + c); sh = ld4(p.shift + c); }
    float4 bsc = zero4(), bsh = zero4(), bmu = zero4(), bis = zero4();
    if (BNB) { bsc = ld4(p.bb_scale + c); bsh = ld4(p.bb_shift + c); bmu = ld4(p.bb_mean + c); bis = ld4(p.bb_invstd + c); }
    const int act = p.act;
    const int th = p.th, nbands = p.nbands, rate = p.rate;
    XcdRange r = xcd_range(p.total, bx, p.nbx, p.px, pl);
    for (int s = r.begin; s < r.end; s += r.step) {
      const int strip = s % p.spr;
      int t2 = s / p.spr;
      const int band = t2 % nbands;
      t2 /= nbands;
      const int phase = t2 % (rate * rate);      // sub-lattice (py, px); 0 when rate == 1
      const int n = t2 / (rate * rate);
      const int py = phase / rate, pxo = phase - py * rate;
      const int u0 = strip * TW;                 // first sub-lattice column of this strip
      const int v0 = band * p.uh / nbands, v1 = (band + 1) * p.uh / nbands;
      const int ox0 = pxo + u0 * rate;
      const int ix0 = ox0 * S - p.pad_l;
      const float* ximg = p.x + (size_t)n * p.H * p.W * p.ldx + c;
      int coff[SEG];
      bool cok[SEG];
#pragma unroll
      for (int i = 0; i < SEG; ++i) {
        const int ix = ix0 + i * rate;
        cok[i] = ix >= 0 && ix < p.W;
        coff[i] = ix * p.ldx;
      }
      // UP: the column coefficients of the strip's input pixels in the low-resolution map (rows: per input row below)
      DwLerp lxs[UP ? SEG : 1];
      const bool upl = UP && c < p.up_C;
      const float* uimg = nullptr;
      float usy = 0.f;
      if (UP) {
        const float usx = (float)p.up_w / (float)p.W;
        usy = (float)p.up_h / (float)p.H;
#pragma unroll
        for (int i = 0; i < SEG; ++i) lxs[i] = dw_lerp(min(max(ix0 + i * rate, 0), p.W - 1), usx, p.up_w);
        uimg = p.up_x + (size_t)n * p.up_h * p.up_w * p.up_ld + (upl ? c : 0);
      }
      float4 win[KS][SEG];
      float4 raw[S][SEG];
      // prime the window with the KS rows of the first output row (exec-masked loads, math afterwards)
      {
        const int oy = py + v0 * rate;
#pragma unroll
        for (int ky = 0; ky < KS; ++ky) {
          const int iy = oy * S - p.pad_t + ky * rate;
          const bool yok = iy >= 0 && iy < p.H;
          const float* xrow = ximg + (size_t)iy * p.W * p.ldx;
#pragma unroll
          for (int i = 0; i < SEG; ++i) {
            win[ky][i] = zero4();
            if (yok && cok[i]) win[ky][i] = (UP && upl) ? dw_up_load(uimg, p.up_w, p.up_ld, dw_lerp(iy, usy, p.up_h), lxs[UP ? i : 0]) : ld4(xrow + coff[i]);
          }
        }
#pragma unroll
        for (int ky = 0; ky < KS; ++ky) {
          const int iy = oy * S - p.pad_t + ky * rate;
          const bool yok = iy >= 0 && iy < p.H;
#pragma unroll
          for (int i = 0; i < SEG; ++i) {
            const float4 a = prologue4<PRO>(win[ky][i], sc, sh, act);
            win[ky][i] = (yok && cok[i]) ? a : zero4();
          }
        }
      }
      // rows of the band that exist on this sub-lattice
      const int vmax = (p.Ho - py + rate - 1) / rate;
      const int vend = v1 < vmax ? v1 : vmax;
      // FAST rows (interior strips of a plain launch: every output column of the strip exists, nothing is accumulated): the loads
      // of the entering rows are UNCONDITIONAL (row / column clamped into the image; what is outside is zeroed after the prologue,
      // as always) and so are the stores, and the band's last row (which loads nothing) is a separate copy of the body -- the
      // compiler can then count: the wait for the entering rows becomes vmcnt(TW), the row's stores stay in flight.  With any
      // of them under an `if` it waits vmcnt(0) at the top of every row: vmcnt retires in order, so each row also waited for the
      // round trip of the stores of the row before (csrc/resize_head.hip, resize_fwd_seg_kernel, has the measurement).
      int coffc[SEG];
#pragma unroll
      for (int i = 0; i < SEG; ++i) coffc[i] = min(max(ix0 + i * rate, 0), p.W - 1) * p.ldx;
      auto row = [&](int v, auto fast_c, auto load_c) __attribute__((always_inline)) {
        constexpr bool FAST = decltype(fast_c)::value, LOAD = decltype(load_c)::value;
        const int oy = py + v * rate;
        const bool more = LOAD;
        float4 zv[TW];
        if (BNB) {      // z at this row's output pixels: in flight during the FMAs below
          const float* zrow = p.bb_z + (((size_t)n * p.Ho + oy) * p.Wo + ox0) * p.bb_ldz + c;
#pragma unroll
          for (int tw = 0; tw < TW; ++tw) zv[tw] = ld4(zrow + (size_t)min(tw * rate, p.Wo - 1 - ox0) * p.bb_ldz);
        }
        // issue the loads of the S rows that enter the window for the next output row
        bool nyok[S];
#pragma unroll
        for (int q = 0; q < S; ++q) {
          const int iy = (oy + rate) * S - p.pad_t + (KS - S + q) * rate;
          nyok[q] = more && iy >= 0 && iy < p.H;
          if (FAST) {
            if (LOAD) {
              const float* xrow = ximg + (size_t)min(max(iy, 0), p.H - 1) * p.W * p.ldx;
#pragma unroll
              for (int i = 0; i < SEG; ++i) raw[q][i] = ld4(xrow + coffc[i]);
            }
          } else {
            const float* xrow = ximg + (size_t)iy * p.W * p.ldx;
#pragma unroll
            for (int i = 0; i < SEG; ++i) {
              raw[q][i] = zero4();
              if (nyok[q] && cok[i]) raw[q][i] = (UP && upl) ? dw_up_load(uimg, p.up_w, p.up_ld, dw_lerp(iy, usy, p.up_h), lxs[UP ? i : 0]) : ld4(xrow + coff[i]);
            }
          }
        }
        // FAST: the loads stay in front of the row's FMAs (without the branches of the general path around them the scheduler sinks
        // them next to their use, behind the FMAs)
        if (FAST) __builtin_amdgcn_sched_barrier(0);
        float4 acc[TW];
#pragma unroll
        for (int i = 0; i < TW; ++i) acc[i] = zero4();
        // keep the LDS weight reads inside the row loop: hoisted, the 25 vectors cost 100 VGPRs (1 wave per SIMD)
        if (WLDS) asm volatile("" ::: "memory");
#pragma unroll
        for (int ky = 0; ky < KS; ++ky)
#pragma unroll
          for (int kx = 0; kx < KS; ++kx) {
            const float4 wv = wtap(ky * KS + kx);
#pragma unroll
            for (int tw = 0; tw < TW; ++tw) acc[tw] = fma4(win[ky][tw * S + kx], wv, acc[tw]);
          }
        float* yrow = p.y + (((size_t)n * p.Ho + oy) * p.Wo + ox0) * p.ldy + c;
#pragma unroll
        for (int tw = 0; tw < TW; ++tw) {
          if (FAST || ox0 + tw * rate < p.Wo) {
            float4 vv = acc[tw];
            float* yp = yrow + (size_t)tw * rate * p.ldy;
            if (!FAST && p.accumulate) vv = add4(vv, ld4(yp));
            // (FAST: the kind of store is known at compile time -- forward launches stream, data gradients do not -- so that the
            // stores of a row can be counted)
            if (FAST) { if (PRO != 0) st4_nt(yp, vv); else st4(yp, vv); }
            else if (p.nt) st4_nt(yp, vv); else st4(yp, vv);
            if (BNB) {
              bnb_accumulate(s1, vv, zv[tw], bsc, bsh, bmu, bis, p.bb_act);
            } else {
              s1[0] = add4(s1[0], vv);
              s1[1] = fma4(vv, vv, s1[1]);
            }
          }
        }
        if (!LOAD) return;
        if (FAST) __builtin_amdgcn_sched_barrier(0);       // ... and their consumption behind the row's stores
        // slide the window down by S rows
#pragma unroll
        for (int ky = 0; ky + S < KS; ++ky)
#pragma unroll
          for (int i = 0; i < SEG; ++i) win[ky][i] = win[ky + S][i];
#pragma unroll
        for (int q = 0; q < S; ++q)
#pragma unroll
          for (int i = 0; i < SEG; ++i) {
            float4 a = prologue4<PRO>(raw[q][i], sc, sh, act);
            // FAST: the prologue runs whether or not the pixel is inside the image (left to itself the compiler branches around
            // it, the wait for raw[q][i] lands inside the branch, and the top of the next row waits vmcnt(0) again)
            if (FAST) asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(a.w));
            win[KS - S + q][i] = (nyok[q] && cok[i]) ? a : zero4();
          }
      };
      // (compiled into the forward instantiations only: the rule never sends a data gradient here)
      constexpr bool HAS_FAST = PRO != 0 && !BNB && !UP;
      const bool fast = HAS_FAST && !p.accumulate && p.fast_rows && p.nt && ox0 + (TW - 1) * rate < p.Wo;
      if (HAS_FAST && fast) {
        for (int v = v0; v + 1 < vend; ++v) row(v, std::true_type{}, std::true_type{});
        if (vend > v0) row(vend - 1, std::true_type{}, std::false_type{});
      } else {
        for (int v = v0; v + 1 < vend; ++v) row(v, std::false_type{}, std::true_type{});
        if (vend > v0) row(vend - 1, std::false_type{}, std::false_type{});
      }
    }
  }
  if (p.partials) block_reduce_store<2>(s1, active, pl, cl, p.c4s, p.px, cbase4, p.C, p.partials + (size_t)bx * 2 * p.C);
}

// compiler fence that also materialises v: arithmetic producing v stays above, memory accesses do not cross
__device__ __forceinline__ void pin4(float4& v) {
  asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w) : : "memory");
}

// ------------------------------------------------------------------------------ 5x5 stride 1, rolling output rows
// The 5-row input window of the kernel above costs 5 x (TW+4) vectors of registers at 5x5 (300+ VGPRs with the
// loads in flight: one wave per SIMD, every load latency exposed).  Here the thread keeps the OUTPUT side instead:
// five rows of TW accumulators.  An input row (TW+4 vectors, loaded once, prologue applied once) feeds the five
// output rows it overlaps -- input row j meets output row j-ky with weight row ky -- and output row j-4 is complete
// after input row j, when the accumulator rows shift up by one (register moves, ~10% of the row's FMAs).
// Same sub-lattice walk for atrous rates, same strips/bands/partials as dw_fwd_seg<5, TW, 1>.
template <int TW, int PRO>
__global__ __launch_bounds__(256, 2) void dw5_rows(DwParams p) {
  constexpr int KS = 5, SEG = TW + KS - 1;
  const int b = blockIdx.x;
  const int slab = b / p.nbx;
  const int bx = b - slab * p.nbx;
  const int t = threadIdx.x;
  const int pl = t / p.c4s;
  const int cl = t - pl * p.c4s;
  const bool active = pl < p.px;
  const int cbase4 = slab * p.c4s;
  const int c = (cbase4 + cl) * 4;
  float4 s1[2] = {zero4(), zero4()};
  extern __shared__ __attribute__((aligned(16))) float4 dw_w_lds[];
  for (int i = t; i < KS * KS * p.c4s; i += 256) {
    const int tap = i / p.c4s, lane = i - tap * p.c4s;
    dw_w_lds[i] = ld4(p.w + (size_t)(p.flip ? KS * KS - 1 - tap : tap) * p.C + (cbase4 + lane) * 4);
  }
  __syncthreads();
  if (active) {
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = zero4();
    if (p.scale) { sc = ld4(p.scale + c); sh = ld4(p.shift + c); }
    const int act = p.act;
    const int th = p.th, nbands = p.nbands, rate = p.rate;
    const float4* wl = dw_w_lds + cl;
    const int c4s = p.c4s;
    XcdRange r = xcd_range(p.total, bx, p.nbx, p.px, pl);
    for (int s = r.begin; s < r.end; s += r.step) {
      const int strip = s % p.spr;
      int t2 = s / p.spr;
      const int band = t2 % nbands;
      t2 /= nbands;
      const int phase = t2 % (rate * rate);
      const int n = t2 / (rate * rate);
      const int py = phase / rate, pxo = phase - py * rate;
      const int v0 = band * p.uh / nbands, v1 = (band + 1) * p.uh / nbands;
      const int oy0 = py + v0 * rate;                       // first output row of the band
      if (oy0 >= p.Ho) continue;
      int nv = (p.Ho - oy0 + rate - 1) / rate;              // output rows of this band
      if (nv > v1 - v0) nv = v1 - v0;
      const int ox0 = pxo + strip * TW * rate;
      const int ix0 = ox0 - p.pad_l;
      const float* ximg = p.x + (size_t)n * p.H * p.W * p.ldx + c;
      int coff[SEG];
      bool cok[SEG];
#pragma unroll
      for (int i = 0; i < SEG; ++i) {
        const int ix = ix0 + i * rate;
        cok[i] = ix >= 0 && ix < p.W;
        coff[i] = ix * p.ldx;
      }
      const int iy0 = oy0 - p.pad_t;                        // input row j is iy0 + j * rate
      const int nj = nv + KS - 1;
      float4 acc[KS][TW];
#pragma unroll
      for (int q = 0; q < KS; ++q)
#pragma unroll
        for (int tw = 0; tw < TW; ++tw) acc[q][tw] = zero4();
      // a = the current input row, prologue applied, zero where it lies in the padding
      float4 a[SEG];
      {
        const bool yok = iy0 >= 0 && iy0 < p.H;
        const float* xrow = ximg + (size_t)iy0 * p.W * p.ldx;
#pragma unroll
        for (int i = 0; i < SEG; ++i) {
          a[i] = zero4();
          if (yok && cok[i]) a[i] = ld4(xrow + coff[i]);
        }
#pragma unroll
        for (int i = 0; i < SEG; ++i) {
          const float4 v = prologue4<PRO>(a[i], sc, sh, act);
          a[i] = (yok && cok[i]) ? v : zero4();
        }
      }
      for (int j = 0; j < nj; ++j) {
        // next input row: in flight during this row's FMAs
        float4 nxt[SEG];
        const int iyn = iy0 + (j + 1) * rate;
        const bool nyok = j + 1 < nj && iyn >= 0 && iyn < p.H;
        {
          const float* xrow = ximg + (size_t)iyn * p.W * p.ldx;
#pragma unroll
          for (int i = 0; i < SEG; ++i) {
            nxt[i] = zero4();
            if (nyok && cok[i]) nxt[i] = ld4(xrow + coff[i]);
          }
        }
        const int vdone = j - (KS - 1);                 // output row completed by this input row
        const int oyd = oy0 + vdone * rate;
        // the LDS weight reads stay inside the row loop (hoisted they occupy 100 VGPRs and the accumulators spill)
        asm volatile("" ::: "memory");
        // acc[d] belongs to output row j - 4 + d: weight row ky meets output row j - ky
        // one weight row in use and the next one in flight: left alone, the scheduler issues all 25 LDS reads at the
        // top of the row (100 VGPRs) and the accumulators spill
        float4 wrow[2][KS];
#pragma unroll
        for (int kx = 0; kx < KS; ++kx) wrow[0][kx] = wl[kx * c4s];
#pragma unroll
        for (int ky = 0; ky < KS; ++ky) {
          if (ky + 1 < KS) {
#pragma unroll
            for (int kx = 0; kx < KS; ++kx) wrow[(ky + 1) & 1][kx] = wl[((ky + 1) * KS + kx) * c4s];
          }
#pragma unroll
          for (int kx = 0; kx < KS; ++kx) {
#pragma unroll
            for (int tw = 0; tw < TW; ++tw)
              acc[KS - 1 - ky][tw] = fma4(a[tw + kx], wrow[ky & 1][kx], acc[KS - 1 - ky][tw]);
          }
          // pin this weight row's FMAs here (instruction selection otherwise sinks all 25 x TW of them below the
          // last LDS read, with every weight live)
#pragma unroll
          for (int tw = 0; tw < TW; ++tw) pin4(acc[KS - 1 - ky][tw]);
        }
        if (vdone >= 0) {
          float* yrow = p.y + (((size_t)n * p.Ho + oyd) * p.Wo + ox0) * p.ldy + c;
#pragma unroll
          for (int tw = 0; tw < TW; ++tw) {
            if (ox0 + tw * rate < p.Wo) {
              float4 vv = acc[0][tw];
              float* yp = yrow + (size_t)tw * rate * p.ldy;
              if (p.accumulate) vv = add4(vv, ld4(yp));
              if (p.nt) st4_nt(yp, vv); else st4(yp, vv);
              s1[0] = add4(s1[0], vv);
              s1[1] = fma4(vv, vv, s1[1]);
            }
          }
        }
#pragma unroll
        for (int q = 0; q + 1 < KS; ++q)
#pragma unroll
          for (int tw = 0; tw < TW; ++tw) acc[q][tw] = acc[q + 1][tw];
#pragma unroll
        for (int tw = 0; tw < TW; ++tw) acc[KS - 1][tw] = zero4();
        // the next row's loads have had this row's FMAs to land
#pragma unroll
        for (int i = 0; i < SEG; ++i) {
          const float4 v = prologue4<PRO>(nxt[i], sc, sh, act);
          a[i] = (nyok && cok[i]) ? v : zero4();
        }
      }
    }
  }
  if (p.partials) block_reduce_store<2>(s1, active, pl, cl, p.c4s, p.px, cbase4, p.C, p.partials + (size_t)bx * 2 * p.C);
}

// ------------------------------------------------------------------------------ 5x5 stride 1 weight gradient, rolling rows
// The same row walk for gw[ky][kx] += a[oy + ky, ox + kx] * dy[oy, ox]: the 25 tap accumulators stay put, the input row
// (prologue applied once per element, not once per tap as in the per-pixel gather below: with hard-swish that was
// 25 x 7 VALU operations per output) meets a rolling window of five dy rows.  One partial row [25][C] per workgroup.
template <int TW, int PRO>
__global__ __launch_bounds__(256, 2) void dw5_wgrad_rows(DwParams p) {
  constexpr int KS = 5, SEG = TW + KS - 1;
  const int b = blockIdx.x;
  const int slab = b / p.nbx;
  const int bx = b - slab * p.nbx;
  const int t = threadIdx.x;
  const int pl = t / p.c4s;
  const int cl = t - pl * p.c4s;
  const bool active = pl < p.px;
  const int cbase4 = slab * p.c4s;
  const int c = (cbase4 + cl) * 4;
  float4 wacc[KS * KS];
#pragma unroll
  for (int i = 0; i < KS * KS; ++i) wacc[i] = zero4();
  if (active) {
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = zero4();
    if (p.scale) { sc = ld4(p.scale + c); sh = ld4(p.shift + c); }
    const int act = p.act;
    const int nbands = p.nbands, rate = p.rate;
    XcdRange r = xcd_range(p.total, bx, p.nbx, p.px, pl);
    for (int s = r.begin; s < r.end; s += r.step) {
      const int strip = s % p.spr;
      int t2 = s / p.spr;
      const int band = t2 % nbands;
      t2 /= nbands;
      const int phase = t2 % (rate * rate);
      const int n = t2 / (rate * rate);
      const int py = phase / rate, pxo = phase - py * rate;
      const int v0 = band * p.uh / nbands, v1 = (band + 1) * p.uh / nbands;
      const int oy0 = py + v0 * rate;
      if (oy0 >= p.Ho) continue;
      int nv = (p.Ho - oy0 + rate - 1) / rate;
      if (nv > v1 - v0) nv = v1 - v0;
      const int ox0 = pxo + strip * TW * rate;
      const int ix0 = ox0 - p.pad_l;
      const float* ximg = p.x + (size_t)n * p.H * p.W * p.ldx + c;
      const float* dimg = p.dy + ((size_t)n * p.Ho * p.Wo + ox0) * p.lddy + c;
      int coff[SEG];
      bool cok[SEG];
#pragma unroll
      for (int i = 0; i < SEG; ++i) {
        const int ix = ix0 + i * rate;
        cok[i] = ix >= 0 && ix < p.W;
        coff[i] = ix * p.ldx;
      }
      bool dok[TW];
#pragma unroll
      for (int tw = 0; tw < TW; ++tw) dok[tw] = ox0 + tw * rate < p.Wo;
      const int iy0 = oy0 - p.pad_t;
      const int nj = nv + KS - 1;
      // a = current input row (activated, zero in the padding); dyw[d] = dy row j - 4 + d of this strip
      float4 a[SEG];
      float4 dyw[KS][TW];
#pragma unroll
      for (int d = 0; d < KS; ++d)
#pragma unroll
        for (int tw = 0; tw < TW; ++tw) dyw[d][tw] = zero4();
      {
        const bool yok = iy0 >= 0 && iy0 < p.H;
        const float* xrow = ximg + (size_t)iy0 * p.W * p.ldx;
#pragma unroll
        for (int i = 0; i < SEG; ++i) {
          a[i] = zero4();
          if (yok && cok[i]) a[i] = ld4(xrow + coff[i]);
        }
        const float* drow = dimg + (size_t)oy0 * p.Wo * p.lddy;
#pragma unroll
        for (int tw = 0; tw < TW; ++tw)
          if (dok[tw]) dyw[KS - 1][tw] = ld4(drow + (size_t)tw * rate * p.lddy);
#pragma unroll
        for (int i = 0; i < SEG; ++i) {
          const float4 v = prologue4<PRO>(a[i], sc, sh, act);
          a[i] = (yok && cok[i]) ? v : zero4();
        }
      }
      for (int j = 0; j < nj; ++j) {
        // next input row and next dy row: in flight during this row's FMAs
        float4 nxt[SEG], dyn[TW];
        const int iyn = iy0 + (j + 1) * rate;
        const bool nyok = j + 1 < nj && iyn >= 0 && iyn < p.H;
        const bool dyok = j + 1 < nv;
        {
          const float* xrow = ximg + (size_t)iyn * p.W * p.ldx;
#pragma unroll
          for (int i = 0; i < SEG; ++i) {
            nxt[i] = zero4();
            if (nyok && cok[i]) nxt[i] = ld4(xrow + coff[i]);
          }
          const float* drow = dimg + (size_t)(oy0 + (j + 1) * rate) * p.Wo * p.lddy;
#pragma unroll
          for (int tw = 0; tw < TW; ++tw) {
            dyn[tw] = zero4();
            if (dyok && dok[tw]) dyn[tw] = ld4(drow + (size_t)tw * rate * p.lddy);
          }
        }
        // input row j meets output row j - ky (= dyw[4 - ky]) under weight row ky
#pragma unroll
        for (int ky = 0; ky < KS; ++ky)
#pragma unroll
          for (int kx = 0; kx < KS; ++kx)
#pragma unroll
            for (int tw = 0; tw < TW; ++tw)
              wacc[ky * KS + kx] = fma4(a[tw + kx], dyw[KS - 1 - ky][tw], wacc[ky * KS + kx]);
#pragma unroll
        for (int d = 0; d + 1 < KS; ++d)
#pragma unroll
          for (int tw = 0; tw < TW; ++tw) dyw[d][tw] = dyw[d + 1][tw];
#pragma unroll
        for (int tw = 0; tw < TW; ++tw) dyw[KS - 1][tw] = dyn[tw];
#pragma unroll
        for (int i = 0; i < SEG; ++i) {
          const float4 v = prologue4<PRO>(nxt[i], sc, sh, act);
          a[i] = (nyok && cok[i]) ? v : zero4();
        }
      }
    }
  }
  block_reduce_store<KS * KS>(wacc, active, pl, cl, p.c4s, p.px, cbase4, p.C,
                              p.partials + (size_t)bx * KS * KS * p.C);
}

// ------------------------------------------------------------------------------ backward weight, sliding window
// Same window walk as the forward kernel (the activated input rows live in registers); every output
// row adds win[ky][tw+kx] * dy[tw] into the k*k per-thread tap accumulators.  One partial row
// [k*k][C] per workgroup, summed in a fixed order afterwards.
template <int KS, int TW, int S, int PRO, bool BNA = false, bool UP = false>
__global__ __launch_bounds__(256) void dw_bwd_weight_seg(DwParams p) {
  constexpr int SEG = (TW - 1) * S + KS;
  const int b = blockIdx.x;
  const int slab = b / p.nbx;
  const int bx = b - slab * p.nbx;
  const int t = threadIdx.x;
  const int pl = t / p.c4s;
  const int cl = t - pl * p.c4s;
  const bool active = pl < p.px;
  const int cbase4 = slab * p.c4s;
  const int c = (cbase4 + cl) * 4;
  float4 wacc[KS * KS];
#pragma unroll
  for (int i = 0; i < KS * KS; ++i) wacc[i] = zero4();
  if (active) {
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = zero4();
    if (p.scale) { sc = ld4(p.scale + c); sh = ld4(p.shift + c); }
    const int act = p.act;
    const int th = p.th, nbands = p.nbands, rate = p.rate;
    // BNA: dz = bA * dy * act'(z * bsc + bsh) - bC * z + bD  (bA = c0, bC = c0 * invstd * c2, bD = bC * mean - c0 * c1)
    float4 bA = zero4(), bC = zero4(), bD = zero4(), bsc = make_float4(1.f, 1.f, 1.f, 1.f), bsh = zero4();
    if (BNA) {
      if (p.fa_scale) { bsc = ld4(p.fa_scale + c); bsh = ld4(p.fa_shift + c); }
      const float4 mu = ld4(p.fa_mean + c), is = ld4(p.fa_invstd + c);
      const float4 c0 = ld4(p.fa_coef + c), c1 = ld4(p.fa_coef + p.C + c), c2 = ld4(p.fa_coef + 2 * p.C + c);
      bA = c0;
      bC = mul4(mul4(c0, is), c2);
      bD = make_float4(bC.x * mu.x - c0.x * c1.x, bC.y * mu.y - c0.y * c1.y, bC.z * mu.z - c0.z * c1.z, bC.w * mu.w - c0.w * c1.w);
    }
    XcdRange r = xcd_range(p.total, bx, p.nbx, p.px, pl);
    for (int s = r.begin; s < r.end; s += r.step) {
      const int strip = s % p.spr;
      int t2 = s / p.spr;
      const int band = t2 % nbands;
      t2 /= nbands;
      const int phase = t2 % (rate * rate);      // sub-lattice (py, px); 0 when rate == 1
      const int n = t2 / (rate * rate);
      const int py = phase / rate, pxo = phase - py * rate;
      const int u0 = strip * TW;                 // first sub-lattice column of this strip
      const int v0 = band * p.uh / nbands, v1 = (band + 1) * p.uh / nbands;
      const int ox0 = pxo + u0 * rate;
      const int ix0 = ox0 * S - p.pad_l;
      const float* ximg = p.x + (size_t)n * p.H * p.W * p.ldx + c;
      int coff[SEG];
      bool cok[SEG];
#pragma unroll
      for (int i = 0; i < SEG; ++i) {
        const int ix = ix0 + i * rate;
        cok[i] = ix >= 0 && ix < p.W;
        coff[i] = ix * p.ldx;
      }
      // UP: the column coefficients of the strip's input pixels in the low-resolution map (rows: per input row below)
      DwLerp lxs[UP ? SEG : 1];
      const bool upl = UP && c < p.up_C;
      const float* uimg = nullptr;
      float usy = 0.f;
      if (UP) {
        const float usx = (float)p.up_w / (float)p.W;
        usy = (float)p.up_h / (float)p.H;
#pragma unroll
        for (int i = 0; i < SEG; ++i) lxs[i] = dw_lerp(min(max(ix0 + i * rate, 0), p.W - 1), usx, p.up_w);
        uimg = p.up_x + (size_t)n * p.up_h * p.up_w * p.up_ld + (upl ? c : 0);
      }
      float4 win[KS][SEG];
      float4 raw[S][SEG];
      // prime the window with the KS rows of the first output row (exec-masked loads, math afterwards)
      {
        const int oy = py + v0 * rate;
#pragma unroll
        for (int ky = 0; ky < KS; ++ky) {
          const int iy = oy * S - p.pad_t + ky * rate;
          const bool yok = iy >= 0 && iy < p.H;
          const float* xrow = ximg + (size_t)iy * p.W * p.ldx;
#pragma unroll
          for (int i = 0; i < SEG; ++i) {
            win[ky][i] = zero4();
            if (yok && cok[i]) win[ky][i] = (UP && upl) ? dw_up_load(uimg, p.up_w, p.up_ld, dw_lerp(iy, usy, p.up_h), lxs[UP ? i : 0]) : ld4(xrow + coff[i]);
          }
        }
#pragma unroll
        for (int ky = 0; ky < KS; ++ky) {
          const int iy = oy * S - p.pad_t + ky * rate;
          const bool yok = iy >= 0 && iy < p.H;
#pragma unroll
          for (int i = 0; i < SEG; ++i) {
            const float4 a = prologue4<PRO>(win[ky][i], sc, sh, act);
            win[ky][i] = (yok && cok[i]) ? a : zero4();
          }
        }
      }
      for (int v = v0; v < v1; ++v) {
        const int oy = py + v * rate;
        if (oy >= p.Ho) break;
        const bool more = v + 1 < v1 && oy + rate < p.Ho;
        // issue the loads of the S rows that enter the window for the next output row
        bool nyok[S];
#pragma unroll
        for (int q = 0; q < S; ++q) {
          const int iy = (oy + rate) * S - p.pad_t + (KS - S + q) * rate;
          nyok[q] = more && iy >= 0 && iy < p.H;
          const float* xrow = ximg + (size_t)iy * p.W * p.ldx;
#pragma unroll
          for (int i = 0; i < SEG; ++i) {
            raw[q][i] = zero4();
            if (nyok[q] && cok[i]) raw[q][i] = (UP && upl) ? dw_up_load(uimg, p.up_w, p.up_ld, dw_lerp(iy, usy, p.up_h), lxs[UP ? i : 0]) : ld4(xrow + coff[i]);
          }
        }
        // gradient of the raw conv output for this row's strip
        float4 dyv[TW];
        const float* drow = p.dy + (((size_t)n * p.Ho + oy) * p.Wo + ox0) * p.lddy + c;
#pragma unroll
        for (int tw = 0; tw < TW; ++tw) {
          dyv[tw] = zero4();
          if (ox0 + tw * rate < p.Wo) dyv[tw] = ld4(drow + (size_t)tw * rate * p.lddy);
        }
        if (BNA) {
          const size_t pix = ((size_t)n * p.Ho + oy) * p.Wo + ox0;
          const float* zrow = p.fa_z + pix * p.fa_ldz + c;
          float4 zv[TW];
#pragma unroll
          for (int tw = 0; tw < TW; ++tw) {
            zv[tw] = zero4();
            if (ox0 + tw * rate < p.Wo) zv[tw] = ld4(zrow + (size_t)tw * rate * p.fa_ldz);
          }
          const int fact = p.fa_act;
#pragma unroll
          for (int tw = 0; tw < TW; ++tw) {
            const float4 z = zv[tw], g = dyv[tw];
            const float4 u = fma4(z, bsc, bsh);
            const float4 v = make_float4(fmaf(bA.x, g.x * act_grad(u.x, fact), fmaf(-bC.x, z.x, bD.x)),
                                         fmaf(bA.y, g.y * act_grad(u.y, fact), fmaf(-bC.y, z.y, bD.y)),
                                         fmaf(bA.z, g.z * act_grad(u.z, fact), fmaf(-bC.z, z.z, bD.z)),
                                         fmaf(bA.w, g.w * act_grad(u.w, fact), fmaf(-bC.w, z.w, bD.w)));
            const bool ok = ox0 + tw * rate < p.Wo;
            if (ok && p.fa_dz) st4(p.fa_dz + (pix + (size_t)tw * rate) * p.fa_lddz + c, v);
            dyv[tw] = ok ? v : zero4();
          }
        }
#pragma unroll
        for (int ky = 0; ky < KS; ++ky)
#pragma unroll
          for (int kx = 0; kx < KS; ++kx)
#pragma unroll
            for (int tw = 0; tw < TW; ++tw) wacc[ky * KS + kx] = fma4(win[ky][tw * S + kx], dyv[tw], wacc[ky * KS + kx]);
        // slide the window down by S rows
#pragma unroll
        for (int ky = 0; ky + S < KS; ++ky)
#pragma unroll
          for (int i = 0; i < SEG; ++i) win[ky][i] = win[ky + S][i];
#pragma unroll
        for (int q = 0; q < S; ++q)
#pragma unroll
          for (int i = 0; i < SEG; ++i) {
            const float4 a = prologue4<PRO>(raw[q][i], sc, sh, act);
            win[KS - S + q][i] = (nyok[q] && cok[i]) ? a : zero4();
          }
      }
    }
  }
  block_reduce_store<KS * KS>(wacc, active, pl, cl, p.c4s, p.px, cbase4, p.C,
                              p.partials + (size_t)bx * KS * KS * p.C);
}

// ------------------------------------------------------------------------------ forward, any rate / stride
// Per-pixel gather for the geometries the window kernel does not cover (sub-lattice narrower than a
// strip: the ASPP rates 12/18 on a 33x33 map, where most taps fall into the zero padding).  Two output
// pixels per thread and iteration: all their in-range taps are exec-masked 16-B loads issued before any
// arithmetic (taps in the padding issue nothing), served by the XCD's L2.
template <int KS, int PRO>
__global__ __launch_bounds__(256) void dw_fwd_gather(DwParams p) {
  constexpr int TI = KS == 3 ? 2 : 1;   // 5x5: 25 taps per pixel already fill the registers
  const int b = blockIdx.x;
  const int slab = b / p.nbx;
  const int bx = b - slab * p.nbx;
  const int t = threadIdx.x;
  const int pl = t / p.c4s;
  const int cl = t - pl * p.c4s;
  const bool active = pl < p.px;
  const int cbase4 = slab * p.c4s;
  const int c = (cbase4 + cl) * 4;
  float4 s1[2] = {zero4(), zero4()};
  if (active) {
    float4 wreg[KS * KS];
#pragma unroll
    for (int i = 0; i < KS * KS; ++i) wreg[i] = ld4(p.w + (size_t)(p.flip ? KS * KS - 1 - i : i) * p.C + c);
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = zero4();
    if (PRO && p.scale) { sc = ld4(p.scale + c); sh = ld4(p.shift + c); }
    const int act = p.act;
    XcdRange r = xcd_range(p.total, bx, p.nbx, p.px, pl);
    for (int s0 = r.begin; s0 < r.end; s0 += TI * r.step) {
      float4 v[TI][KS * KS];
      bool ok[TI][KS * KS];
      float* yp[TI];
      bool live[TI];
#pragma unroll
      for (int ti = 0; ti < TI; ++ti) {
        const int s = s0 + ti * r.step;
        live[ti] = s < r.end;
        const int ox = s % p.Wo;
        const int row = s / p.Wo;
        const int oy = row % p.Ho;
        const int n = row / p.Ho;
        const float* ximg = p.x + (size_t)n * p.H * p.W * p.ldx + c;
        yp[ti] = p.y + (((size_t)n * p.Ho + oy) * p.Wo + ox) * p.ldy + c;
#pragma unroll
        for (int ky = 0; ky < KS; ++ky) {
          const int iy = oy * p.stride - p.pad_t + ky * p.rate;
          const bool yok = live[ti] && iy >= 0 && iy < p.H;
#pragma unroll
          for (int kx = 0; kx < KS; ++kx) {
            const int ix = ox * p.stride - p.pad_l + kx * p.rate;
            const bool o = yok && ix >= 0 && ix < p.W;
            ok[ti][ky * KS + kx] = o;
            v[ti][ky * KS + kx] = zero4();
            if (o) v[ti][ky * KS + kx] = ld4(ximg + ((size_t)iy * p.W + ix) * p.ldx);
          }
        }
      }
#pragma unroll
      for (int ti = 0; ti < TI; ++ti) {
        float4 acc = zero4();
#pragma unroll
        for (int i = 0; i < KS * KS; ++i) {
          const float4 a = prologue4<PRO>(v[ti][i], sc, sh, act);
          acc = fma4(ok[ti][i] ? a : zero4(), wreg[i], acc);
        }
        if (live[ti]) {
          if (p.accumulate) acc = add4(acc, ld4(yp[ti]));
          if (p.nt) st4_nt(yp[ti], acc); else st4(yp[ti], acc);
          s1[0] = add4(s1[0], acc);
          s1[1] = fma4(acc, acc, s1[1]);
        }
      }
    }
  }
  if (p.partials) block_reduce_store<2>(s1, active, pl, cl, p.c4s, p.px, cbase4, p.C, p.partials + (size_t)bx * 2 * p.C);
}

// ------------------------------------------------------------------------------ forward, largest atrous rate
// When 2*rate >= max(H, W) (ASPP rate 18 on the 33x33 map of a 513x513 input at OS16) every residue class
// (py, px) of the rate x rate sub-lattice holds at most 2 x 2 pixels, and on a sub-lattice the atrous 3x3 is
// a dense 3x3 with padding 1: the <= 4 outputs of a class depend on exactly its <= 4 inputs.  One thread
// owns one class (x 4 channels): 4 loads feed 4 outputs, i.e. every input element crosses the CU's load
// path ONCE (the per-pixel gather fetches it 3.6 times through L1 misses and is bound by the ~24 GB/s/CU load
// path, not by HBM).  Two classes per thread are in flight (8 loads issued before any arithmetic); all tap
// indices are compile-time, so the 9 weights live in registers.  HBM-bound: reads x once, writes y once.
#ifndef DL3P_LAT2_PER_CU
#define DL3P_LAT2_PER_CU 3
#endif
struct Lat2Item { unsigned pix[2][2]; bool rv[2], cv[2]; };

__device__ __forceinline__ void lat2_decode(int s, int end, int rate, int H, int W, Lat2Item& it) {
  const bool live = s < end;
  const int pxo = s % rate;
  const int t2 = s / rate;
  const int py = t2 % rate;
  const int n = t2 / rate;
  it.rv[0] = live; it.rv[1] = live && py + rate < H;
  it.cv[0] = live; it.cv[1] = live && pxo + rate < W;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
      it.pix[a][b] = ((unsigned)n * H + (unsigned)(py + a * rate)) * W + (unsigned)(pxo + b * rate);
}

template <int PRO>
__global__ __launch_bounds__(256) void dw_fwd_lattice2(DwParams p) {
  constexpr int TI = 1;   // residue classes per pipeline stage (2 was slower: fewer, longer stages overlap worse)
  const int b = blockIdx.x;
  const int slab = b / p.nbx;
  const int bx = b - slab * p.nbx;
  const int t = threadIdx.x;
  const int pl = t / p.c4s;
  const int cl = t - pl * p.c4s;
  const bool active = pl < p.px;
  const int cbase4 = slab * p.c4s;
  const int c = (cbase4 + cl) * 4;
  float4 s1[2] = {zero4(), zero4()};
  if (active) {
    const int rate = p.rate, H = p.H, W = p.W;
    const unsigned ldx = (unsigned)p.ldx, ldy = (unsigned)p.ldy;
    const float* xb = p.x + c;
    float* yb = p.y + c;
    XcdRange r = xcd_range(p.total, bx, p.nbx, p.px, pl);
    // software pipeline: the loads of the NEXT stage are issued before the current stage is computed and
    // stored, so reads and writes of different waves/iterations overlap instead of running in two phases
    Lat2Item cur[TI], nxt[TI];
    float4 in[TI][2][2], inn[TI][2][2];
#pragma unroll
    for (int ti = 0; ti < TI; ++ti) {
      lat2_decode(r.begin + ti * r.step, r.end, rate, H, W, cur[ti]);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int bb = 0; bb < 2; ++bb) {
          // unconditional load (pixel 0 for pixels outside the map; masked to zero below): no branch per load
          in[ti][a][bb] = ld4(xb + ((cur[ti].rv[a] && cur[ti].cv[bb]) ? cur[ti].pix[a][bb] : 0u) * ldx);
        }
    }
    float4 wreg[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) wreg[i] = ld4(p.w + (size_t)(p.flip ? 8 - i : i) * p.C + c);
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = zero4();
    if (PRO && p.scale) { sc = ld4(p.scale + c); sh = ld4(p.shift + c); }
    const int act = p.act;
    // streaming (non-temporal) stores: y is not re-read by this kernel and must not evict x, which the two other
    // atrous branches have just left in L2 / Infinity Cache (measured in-step: 9.6 -> 8.8 us)
    const bool nt_store = p.flip == 0;
    for (int s = r.begin; s < r.end; s += TI * r.step) {
#pragma unroll
      for (int ti = 0; ti < TI; ++ti) {
        lat2_decode(s + (TI + ti) * r.step, r.end, rate, H, W, nxt[ti]);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int bb = 0; bb < 2; ++bb) {
            inn[ti][a][bb] = ld4(xb + ((nxt[ti].rv[a] && nxt[ti].cv[bb]) ? nxt[ti].pix[a][bb] : 0u) * ldx);
          }
      }
#pragma unroll
      for (int ti = 0; ti < TI; ++ti) {
        float4 av[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int bb = 0; bb < 2; ++bb) {
            const float4 v = prologue4<PRO>(in[ti][a][bb], sc, sh, act);
            av[a][bb] = (cur[ti].rv[a] && cur[ti].cv[bb]) ? v : zero4();     // zero padding is in activation space
          }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int bb = 0; bb < 2; ++bb) {
            // output (a,bb) of the class: tap (ky,kx) = (a'-a+1, b'-bb+1) reads input (a',b')
            float4 acc = zero4();
#pragma unroll
            for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
              for (int b2 = 0; b2 < 2; ++b2) acc = fma4(av[a2][b2], wreg[(a2 - a + 1) * 3 + (b2 - bb + 1)], acc);
            if (cur[ti].rv[a] && cur[ti].cv[bb]) {
              float* yp = yb + cur[ti].pix[a][bb] * ldy;
              if (p.accumulate) acc = add4(acc, ld4(yp));
              if (nt_store) st4_nt(yp, acc);
              else st4(yp, acc);
              s1[0] = add4(s1[0], acc);
              s1[1] = fma4(acc, acc, s1[1]);
            }
          }
      }
#pragma unroll
      for (int ti = 0; ti < TI; ++ti) {
        cur[ti] = nxt[ti];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int bb = 0; bb < 2; ++bb) in[ti][a][bb] = inn[ti][a][bb];
      }
    }
  }
  if (p.partials) block_reduce_store<2>(s1, active, pl, cl, p.c4s, p.px, cbase4, p.C, p.partials + (size_t)bx * 2 * p.C);
}

// ------------------------------------------------------------------------------ forward, the atrous rate below it
// 3*rate >= max(H, W) > 2*rate (ASPP rate 12 on the 33x33 map, rate 36 on the 97x97 map of a 769x769 input at OS 8): a
// residue class holds at most 3 x 3 pixels and the atrous 3x3 is a dense 3x3 with padding 1 ON the class: its <= 9 outputs
// depend on exactly its <= 9 inputs.  The same scheme as dw_fwd_lattice2 with a 3 x 3 class per thread: 9 loads feed 9 outputs,
// every input element crosses the load path once (the per-pixel gather it replaces fetched each element ~5 times and sat at
// 0.35 of the HBM rate), one class ahead in flight, tap indices compile-time, weights in registers, streaming stores.
struct Lat3Item { unsigned base; bool rv[3], cv[3]; };

__device__ __forceinline__ void lat3_decode(int s, int end, int rate, int H, int W, Lat3Item& it) {
  const bool live = s < end;
  const int pxo = s % rate;
  const int t2 = s / rate;
  const int py = t2 % rate;
  const int n = t2 / rate;
  it.base = ((unsigned)n * H + (unsigned)py) * W + (unsigned)pxo;       // pixel (a, b) of the class = base + (a * W + b) * rate
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    it.rv[a] = live && py + a * rate < H;
    it.cv[a] = live && pxo + a * rate < W;
  }
}

template <int PRO>
__global__ __launch_bounds__(256) void dw_fwd_lattice3(DwParams p) {
  const int b = blockIdx.x;
  const int slab = b / p.nbx;
  const int bx = b - slab * p.nbx;
  const int t = threadIdx.x;
  const int pl = t / p.c4s;
  const int cl = t - pl * p.c4s;
  const bool active = pl < p.px;
  const int cbase4 = slab * p.c4s;
  const int c = (cbase4 + cl) * 4;
  float4 s1[2] = {zero4(), zero4()};
  if (active) {
    const int rate = p.rate, H = p.H, W = p.W;
    const unsigned ldx = (unsigned)p.ldx, ldy = (unsigned)p.ldy;
    const unsigned rW = (unsigned)rate * (unsigned)W, r1 = (unsigned)rate;
    const float* xb = p.x + c;
    float* yb = p.y + c;
    XcdRange r = xcd_range(p.total, bx, p.nbx, p.px, pl);
    Lat3Item cur, nxt;
    float4 in[3][3], inn[3][3];
    lat3_decode(r.begin, r.end, rate, H, W, cur);
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int bb = 0; bb < 3; ++bb)          // unconditional loads (pixel 0 outside the map, masked below): no branch per load
        in[a][bb] = ld4(xb + ((cur.rv[a] && cur.cv[bb]) ? cur.base + a * rW + bb * r1 : 0u) * ldx);
    float4 wreg[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) wreg[i] = ld4(p.w + (size_t)(p.flip ? 8 - i : i) * p.C + c);
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = zero4();
    if (PRO && p.scale) { sc = ld4(p.scale + c); sh = ld4(p.shift + c); }
    const int act = p.act;
    const bool nt_store = p.flip == 0;
    for (int s = r.begin; s < r.end; s += r.step) {
      lat3_decode(s + r.step, r.end, rate, H, W, nxt);
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int bb = 0; bb < 3; ++bb)
          inn[a][bb] = ld4(xb + ((nxt.rv[a] && nxt.cv[bb]) ? nxt.base + a * rW + bb * r1 : 0u) * ldx);
      float4 av[3][3];
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int bb = 0; bb < 3; ++bb) {
          const float4 v = prologue4<PRO>(in[a][bb], sc, sh, act);
          av[a][bb] = (cur.rv[a] && cur.cv[bb]) ? v : zero4();          // zero padding is in activation space
        }
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int bb = 0; bb < 3; ++bb) {
          // output (a, bb) of the class: tap (ky, kx) = (a' - a + 1, b' - bb + 1) reads input (a', b'), |a' - a| <= 1
          float4 acc = zero4();
#pragma unroll
          for (int a2 = 0; a2 < 3; ++a2)
#pragma unroll
            for (int b2 = 0; b2 < 3; ++b2)
              if (a2 - a >= -1 && a2 - a <= 1 && b2 - bb >= -1 && b2 - bb <= 1)
                acc = fma4(av[a2][b2], wreg[(a2 - a + 1) * 3 + (b2 - bb + 1)], acc);
          if (cur.rv[a] && cur.cv[bb]) {
            float* yp = yb + (cur.base + a * rW + bb * r1) * ldy;
            if (p.accumulate) acc = add4(acc, ld4(yp));
            if (nt_store) st4_nt(yp, acc);
            else st4(yp, acc);
            s1[0] = add4(s1[0], acc);
            s1[1] = fma4(acc, acc, s1[1]);
          }
        }
      cur = nxt;
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int bb = 0; bb < 3; ++bb) in[a][bb] = inn[a][bb];
    }
  }
  if (p.partials) block_reduce_store<2>(s1, active, pl, cl, p.c4s, p.px, cbase4, p.C, p.partials + (size_t)bx * 2 * p.C);
}

// ------------------------------------------------------------------------------ backward data, stride > 1
// gx[n,iy,ix] = sum_taps w[ky,kx] * dy[n,(iy+pad_t-ky*r)/s,(ix+pad_l-kx*r)/s] where divisible.
// Here (H,W) are the conv INPUT dims (the output of this kernel) and (Ho,Wo) the dy dims.
template <int KS>
__global__ __launch_bounds__(256) void dw_bwd_data_strided(DwParams p) {
  const int b = blockIdx.x;
  const int slab = b / p.nbx;
  const int bx = b - slab * p.nbx;
  const int t = threadIdx.x;
  const int pl = t / p.c4s;
  const int cl = t - pl * p.c4s;
  if (pl >= p.px) return;
  const int c = (slab * p.c4s + cl) * 4;
  float4 wreg[KS * KS];
#pragma unroll
  for (int i = 0; i < KS * KS; ++i) wreg[i] = ld4(p.w + (size_t)i * p.C + c);
  XcdRange r = xcd_range(p.total, bx, p.nbx, p.px, pl);
  for (int s = r.begin; s < r.end; s += r.step) {
    const int ix = s % p.W;
    const int row = s / p.W;
    const int iy = row % p.H;
    const int n = row / p.H;
    const float* dimg = p.dy + (size_t)n * p.Ho * p.Wo * p.lddy + c;
    float4 acc = zero4();
#pragma unroll
    for (int ky = 0; ky < KS; ++ky) {
      const int ty = iy + p.pad_t - ky * p.rate;
      if (ty < 0 || ty % p.stride) continue;
      const int oy = ty / p.stride;
      if (oy >= p.Ho) continue;
#pragma unroll
      for (int kx = 0; kx < KS; ++kx) {
        const int tx = ix + p.pad_l - kx * p.rate;
        if (tx < 0 || tx % p.stride) continue;
        const int ox = tx / p.stride;
        if (ox >= p.Wo) continue;
        acc = fma4(ld4(dimg + ((size_t)oy * p.Wo + ox) * p.lddy), wreg[ky * KS + kx], acc);
      }
    }
    float* gp = p.y + (((size_t)n * p.H + iy) * p.W + ix) * p.ldy + c;
    if (p.accumulate) {
      float4 o = ld4(gp);
      acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w;
    }
    st4(gp, acc);
  }
}

// stride-2 backward data (rate 1), by 2x2 quads of input pixels.  With u = iy + pad_t, only taps ky of u's
// parity reach input row iy (oy = (u - ky) / 2), so the quad rows u in {2a, 2a+1}, columns v in {2b, 2b+1}
// read the same (J+1)^2 dy pixels (rows a-j, columns b-i, J = (KS-1)/2): all loads are issued first on
// clamped addresses (no branch, no % or / per tap), 9 (25) FMAs per quad, 4 stores.
template <int KS, bool BNB = false>
__global__ __launch_bounds__(256) void dw_bwd_data_s2(DwParams p) {
  constexpr int J = (KS - 1) / 2;
  const int b = blockIdx.x;
  const int slab = b / p.nbx;
  const int bx = b - slab * p.nbx;
  const int t = threadIdx.x;
  const int pl = t / p.c4s;
  const int cl = t - pl * p.c4s;
  const bool active = pl < p.px;
  const int cbase4 = slab * p.c4s;
  const int c = (cbase4 + cl) * 4;
  float4 s1[2] = {zero4(), zero4()};
  if (active) {
  float4 bsc = zero4(), bsh = zero4(), bmu = zero4(), bis = zero4();
  if (BNB) { bsc = ld4(p.bb_scale + c); bsh = ld4(p.bb_shift + c); bmu = ld4(p.bb_mean + c); bis = ld4(p.bb_invstd + c); }
  float4 wreg[KS * KS];
#pragma unroll
  for (int i = 0; i < KS * KS; ++i) wreg[i] = ld4(p.w + (size_t)i * p.C + c);
  const int QA = p.th, QB = p.spr;          // quads per image column / row (host)
  XcdRange r = xcd_range(p.total, bx, p.nbx, p.px, pl);
  for (int s = r.begin; s < r.end; s += r.step) {
    const int qb = s % QB;
    const int row = s / QB;
    const int qa = row % QA;
    const int n = row / QA;
    const float* dimg = p.dy + (size_t)n * p.Ho * p.Wo * p.lddy + c;
    float4 d[J + 1][J + 1];
#pragma unroll
    for (int j = 0; j <= J; ++j) {
      const int oy = qa - j;
      const bool yok = oy >= 0 && oy < p.Ho;
      const int oyc = min(max(oy, 0), p.Ho - 1);
#pragma unroll
      for (int i = 0; i <= J; ++i) {
        const int ox = qb - i;
        const bool ok = yok && ox >= 0 && ox < p.Wo;
        const int oxc = min(max(ox, 0), p.Wo - 1);
        const float4 v = ld4(dimg + ((size_t)oyc * p.Wo + oxc) * p.lddy);
        d[j][i] = ok ? v : zero4();
      }
    }
#pragma unroll
    for (int du = 0; du < 2; ++du) {
      const int iy = 2 * qa + du - p.pad_t;
#pragma unroll
      for (int dv = 0; dv < 2; ++dv) {
        const int ix = 2 * qb + dv - p.pad_l;
        float4 acc = zero4();
#pragma unroll
        for (int j = 0; j <= J; ++j) {
          if (du + 2 * j >= KS) continue;
#pragma unroll
          for (int i = 0; i <= J; ++i) {
            if (dv + 2 * i >= KS) continue;
            acc = fma4(d[j][i], wreg[(du + 2 * j) * KS + dv + 2 * i], acc);
          }
        }
        if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) {
          const size_t pix = ((size_t)n * p.H + iy) * p.W + ix;
          float* gp = p.y + pix * p.ldy + c;
          if (p.accumulate) acc = add4(acc, ld4(gp));
          st4(gp, acc);
          if (BNB) bnb_accumulate(s1, acc, ld4(p.bb_z + pix * p.bb_ldz + c), bsc, bsh, bmu, bis, p.bb_act);
        }
      }
    }
  }
  }
  if (BNB) block_reduce_store<2>(s1, active, pl, cl, p.c4s, p.px, cbase4, p.C, p.partials + (size_t)bx * 2 * p.C);
}

// ------------------------------------------------------------------------------ backward weight
// gw[tap][c] = sum_{n,oy,ox} a[n, oy*s-pad+ky*r, ox*s-pad+kx*r, c] * dy[n,oy,ox,c]; a = act(x*scale+shift).
// Per-thread tap accumulators over a persistent loop, then one partial row [k*k][C] per workgroup.
template <int KS, int PRO>
__global__ __launch_bounds__(256) void dw_bwd_weight(DwParams p) {
  const int b = blockIdx.x;
  const int slab = b / p.nbx;
  const int bx = b - slab * p.nbx;
  const int t = threadIdx.x;
  const int pl = t / p.c4s;
  const int cl = t - pl * p.c4s;
  const bool active = pl < p.px;
  const int cbase4 = slab * p.c4s;
  const int c = (cbase4 + cl) * 4;
  float4 wacc[KS * KS];
#pragma unroll
  for (int i = 0; i < KS * KS; ++i) wacc[i] = zero4();
  if (active) {
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = zero4();
    if (p.scale) { sc = ld4(p.scale + c); sh = ld4(p.shift + c); }
    const int act = p.act;
    XcdRange r = xcd_range(p.total, bx, p.nbx, p.px, pl);
    for (int s = r.begin; s < r.end; s += r.step) {
      const int ox = s % p.Wo;
      const int row = s / p.Wo;
      const int oy = row % p.Ho;
      const int n = row / p.Ho;
      const float4 g = ld4(p.dy + (((size_t)n * p.Ho + oy) * p.Wo + ox) * p.lddy + c);
      const float* ximg = p.x + (size_t)n * p.H * p.W * p.ldx + c;
      float4 v[KS * KS];
      bool ok[KS * KS];
#pragma unroll
      for (int ky = 0; ky < KS; ++ky) {
        const int iy = oy * p.stride - p.pad_t + ky * p.rate;
        const bool yok = iy >= 0 && iy < p.H;
#pragma unroll
        for (int kx = 0; kx < KS; ++kx) {
          const int ix = ox * p.stride - p.pad_l + kx * p.rate;
          const bool o = yok && ix >= 0 && ix < p.W;
          ok[ky * KS + kx] = o;
          v[ky * KS + kx] = zero4();
          if (o) v[ky * KS + kx] = ld4(ximg + ((size_t)iy * p.W + ix) * p.ldx);
        }
      }
#pragma unroll
      for (int i = 0; i < KS * KS; ++i) {
        const float4 a = prologue4<PRO>(v[i], sc, sh, act);
        if (ok[i]) wacc[i] = fma4(a, g, wacc[i]);
      }
    }
  }
  block_reduce_store<KS * KS>(wacc, active, pl, cl, p.c4s, p.px, cbase4, p.C,
                              p.partials + (size_t)bx * KS * KS * p.C);
}

// ------------------------------------------------------------------------------ host side
static int check_dw_common(const char* fn, const void* x, int ldx, int C, int k) {
  DL3P_CHECK_ARG(C > 0 && C % 4 == 0, "%s: C=%d must be a positive multiple of 4", fn, C);
  DL3P_CHECK_ARG(ldx % 4 == 0 && ldx >= C, "%s: ld=%d must be a multiple of 4 and >= C", fn, ldx);
  DL3P_CHECK_ARG(aligned16(x), "%s: tensor pointer must be 16-byte aligned", fn);
  DL3P_CHECK_ARG(k == 3 || k == 5, "%s: kernel size %d unsupported (3 or 5)", fn, k);
  return DL3P_OK;
}

// rows per band: as tall as possible (less halo re-reading: at th = 1 every input row is fetched 3x from L2 and the
// 33x33 layers were L2-bound) while every CU still gets 1-2 workgroup-iterations (measured optimum 256-512)
static int pick_band(long long items_per_row_band, int rows, int px, int nslab) {
  static const int want_env = getenv("DL3P_DW_WANT") ? atoi(getenv("DL3P_DW_WANT")) : DL3P_NUM_CUS * 3 / 2;
  const long long want = want_env;                               // workgroup-iterations wanted overall
  static const int force = getenv("DL3P_DW_BAND") ? atoi(getenv("DL3P_DW_BAND")) : 0;
  if (force && rows <= 40) return force;
  int th = 16;
  while (th > 1 && (items_per_row_band * ceil_div(rows, th) / px) * nslab < want) th >>= 1;
  return th;
}

// variant + work decomposition of a forward depthwise launch
//   kind 1: window kernel TW=4, stride 1 (any rate, on the rate x rate sub-lattices)
//   kind 2: window kernel TW=2, stride 2, rate 1
//   kind 0: per-pixel gather (stride > 1 with rate > 1, or maps narrower than a strip)
// strip width of the 5x5 stride-1 rolling-row kernel: 4 columns per thread on the narrow maps (33 wide and the 17-wide
// sub-lattices of rate 2: 37 vs 55 us at 33x33x672), 2 on the wide ones (65x65x120: 49 vs 58 us, the bands stay taller).
// DL3P_DW5_ROWS = 2 | 4 forces one, 0 selects the older register-window kernel (1 wave per SIMD, kept for comparison).
static int dw5_rows_tw(int uw) {
  static const int v = getenv("DL3P_DW5_ROWS") ? atoi(getenv("DL3P_DW5_ROWS")) : -1;
  if (v >= 0) return v;
  return uw <= 40 ? 4 : 2;
}
static int dw5_tw(int uw) { return dw5_rows_tw(uw) == 4 ? 4 : 2; }

// Measured plan choices of the window kernels for the depthwise shapes of the BASELINE graphs (scripts/tune_dw.py writes
// dw_tuned.h): role 0 forward, 1 data gradient, 2 data gradient + fused BatchNorm sums, 3 weight gradient; the key is the
// geometry the planner sees (for the data gradient that is the flipped problem: H x W of dy).  per_cu: persistent
// workgroups per CU; want: workgroup-iterations the row-band split aims for; maxth: tallest band.  0 = keep the default.
struct DwTuned { int role, N, H, W, C, k, stride, rate, per_cu, want, maxth, tw; };   // tw: strip width of the 3x3 stride-1 window kernels (0 / 4, or 2)
#include "dw_tuned.h"
int dl3p_dw_force_per_cu = 0, dl3p_dw_force_want = 0, dl3p_dw_force_maxth = 0, dl3p_dw_force_tw = 0, dl3p_dw_use_table = -1;
static const DwTuned* dw_tuned_lookup(const DwParams& p) {
  if (dl3p_dw_use_table < 0) dl3p_dw_use_table = getenv("DL3P_DW_TUNED") ? atoi(getenv("DL3P_DW_TUNED")) : 1;
  if (!dl3p_dw_use_table) return nullptr;
  const int role = p.bb_z ? 2 : (p.flip ? 1 : (p.dy ? 3 : 0));
  for (size_t i = 0; i < sizeof(g_dw_tuned) / sizeof(g_dw_tuned[0]); ++i) {
    const DwTuned& e = g_dw_tuned[i];
    if (e.role == role && e.N == p.N && e.H == p.H && e.W == p.W && e.C == p.C && e.k == (p.ks5 ? 5 : 3) && e.stride == p.stride &&
        e.rate == p.rate)
      return &e;
  }
  return nullptr;
}

static int fwd_plan(DwParams& p, int per_cu = 8, bool window5 = false, int tw5 = 0) {
  int kind = 0;
  int t_want = 0, t_maxth = 0, t_tw = 0;
  if (const DwTuned* e = dw_tuned_lookup(p)) {
    if (e->per_cu) per_cu = e->per_cu;
    t_want = e->want; t_maxth = e->maxth; t_tw = e->tw;
  }
  if (dl3p_dw_force_tw) t_tw = dl3p_dw_force_tw;
  if (dl3p_dw_force_per_cu) per_cu = dl3p_dw_force_per_cu;
  if (dl3p_dw_force_want) t_want = dl3p_dw_force_want;
  if (dl3p_dw_force_maxth) t_maxth = dl3p_dw_force_maxth;
  // 5x5: window kernels with LDS weights in the forward / data-gradient role (strips of 2 at stride 1, 1 at stride 2);
  // the weight gradient keeps the per-pixel gather (25 tap accumulators + a 5-row window do not fit)
  if (p.ks5 && !window5) kind = 0;
  else if (p.stride == 1 && ceil_div(p.Wo, p.rate) >= 4) kind = 1;
  else if (p.stride == 2 && p.rate == 1 && p.Wo >= 4) kind = 2;
  if (kind == 0 && p.ks == 3 && p.stride == 1 && p.pad_t == p.rate && p.pad_l == p.rate && p.Ho == p.H && p.Wo == p.W &&
      2 * p.rate >= p.H && 2 * p.rate >= p.W && p.rate < p.H && p.rate < p.W &&
      (long long)p.N * p.H * p.W * (p.ldx > p.ldy ? p.ldx : p.ldy) < (1LL << 31)) { kind = 3; p.lat = 2; }
  // one step down: 3 * rate covers the map (rate 12 on 33 x 33, rate 36 on 97 x 97) -> 3 x 3 pixels per class
  static const int lat3 = getenv("DL3P_DW_LAT3") ? atoi(getenv("DL3P_DW_LAT3")) : 1;
  if (kind == 0 && lat3 && p.ks == 3 && p.stride == 1 && p.pad_t == p.rate && p.pad_l == p.rate && p.Ho == p.H && p.Wo == p.W &&
      3 * p.rate >= p.H && 3 * p.rate >= p.W && p.rate < p.H && p.rate < p.W &&
      (long long)p.N * p.H * p.W * (p.ldx > p.ldy ? p.ldx : p.ldy) < (1LL << 31)) { kind = 3; p.lat = 3; }
  if (kind == 3) {
    p.spr = p.rate; p.th = 1; p.nbands = p.rate;
    p.total = (long long)p.N * p.rate * p.rate;      // one work item per residue class (py, px)
  } else if (kind == 0) {
    p.spr = p.Wo; p.th = 1; p.nbands = p.Ho;
    p.total = (long long)p.N * p.Ho * p.Wo;
  } else {
    const int r = p.rate;
    const int uw = ceil_div(p.Wo, r), uh = ceil_div(p.Ho, r);      // sub-lattice size
    // 3x3 stride 1: strips of 4 outputs (6-column window, 184-240 VGPRs: two waves per SIMD) or of 2 (160-192 VGPRs, three
    // waves per SIMD for the forward and the weight gradient, a third more window loads per output) -- the tuner's call
    const int TW = p.ks5 ? (kind == 1 ? (tw5 ? tw5 : dw5_tw(uw)) : 1) : (kind == 1 ? (t_tw == 2 ? 2 : 4) : 2);
    p.tw = TW;
    p.spr = ceil_div(uw, TW);
    // bands of equal height (+-1 row) instead of full ones and a remainder: 33 rows as 11+11+11, not 16+16+1
    static const int balance = getenv("DL3P_DW_BALANCE") ? atoi(getenv("DL3P_DW_BALANCE")) : 2;
    if (balance == 2) {
      // the fewest bands (tallest, least halo re-reading) that still give every CU its workgroup-iterations
      static const int want_env = getenv("DL3P_DW_WANT") ? atoi(getenv("DL3P_DW_WANT")) : DL3P_NUM_CUS * 3 / 2;
      static const int maxth_env = getenv("DL3P_DW_MAXTH") ? atoi(getenv("DL3P_DW_MAXTH")) : 16;
      const int want = t_want ? t_want : want_env, maxth = t_maxth ? t_maxth : maxth_env;
      const long long per_band = (long long)p.N * r * r * p.spr;
      int nb = ceil_div(uh, maxth);
      while (nb < uh && (per_band * nb / p.px) * p.nslab < want) ++nb;
      p.nbands = nb;
      p.th = ceil_div(uh, nb);
      p.uh = uh;
    } else {
      p.th = pick_band((long long)p.N * r * r * p.spr, uh, p.px, p.nslab);
      if (p.th > uh) p.th = uh;
      p.nbands = ceil_div(uh, p.th);
      p.uh = balance ? uh : p.nbands * p.th;
    }
    p.total = (long long)p.N * r * r * p.nbands * p.spr;
  }
  static const int lat2_per_cu = getenv("DL3P_LAT2_PER_CU") ? atoi(getenv("DL3P_LAT2_PER_CU")) : DL3P_LAT2_PER_CU;
  static const int lat3_per_cu = getenv("DL3P_LAT3_PER_CU") ? atoi(getenv("DL3P_LAT3_PER_CU")) : 3;
  p.nbx = pick_nbx(p.total, p.px, p.nslab, kind == 3 ? (p.lat == 3 ? lat3_per_cu : lat2_per_cu) : per_cu);
  return kind;
}

template <int KS, int PRO>
static void launch_fwd_pro(const DwParams& p, int kind, dim3 grid, hipStream_t st) {
  dim3 block(256);
  if (KS == 5) {
    // 5x5: narrower strips (window = 5 rows) and the weights in LDS (25 x c4s float4)
    const size_t lds = (size_t)25 * p.c4s * sizeof(float4);
    const int rows_tw = dw5_rows_tw(ceil_div(p.Wo, p.rate));
    if (kind == 1 && rows_tw == 4) dl3p_launch(dw5_rows<4, PRO>, grid, block, lds, st, p);
    else if (kind == 1 && rows_tw == 2) dl3p_launch(dw5_rows<2, PRO>, grid, block, lds, st, p);
    else if (kind == 1) dl3p_launch(dw_fwd_seg<5, 2, 1, PRO>, grid, block, lds, st, p);
    else if (kind == 2) dl3p_launch(dw_fwd_seg<5, 1, 2, PRO>, grid, block, lds, st, p);
    else dl3p_launch(dw_fwd_gather<5, PRO>, grid, block, 0, st, p);
    return;
  }
  if (p.up_x) {          // (dl3p_dw_upsampled_input: checked by the caller -- 3x3, stride 1, rate 1, window kernel, BatchNorm + activation prologue)
    if constexpr (PRO == 2) {
      if (p.tw == 2) dl3p_launch(dw_fwd_seg<3, 2, 1, 2, false, true>, grid, block, 0, st, p);
      else dl3p_launch(dw_fwd_seg<3, 4, 1, 2, false, true>, grid, block, 0, st, p);
    }
    return;
  }
  if (kind == 1 && p.tw == 2) dl3p_launch(dw_fwd_seg<3, 2, 1, PRO>, grid, block, 0, st, p);
  else if (kind == 1) dl3p_launch(dw_fwd_seg<3, 4, 1, PRO>, grid, block, 0, st, p);
  else if (kind == 2) dl3p_launch(dw_fwd_seg<3, 2, 2, PRO>, grid, block, 0, st, p);
  else if (kind == 3 && p.lat == 3) dl3p_launch(dw_fwd_lattice3<PRO>, grid, block, 0, st, p);
  else if (kind == 3) dl3p_launch(dw_fwd_lattice2<PRO>, grid, block, 0, st, p);
  else dl3p_launch(dw_fwd_gather<3, PRO>, grid, block, 0, st, p);
}

// lanes + work decomposition of a forward-role launch (also what dl3p_dwconv2d_fwd reports as partial rows)
static int plan_forward(DwParams& p, int KS, int tw5) {
  p.ks = KS;
  p.ks5 = KS == 5;
  static const int dwf_per_cu = getenv("DL3P_DWF_PER_CU") ? atoi(getenv("DL3P_DWF_PER_CU")) : 8;
  if (KS == 5) {          // keep the LDS weight tile small: at most 64 channel lanes per workgroup
    const int c4 = p.C / 4;
    int best = 1;
    for (int d = 1; d <= c4 && d <= 64; ++d)
      if (c4 % d == 0 && (256 / d) * d >= (256 / best) * best) best = d;
    p.c4s = best; p.px = 256 / best; p.nslab = c4 / best;
  }
  return fwd_plan(p, dwf_per_cu, true, tw5);
}

template <int KS>
static void launch_fwd(const DwParams& p0, hipStream_t st) {
  DwParams p = p0;
  const int kind = plan_forward(p, KS, 0);
  // streaming stores for forward outputs (bit 0 window kernels, bit 1 gather): -0.09 ms per step, and the rate-18
  // lattice kernel (always streaming) keeps its input in L2: 9.6 -> 8.8 us in-step
  static const int nt_mask = getenv("DL3P_DW_NT") ? atoi(getenv("DL3P_DW_NT")) : 3;
  p.nt = (p.flip == 0 && !p.accumulate) ? ((kind == 0 ? (nt_mask >> 1) : nt_mask) & 1) : 0;
  // counted-wait rows (dw_fwd_seg FAST): measured on MI355X they pay where the row's stores are slow to retire -- the STREAMING stores
  // of forward launches on tensors far beyond the caches (decoder_conv0 / conv1_depthwise at batch 16: 142.9 -> 137.5, 127.4 -> 108.0 us)
  // -- and lose 10-15 % on cache-sized tensors and on every data gradient (plain stores retire at L2 anyway).  DL3P_DW_FAST_ROWS:
  // 0 never, 1 that rule (default), 2 every launch that can
  static const int fast_rows = getenv("DL3P_DW_FAST_ROWS") ? atoi(getenv("DL3P_DW_FAST_ROWS")) : 1;
  const long long out_bytes = (long long)p.N * p.Ho * p.Wo * p.C * 4;
  p.fast_rows = fast_rows >= 2 || (fast_rows == 1 && p.nt && out_bytes >= (200ll << 20));
  dim3 grid(p.nbx * p.nslab);
  const int pro = (p.act != DL3P_ACT_NONE) ? 2 : (p.scale ? 1 : 0);
  if (pro == 2) launch_fwd_pro<KS, 2>(p, kind, grid, st);
  else if (pro == 1) launch_fwd_pro<KS, 1>(p, kind, grid, st);
  else launch_fwd_pro<KS, 0>(p, kind, grid, st);
}

// ---- an input whose first channels are a bilinear upsampling formed on the fly (DwParams::up_x)
static thread_local DwParams t_dw_up = {};
extern "C" int dl3p_dw_upsampled_input(const float* up_x, int up_ld, int up_h, int up_w, int up_C, void* stream) {
  (void)stream;          // (every entry of a launch list ends in the stream)
  DL3P_CHECK_ARG(up_x && aligned16(up_x) && up_ld % 4 == 0 && up_C > 0 && up_C % 4 == 0 && up_ld >= up_C && up_h > 0 && up_w > 0,
                 "dl3p_dw_upsampled_input: bad arguments");
  t_dw_up.up_x = up_x; t_dw_up.up_ld = up_ld; t_dw_up.up_h = up_h; t_dw_up.up_w = up_w; t_dw_up.up_C = up_C;
  return DL3P_OK;
}
static int plan_forward(DwParams& p, int KS, int tw5);
static int dwconv2d_bwd_weight_impl(const float* x, int ldx, const float* in_scale, const float* in_shift, int in_act, const float* dy,
                                    int lddy, float* gw, float* workspace, size_t workspace_bytes, int N, int H, int W, int C, int k,
                                    int stride, int rate, int pad_t, int pad_l, int Ho, int Wo, int* rows_out, void* stream,
                                    const DwParams* fold, int* kind_out);
// role 0: dl3p_dwconv2d_fwd, 1: dl3p_dwconv2d_bwd_weight_slabs[_bn] -- is the launch one of the 3x3 stride-1 window kernels that carry
// the UP form (the caller's prologue must be a BatchNorm affine + activation)?
extern "C" int dl3p_dw_upsampled_input_supported(int role, int N, int H, int W, int C, int up_C, int k, int stride, int rate, int pad_t,
                                                 int pad_l, int Ho, int Wo) {
  if (k != 3 || stride != 1 || rate != 1 || C <= 0 || C % 4 || up_C <= 0 || up_C % 4 || up_C > C || N <= 0 || H <= 0 || W <= 0) return 0;
  DwParams p = {};
  p.N = N; p.H = H; p.W = W; p.C = C; p.Ho = Ho; p.Wo = Wo; p.stride = stride; p.rate = rate; p.pad_t = pad_t; p.pad_l = pad_l;
  if (role == 0) {
    pick_lanes(C, &p.c4s, &p.px, &p.nslab);
    return plan_forward(p, 3, 0) == 1;
  }
  int kind = -1;
  alignas(16) static float dummy[4];
  const int rc = dwconv2d_bwd_weight_impl(dummy, C, nullptr, nullptr, DL3P_ACT_NONE, dummy, C, dummy, dummy, (size_t)-1, N, H, W, C, k,
                                          stride, rate, pad_t, pad_l, Ho, Wo, nullptr, nullptr, nullptr, &kind);
  return rc == DL3P_OK && kind == 1;
}
// (consumes the armed description: it applies to ONE call)
static bool dw_take_up(DwParams& p) {
  if (!t_dw_up.up_x) return false;
  p.up_x = t_dw_up.up_x; p.up_ld = t_dw_up.up_ld; p.up_h = t_dw_up.up_h; p.up_w = t_dw_up.up_w; p.up_C = t_dw_up.up_C;
  t_dw_up.up_x = nullptr;
  return true;
}

extern "C" int dl3p_dwconv2d_fwd(const float* x, int ldx, const float* in_scale, const float* in_shift, int in_act,
                                 const float* w, float* y, int ldy, float* stat_partials, int* rows_out,
                                 int N, int H, int W, int C, int k, int stride, int rate, int pad_t, int pad_l,
                                 int Ho, int Wo, void* stream) {
  int rc = check_dw_common("dl3p_dwconv2d_fwd", x, ldx, C, k);
  if (rc) return rc;
  DL3P_CHECK_ARG(x && w && y, "dl3p_dwconv2d_fwd: null pointer");
  DL3P_CHECK_ARG(ldy % 4 == 0 && ldy >= C && aligned16(y) && aligned16(w), "dl3p_dwconv2d_fwd: bad y/w layout");
  DL3P_CHECK_ARG(N > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0 && stride >= 1 && rate >= 1, "dl3p_dwconv2d_fwd: bad dims");
  DwParams p = {};
  p.x = x; p.ldx = ldx; p.scale = in_scale; p.shift = in_shift; p.act = in_act; p.w = w;
  p.y = y; p.ldy = ldy; p.partials = stat_partials;
  p.N = N; p.H = H; p.W = W; p.C = C; p.Ho = Ho; p.Wo = Wo; p.stride = stride; p.rate = rate;
  p.pad_t = pad_t; p.pad_l = pad_l;
  pick_lanes(C, &p.c4s, &p.px, &p.nslab);
  const bool up = dw_take_up(p);
  {
    DwParams q = p;
    const int kind = plan_forward(q, k, 0);
    if (rows_out) *rows_out = q.nbx;
    DL3P_CHECK_ARG(!up || (k == 3 && stride == 1 && rate == 1 && kind == 1 && in_scale && in_act != DL3P_ACT_NONE && p.up_C <= C),
                   "dl3p_dwconv2d_fwd: the upsampled-input form serves 3x3 stride-1 window launches behind a BatchNorm + activation (dl3p_dw_upsampled_input_supported)");
  }
  hipStream_t st = (hipStream_t)stream;
  if (k == 3) launch_fwd<3>(p, st); else launch_fwd<5>(p, st);
  DL3P_CHECK_LAUNCH("dl3p_dwconv2d_fwd");
  return DL3P_OK;
}

extern "C" int dl3p_dwconv2d_bwd_data(const float* dy, int lddy, const float* w, float* gx, int ldgx, int accumulate,
                                      int N, int H, int W, int C, int k, int stride, int rate, int pad_t, int pad_l,
                                      int Ho, int Wo, void* stream) {
  int rc = check_dw_common("dl3p_dwconv2d_bwd_data", dy, lddy, C, k);
  if (rc) return rc;
  DL3P_CHECK_ARG(dy && w && gx, "dl3p_dwconv2d_bwd_data: null pointer");
  DL3P_CHECK_ARG(ldgx % 4 == 0 && ldgx >= C && aligned16(gx) && aligned16(w), "dl3p_dwconv2d_bwd_data: bad gx/w layout");
  hipStream_t st = (hipStream_t)stream;
  DwParams p = {};
  p.w = w; p.C = C; p.N = N; p.accumulate = accumulate;
  pick_lanes(C, &p.c4s, &p.px, &p.nslab);
  if (stride == 1) {
    // correlation with the flipped kernel: a forward conv from (Ho,Wo) to (H,W) with
    // pad' = rate*(k-1) - pad
    p.x = dy; p.ldx = lddy; p.y = gx; p.ldy = ldgx; p.flip = 1;
    p.H = Ho; p.W = Wo; p.Ho = H; p.Wo = W; p.stride = 1; p.rate = rate;
    p.pad_t = rate * (k - 1) - pad_t; p.pad_l = rate * (k - 1) - pad_l;
    p.act = DL3P_ACT_NONE;
    if (k == 3) launch_fwd<3>(p, st); else launch_fwd<5>(p, st);
  } else {
    p.dy = dy; p.lddy = lddy; p.y = gx; p.ldy = ldgx;
    p.H = H; p.W = W; p.Ho = Ho; p.Wo = Wo; p.stride = stride; p.rate = rate; p.pad_t = pad_t; p.pad_l = pad_l;
    if (stride == 2 && rate == 1) {
      p.th = ((H - 1 + pad_t) >> 1) + 1;     // quads per column
      p.spr = ((W - 1 + pad_l) >> 1) + 1;    // quads per row
      p.total = (long long)N * p.th * p.spr;
      p.nbx = pick_nbx(p.total, p.px, p.nslab);
      dim3 grid(p.nbx * p.nslab), block(256);
      if (k == 3) hipLaunchKernelGGL((dw_bwd_data_s2<3>), grid, block, 0, st, p);
      else hipLaunchKernelGGL((dw_bwd_data_s2<5>), grid, block, 0, st, p);
    } else {
      p.total = (long long)N * H * W;
      p.nbx = pick_nbx(p.total, p.px, p.nslab);
      dim3 grid(p.nbx * p.nslab), block(256);
      if (k == 3) hipLaunchKernelGGL((dw_bwd_data_strided<3>), grid, block, 0, st, p);
      else hipLaunchKernelGGL((dw_bwd_data_strided<5>), grid, block, 0, st, p);
    }
  }
  DL3P_CHECK_LAUNCH("dl3p_dwconv2d_bwd_data");
  return DL3P_OK;
}

// bwd-data of a layer whose input is act(BN(z)) + the BN-backward partial sums of that BN.  The stride-1 window
// kernel and the stride-2 quad kernel fold the sums into their store loop (one extra read of z at the output pixel);
// other decompositions (gather, residue-class kernel, 5x5) run the plain data gradient followed by the reduce pass: the
// 5x5 rolling-row kernel is VALU-bound and the sums cost it more (65x65x120: +18 us, 4-column strips spill) than the
// streaming reduce pass does (12 us).
extern "C" int dl3p_bn_bwd_reduce(const float* g, int ldg, const float* z, int ldz, const float* scale, const float* shift,
                                  int act, const float* save_mean, const float* save_invstd, float* partials,
                                  int* rows_out, int M, int C, void* stream);
extern "C" int dl3p_dwconv2d_bwd_data_bn(const float* dy, int lddy, const float* w, float* gx, int ldgx, int accumulate,
                                         int N, int H, int W, int C, int k, int stride, int rate, int pad_t, int pad_l,
                                         int Ho, int Wo, const float* z, int ldz, const float* scale, const float* shift,
                                         int act, const float* save_mean, const float* save_invstd, float* partials,
                                         int* rows_out, void* stream) {
  int rc = check_dw_common("dl3p_dwconv2d_bwd_data_bn", dy, lddy, C, k);
  if (rc) return rc;
  DL3P_CHECK_ARG(dy && w && gx && z && scale && shift && save_mean && save_invstd && partials && rows_out,
                 "dl3p_dwconv2d_bwd_data_bn: null pointer");
  DL3P_CHECK_ARG(ldgx % 4 == 0 && ldgx >= C && ldz % 4 == 0 && ldz >= C && aligned16(gx) && aligned16(w) && aligned16(z),
                 "dl3p_dwconv2d_bwd_data_bn: bad gx/z/w layout");
  hipStream_t st = (hipStream_t)stream;
  DwParams p = {};
  p.w = w; p.C = C; p.N = N; p.accumulate = accumulate; p.partials = partials;
  p.bb_z = z; p.bb_ldz = ldz; p.bb_scale = scale; p.bb_shift = shift; p.bb_mean = save_mean; p.bb_invstd = save_invstd;
  p.bb_act = act;
  pick_lanes(C, &p.c4s, &p.px, &p.nslab);
  bool fused = false;
  if (stride == 1 && k == 3) {
    p.x = dy; p.ldx = lddy; p.y = gx; p.ldy = ldgx; p.flip = 1;
    p.H = Ho; p.W = Wo; p.Ho = H; p.Wo = W; p.stride = 1; p.rate = rate;
    p.pad_t = rate * (k - 1) - pad_t; p.pad_l = rate * (k - 1) - pad_l;
    p.act = DL3P_ACT_NONE;
    const int kind = plan_forward(p, k, 0);
    if (kind == 1) {
      if (p.tw == 2) dl3p_launch(dw_fwd_seg<3, 2, 1, 0, true>, dim3(p.nbx * p.nslab), dim3(256), 0, st, p);
      else dl3p_launch(dw_fwd_seg<3, 4, 1, 0, true>, dim3(p.nbx * p.nslab), dim3(256), 0, st, p);
      fused = true;
    }
  } else if (stride == 2 && rate == 1 && k == 3) {
    p.dy = dy; p.lddy = lddy; p.y = gx; p.ldy = ldgx;
    p.H = H; p.W = W; p.Ho = Ho; p.Wo = Wo; p.stride = stride; p.rate = rate; p.pad_t = pad_t; p.pad_l = pad_l;
    p.th = ((H - 1 + pad_t) >> 1) + 1;
    p.spr = ((W - 1 + pad_l) >> 1) + 1;
    p.total = (long long)N * p.th * p.spr;
    p.nbx = pick_nbx(p.total, p.px, p.nslab);
    hipLaunchKernelGGL((dw_bwd_data_s2<3, true>), dim3(p.nbx * p.nslab), dim3(256), 0, st, p);
    fused = true;
  }
  if (fused) {
    *rows_out = p.nbx;
    DL3P_CHECK_LAUNCH("dl3p_dwconv2d_bwd_data_bn");
    return DL3P_OK;
  }
  rc = dl3p_dwconv2d_bwd_data(dy, lddy, w, gx, ldgx, accumulate, N, H, W, C, k, stride, rate, pad_t, pad_l, Ho, Wo, stream);
  if (rc) return rc;
  return dl3p_bn_bwd_reduce(gx, ldgx, z, ldz, scale, shift, act, save_mean, save_invstd, partials, rows_out, N * H * W, C,
                            stream);
}

template <int PRO>
static void launch_bwdw_bna(const DwParams& p, int kind, dim3 grid, hipStream_t st) {
  dim3 block(256);
  if (p.up_x) {
    if constexpr (PRO == 2) {
      if (p.tw == 2) hipLaunchKernelGGL((dw_bwd_weight_seg<3, 2, 1, 2, true, true>), grid, block, 0, st, p);
      else hipLaunchKernelGGL((dw_bwd_weight_seg<3, 4, 1, 2, true, true>), grid, block, 0, st, p);
    }
    return;
  }
  if (kind == 1 && p.tw == 2) hipLaunchKernelGGL((dw_bwd_weight_seg<3, 2, 1, PRO, true>), grid, block, 0, st, p);
  else if (kind == 1) hipLaunchKernelGGL((dw_bwd_weight_seg<3, 4, 1, PRO, true>), grid, block, 0, st, p);
  else hipLaunchKernelGGL((dw_bwd_weight_seg<3, 2, 2, PRO, true>), grid, block, 0, st, p);
}

template <int KS, int PRO>
static void launch_bwdw(const DwParams& p, int kind, dim3 grid, hipStream_t st) {
  dim3 block(256);
  if (KS == 5) {     // 25 tap accumulators + a 5-row window do not fit the register file: per-pixel gather only
    hipLaunchKernelGGL((dw_bwd_weight<5, PRO>), grid, block, 0, st, p);
    return;
  }
  if (p.up_x) {
    if constexpr (PRO == 2) {
      if (p.tw == 2) hipLaunchKernelGGL((dw_bwd_weight_seg<3, 2, 1, 2, false, true>), grid, block, 0, st, p);
      else hipLaunchKernelGGL((dw_bwd_weight_seg<3, 4, 1, 2, false, true>), grid, block, 0, st, p);
    }
    return;
  }
  if (kind == 1 && p.tw == 2) hipLaunchKernelGGL((dw_bwd_weight_seg<3, 2, 1, PRO>), grid, block, 0, st, p);
  else if (kind == 1) hipLaunchKernelGGL((dw_bwd_weight_seg<3, 4, 1, PRO>), grid, block, 0, st, p);
  else if (kind == 2) hipLaunchKernelGGL((dw_bwd_weight_seg<3, 2, 2, PRO>), grid, block, 0, st, p);
  else hipLaunchKernelGGL((dw_bwd_weight<3, PRO>), grid, block, 0, st, p);   // kind 0 (ks == 0 there: no kind 3)
}

extern "C" size_t dl3p_dwconv2d_bwd_weight_workspace(int N, int Ho, int Wo, int C, int k) {
  if (C <= 0 || C % 4) return 0;
  (void)N; (void)Ho; (void)Wo;
  return (size_t)DL3P_MAX_STAT_ROWS * k * k * C * sizeof(float);   // one partial row per workgroup, at most
}

// rows_out != NULL: leave the partial rows in the workspace for dl3p_reduce_rows_batched (gw unused) and report how many
static int dwconv2d_bwd_weight_impl(const float* x, int ldx, const float* in_scale, const float* in_shift,
                                    int in_act, const float* dy, int lddy, float* gw, float* workspace,
                                    size_t workspace_bytes, int N, int H, int W, int C, int k, int stride,
                                    int rate, int pad_t, int pad_l, int Ho, int Wo, int* rows_out, void* stream,
                                    const DwParams* fold, int* kind_out) {
  int rc = check_dw_common("dl3p_dwconv2d_bwd_weight", x, ldx, C, k);
  if (rc) return rc;
  DL3P_CHECK_ARG(x && dy && (gw || rows_out) && workspace, "dl3p_dwconv2d_bwd_weight: null pointer");
  DL3P_CHECK_ARG(lddy % 4 == 0 && lddy >= C && aligned16(dy) && aligned16(workspace) && (rows_out || aligned16(gw)),
                 "dl3p_dwconv2d_bwd_weight: bad dy/workspace layout");
  const size_t need = dl3p_dwconv2d_bwd_weight_workspace(N, Ho, Wo, C, k);
  if (workspace_bytes < need) {
    dl3p_set_error("dl3p_dwconv2d_bwd_weight: workspace %zu < %zu bytes", workspace_bytes, need);
    return DL3P_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  DwParams p = {};
  p.x = x; p.ldx = ldx; p.scale = in_scale; p.shift = in_shift; p.act = in_act;
  p.dy = dy; p.lddy = lddy; p.partials = workspace;
  p.N = N; p.H = H; p.W = W; p.C = C; p.Ho = Ho; p.Wo = Wo; p.stride = stride; p.rate = rate;
  p.pad_t = pad_t; p.pad_l = pad_l;
  pick_lanes(C, &p.c4s, &p.px, &p.nslab);
  p.ks5 = k == 5;
  static const int dww_per_cu = getenv("DL3P_DWW_PER_CU") ? atoi(getenv("DL3P_DWW_PER_CU")) : 2;   // fewer slabs: the slab reduce costs as much as the kernel at 8
  // 5x5 stride 1: rolling-row kernel with strips of DL3P_DW5_WROWS (1 | 2) columns, 0 = per-pixel gather
  static const int wrows = getenv("DL3P_DW5_WROWS") ? atoi(getenv("DL3P_DW5_WROWS")) : 2;
  const bool rows5 = k == 5 && stride == 1 && wrows > 0;
  const int kind = fwd_plan(p, dww_per_cu, rows5, wrows);
  if (kind_out) { *kind_out = kind; return DL3P_OK; }      // plan query (dl3p_dwconv2d_bwd_weight_bn_supported)
  dim3 grid(p.nbx * p.nslab);
  const int pro = (in_act != DL3P_ACT_NONE) ? 2 : (in_scale ? 1 : 0);
  if (dw_take_up(p))
    DL3P_CHECK_ARG(k == 3 && stride == 1 && rate == 1 && kind == 1 && pro == 2 && p.up_C <= C,
                   "dl3p_dwconv2d_bwd_weight: the upsampled-input form serves 3x3 stride-1 window launches behind a BatchNorm + activation");
  if (fold) {
    DL3P_CHECK_ARG(k == 3 && (kind == 1 || kind == 2), "dl3p_dwconv2d_bwd_weight_slabs_bn: geometry not served by the window kernels");
    p.fa_z = fold->fa_z; p.fa_ldz = fold->fa_ldz; p.fa_scale = fold->fa_scale; p.fa_shift = fold->fa_shift;
    p.fa_mean = fold->fa_mean; p.fa_invstd = fold->fa_invstd; p.fa_coef = fold->fa_coef; p.fa_act = fold->fa_act;
    p.fa_dz = fold->fa_dz; p.fa_lddz = fold->fa_lddz;
    if (pro == 2) launch_bwdw_bna<2>(p, kind, grid, st); else if (pro == 1) launch_bwdw_bna<1>(p, kind, grid, st); else launch_bwdw_bna<0>(p, kind, grid, st);
  } else if (rows5 && kind == 1) {
    dim3 block(256);
    if (wrows == 1) {
      if (pro == 2) hipLaunchKernelGGL((dw5_wgrad_rows<1, 2>), grid, block, 0, st, p);
      else if (pro == 1) hipLaunchKernelGGL((dw5_wgrad_rows<1, 1>), grid, block, 0, st, p);
      else hipLaunchKernelGGL((dw5_wgrad_rows<1, 0>), grid, block, 0, st, p);
    } else {
      if (pro == 2) hipLaunchKernelGGL((dw5_wgrad_rows<2, 2>), grid, block, 0, st, p);
      else if (pro == 1) hipLaunchKernelGGL((dw5_wgrad_rows<2, 1>), grid, block, 0, st, p);
      else hipLaunchKernelGGL((dw5_wgrad_rows<2, 0>), grid, block, 0, st, p);
    }
  } else if (k == 3) {
    if (pro == 2) launch_bwdw<3, 2>(p, kind, grid, st); else if (pro == 1) launch_bwdw<3, 1>(p, kind, grid, st); else launch_bwdw<3, 0>(p, kind, grid, st);
  } else {
    if (pro == 2) launch_bwdw<5, 2>(p, kind, grid, st); else if (pro == 1) launch_bwdw<5, 1>(p, kind, grid, st); else launch_bwdw<5, 0>(p, kind, grid, st);
  }
  DL3P_CHECK_LAUNCH("dl3p_dwconv2d_bwd_weight");
  if (rows_out) { *rows_out = p.nbx; return DL3P_OK; }
  return dl3p_reduce_rows_impl(workspace, p.nbx, (size_t)k * k * C, gw, 0, st);
}

extern "C" int dl3p_dwconv2d_bwd_weight(const float* x, int ldx, const float* in_scale, const float* in_shift,
                                        int in_act, const float* dy, int lddy, float* gw, float* workspace,
                                        size_t workspace_bytes, int N, int H, int W, int C, int k, int stride,
                                        int rate, int pad_t, int pad_l, int Ho, int Wo, void* stream) {
  return dwconv2d_bwd_weight_impl(x, ldx, in_scale, in_shift, in_act, dy, lddy, gw, workspace, workspace_bytes, N, H, W, C, k,
                                  stride, rate, pad_t, pad_l, Ho, Wo, nullptr, stream, nullptr, nullptr);
}

// dl3p_dwconv2d_bwd_weight_slabs with the BatchNorm-backward apply of the conv's own output folded in (the depthwise
// weight gradient visits every output pixel once, so dz is formed there from g and z and written for the data gradient:
// the apply pass over (g, z, dz) and its launch disappear).  3x3 window kernels only (stride 1 at any rate, stride 2).
extern "C" int dl3p_dwconv2d_bwd_weight_bn_supported(int N, int H, int W, int C, int k, int stride, int rate, int pad_t,
                                                     int pad_l, int Ho, int Wo) {
  if (k != 3 || C <= 0 || C % 4 || N <= 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0) return 0;
  int kind = -1;
  alignas(16) static float dummy[4];
  const int rc = dwconv2d_bwd_weight_impl(dummy, C, nullptr, nullptr, DL3P_ACT_NONE, dummy, C, dummy, dummy, (size_t)-1, N, H, W, C,
                                          k, stride, rate, pad_t, pad_l, Ho, Wo, nullptr, nullptr, nullptr, &kind);
  return rc == DL3P_OK && (kind == 1 || kind == 2);
}

extern "C" int dl3p_dwconv2d_bwd_weight_slabs_bn(const float* x, int ldx, const float* in_scale, const float* in_shift,
                                                 int in_act, const float* g, int ldg, const float* z, int ldz,
                                                 const float* bn_scale, const float* bn_shift, int bn_act,
                                                 const float* save_mean, const float* save_invstd, const float* coef,
                                                 float* dz, int lddz, float* workspace, size_t workspace_bytes,
                                                 int* rows_out, int N, int H, int W, int C, int k, int stride, int rate,
                                                 int pad_t, int pad_l, int Ho, int Wo, void* stream) {
  const char* fn = "dl3p_dwconv2d_bwd_weight_slabs_bn";
  DL3P_CHECK_ARG(rows_out && z && save_mean && save_invstd && coef && ldz % 4 == 0 && ldz >= C && aligned16(z) && dz != g &&
                     (!dz || (lddz % 4 == 0 && lddz >= C && aligned16(dz))), "%s: bad arguments", fn);
  DwParams f = {};
  f.fa_z = z; f.fa_ldz = ldz; f.fa_scale = bn_scale; f.fa_shift = bn_shift; f.fa_mean = save_mean; f.fa_invstd = save_invstd;
  f.fa_coef = coef; f.fa_act = bn_act; f.fa_dz = dz; f.fa_lddz = lddz;
  return dwconv2d_bwd_weight_impl(x, ldx, in_scale, in_shift, in_act, g, ldg, nullptr, workspace, workspace_bytes, N, H, W, C, k,
                                  stride, rate, pad_t, pad_l, Ho, Wo, rows_out, stream, &f, nullptr);
}

extern "C" int dl3p_dwconv2d_bwd_weight_slabs(const float* x, int ldx, const float* in_scale, const float* in_shift,
                                              int in_act, const float* dy, int lddy, float* workspace,
                                              size_t workspace_bytes, int* rows_out, int N, int H, int W, int C, int k,
                                              int stride, int rate, int pad_t, int pad_l, int Ho, int Wo, void* stream) {
  DL3P_CHECK_ARG(rows_out != nullptr, "dl3p_dwconv2d_bwd_weight_slabs: rows_out is required");
  return dwconv2d_bwd_weight_impl(x, ldx, in_scale, in_shift, in_act, dy, lddy, nullptr, workspace, workspace_bytes, N, H, W, C,
                                  k, stride, rate, pad_t, pad_l, Ho, Wo, rows_out, stream, nullptr, nullptr);
}

#include "dw_bf16_window.h"

// ------------------------------------------------------------------------------ mixed-precision entry of the window kernels
// Called by dw_bf16.hip (dl3p_dwconv2d_{fwd,bwd_data,bwd_weight}_bf16) before its own strip / gather kernels: the sliding-window
// decomposition above -- every input row loaded and activated ONCE per strip, bands of rows for parallelism at batch 1, the
// XCD-aware split -- with bf16 tensors (T = bf16 instantiations).  The strip kernels of dw_bf16_strip.h re-load and re-activate the
// k input rows for every output row; on BASELINE configs[4] (MobileNetV3-Large 1024 x 2048, batch 1) they ran at 0.4-2.4 TB/s.
// Returns 1 = launched, 0 = geometry not served by a window kernel (caller falls back), < 0 = error code.
// role 0: forward x (N,H,W,C) -> y (N,Ho,Wo,C) [+ statistics]; role 1: data gradient of a STRIDE-1 conv, x = dy (N,Ho,Wo,C),
// y = gx (N,H,W,C), the kernel mirror read back to front.
int dl3p_dw_window_bf16(int role, const void* x, int ldx, const float* in_scale, const float* in_shift, int in_act, const void* w,
                        void* y, int ldy, float* partials, int* rows_out, int accumulate, int N, int H, int W, int C, int k,
                        int stride, int rate, int pad_t, int pad_l, int Ho, int Wo, hipStream_t st) {
  static const int on = getenv("DL3P_BF16_DW_WINDOW") ? atoi(getenv("DL3P_BF16_DW_WINDOW")) : 1;
  if (!on || C <= 0 || C % 4 || (k != 3 && k != 5) || ldx % 4 || ldy % 4 || ((uintptr_t)x & 7u) || ((uintptr_t)y & 7u) || ((uintptr_t)w & 7u))
    return 0;
  if (role == 1 && stride != 1) return 0;
  // 5x5: the 2-column window kernel needs all 256 registers (one wave per SIMD) and loses to the strip kernels (64x128x960 rate 2:
  // 44.7 against 38.6 us); stride 2 ties.  3x3 only.
  if (k != 3) return 0;
  DwParams p = {};
  p.bf16_io = 1;
  p.x = reinterpret_cast<const float*>(x); p.ldx = ldx; p.w = reinterpret_cast<const float*>(w);
  p.y = reinterpret_cast<float*>(y); p.ldy = ldy; p.C = C; p.N = N; p.accumulate = accumulate;
  if (role == 0) {
    p.scale = in_scale; p.shift = in_shift; p.act = in_act; p.partials = partials;
    p.H = H; p.W = W; p.Ho = Ho; p.Wo = Wo; p.stride = stride; p.rate = rate; p.pad_t = pad_t; p.pad_l = pad_l;
  } else {
    p.flip = 1; p.act = DL3P_ACT_NONE;
    p.H = Ho; p.W = Wo; p.Ho = H; p.Wo = W; p.stride = 1; p.rate = rate;
    p.pad_t = rate * (k - 1) - pad_t; p.pad_l = rate * (k - 1) - pad_l;
  }
  pick_lanes(C, &p.c4s, &p.px, &p.nslab);
  const int kind = plan_forward(p, k, 2);
  if (kind != 1 && kind != 2) return 0;
  if (rows_out) *rows_out = p.nbx;
  const dim3 grid(p.nbx * p.nslab), block(256);
  const int pro = (p.act != DL3P_ACT_NONE) ? 2 : (p.scale ? 1 : 0);
  const bool hs = p.act >= DL3P_ACT_HSWISH;
  const size_t lds5 = (size_t)25 * p.c4s * sizeof(float4);
  // (dl3p_launch: the forward depthwise launch can carry bench.py's HIP event pair, dl3p_probe_arm)
#define DL3P_DWB_F(PRO, HS)                                                                                              \
  do {                                                                                                                   \
    if (k == 5 && kind == 1) dl3p_launch(dwb_fwd_seg<5, 2, 1, PRO, HS>, grid, block, lds5, st, p);                       \
    else if (k == 5) dl3p_launch(dwb_fwd_seg<5, 1, 2, PRO, HS>, grid, block, lds5, st, p);                               \
    else if (kind == 1 && p.tw == 2) dl3p_launch(dwb_fwd_seg<3, 2, 1, PRO, HS>, grid, block, 0, st, p);                  \
    else if (kind == 1) dl3p_launch(dwb_fwd_seg<3, 4, 1, PRO, HS>, grid, block, 0, st, p);                               \
    else dl3p_launch(dwb_fwd_seg<3, 2, 2, PRO, HS>, grid, block, 0, st, p);                                              \
  } while (0)
  if (pro == 2 && hs) DL3P_DWB_F(2, true); else if (pro == 2) DL3P_DWB_F(2, false); else if (pro == 1) DL3P_DWB_F(1, false); else DL3P_DWB_F(0, false);
#undef DL3P_DWB_F
  return 1;
}

// weight gradient, 3x3: x (bf16, producer's BatchNorm + activation applied and rounded on load) x dy (bf16) -> fp32 partial rows
// [rows][9][C] in `workspace` (room for max_rows of them)
int dl3p_dw_window_wgrad_bf16(const void* x, int ldx, const float* in_scale, const float* in_shift, int in_act, const void* dy,
                              int lddy, float* workspace, int max_rows, int* rows_out, int N, int H, int W, int C, int k,
                              int stride, int rate, int pad_t, int pad_l, int Ho, int Wo, hipStream_t st) {
  static const int on = getenv("DL3P_BF16_DW_WINDOW") ? atoi(getenv("DL3P_BF16_DW_WINDOW")) : 1;
  if (!on || k != 3 || C <= 0 || C % 4 || ldx % 4 || lddy % 4 || ((uintptr_t)x & 7u) || ((uintptr_t)dy & 7u) || max_rows < DL3P_NUM_XCDS) return 0;
  // few pixels (Xception's 33 x 33 maps at batch 4: 4356): the strip kernel's finer work items fill the chip better -- 27.4 against
  // 34.2 us on 33 x 33 x 728; from 65 x 65 x 4 up the window kernel ties or wins (129 x 129 x 256 stride 2: 35.5 against 49.5)
  if ((long long)N * Ho * Wo < 16384) return 0;
  DwParams p = {};
  p.bf16_io = 1;
  p.x = reinterpret_cast<const float*>(x); p.ldx = ldx; p.scale = in_scale; p.shift = in_shift; p.act = in_act;
  p.dy = reinterpret_cast<const float*>(dy); p.lddy = lddy; p.partials = workspace;
  p.N = N; p.H = H; p.W = W; p.C = C; p.Ho = Ho; p.Wo = Wo; p.stride = stride; p.rate = rate; p.pad_t = pad_t; p.pad_l = pad_l;
  pick_lanes(C, &p.c4s, &p.px, &p.nslab);
  p.ks = 3;
  const int kind = fwd_plan(p, 2, false, 0);
  if (kind != 1 && kind != 2) return 0;
  if (p.nbx > max_rows) p.nbx = max_rows / DL3P_NUM_XCDS * DL3P_NUM_XCDS;      // the workspace's rows (a multiple of the XCD count stays one)
  const dim3 grid(p.nbx * p.nslab), block(256);
  const int pro = (in_act != DL3P_ACT_NONE) ? 2 : (in_scale ? 1 : 0);
  const bool hs = in_act >= DL3P_ACT_HSWISH;
#define DL3P_DWW_B(PRO, HS)                                                                                              \
  do {                                                                                                                   \
    if (kind == 1 && p.tw == 2) hipLaunchKernelGGL((dwb_bwd_weight_seg<3, 2, 1, PRO, HS>), grid, block, 0, st, p);        \
    else if (kind == 1) hipLaunchKernelGGL((dwb_bwd_weight_seg<3, 4, 1, PRO, HS>), grid, block, 0, st, p);                \
    else hipLaunchKernelGGL((dwb_bwd_weight_seg<3, 2, 2, PRO, HS>), grid, block, 0, st, p);                               \
  } while (0)
  if (pro == 2 && hs) DL3P_DWW_B(2, true); else if (pro == 2) DL3P_DWW_B(2, false); else if (pro == 1) DL3P_DWW_B(1, false); else DL3P_DWW_B(0, false);
#undef DL3P_DWW_B
  *rows_out = p.nbx;
  return 1;
}

// ------------------------------------------------------------------------------ plan query (include/dl3p.h)
// the decomposition the entry points above would launch for this conv, reported instead of launched
extern "C" int dl3p_dw_plan_query(int role, int N, int H, int W, int C, int k, int stride, int rate, int pad_t, int pad_l, int Ho,
                                  int Wo, int* out6) {
  DL3P_CHECK_ARG(out6 && role >= 0 && role <= 3 && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && (k == 3 || k == 5) &&
                     stride >= 1 && rate >= 1 && Ho > 0 && Wo > 0, "dl3p_dw_plan_query: bad arguments");
  for (int i = 0; i < 6; ++i) out6[i] = 0;
  alignas(16) static float dummy[4];
  DwParams p = {};
  p.C = C; p.N = N; p.w = dummy;
  pick_lanes(C, &p.c4s, &p.px, &p.nslab);
  int kind;
  if (role == 0) {
    p.x = dummy; p.ldx = C; p.y = dummy; p.ldy = C;
    p.H = H; p.W = W; p.Ho = Ho; p.Wo = Wo; p.stride = stride; p.rate = rate; p.pad_t = pad_t; p.pad_l = pad_l;
    kind = plan_forward(p, k, 0);
  } else if (role == 1 || role == 2) {
    if (role == 2) p.bb_z = dummy;
    if (stride == 1 && (role == 1 || k == 3)) {
      p.x = dummy; p.ldx = C; p.y = dummy; p.ldy = C; p.flip = 1;
      p.H = Ho; p.W = Wo; p.Ho = H; p.Wo = W; p.stride = 1; p.rate = rate;
      p.pad_t = rate * (k - 1) - pad_t; p.pad_l = rate * (k - 1) - pad_l;
      kind = plan_forward(p, k, 0);
    } else {
      out6[0] = 4;          // stride-2 quads / strided gather of the data gradient: no row bands, no table
      return DL3P_OK;
    }
  } else {
    p.x = dummy; p.ldx = C; p.dy = dummy; p.lddy = C;
    p.H = H; p.W = W; p.Ho = Ho; p.Wo = Wo; p.stride = stride; p.rate = rate; p.pad_t = pad_t; p.pad_l = pad_l;
    p.ks5 = k == 5;
    static const int dww_per_cu = getenv("DL3P_DWW_PER_CU") ? atoi(getenv("DL3P_DWW_PER_CU")) : 2;
    static const int wrows = getenv("DL3P_DW5_WROWS") ? atoi(getenv("DL3P_DW5_WROWS")) : 2;
    kind = fwd_plan(p, dww_per_cu, k == 5 && stride == 1 && wrows > 0, wrows);
  }
  out6[0] = kind; out6[1] = p.tw; out6[2] = p.th; out6[3] = p.nbands; out6[4] = p.nbx; out6[5] = dw_tuned_lookup(p) != nullptr;
  return DL3P_OK;
}
