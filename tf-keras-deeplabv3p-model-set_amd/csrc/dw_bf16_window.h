// Sliding-window depthwise kernels of the mixed-precision path (included at the end of dwconv.hip: same DwParams, same plans --
// strips x bands per residue class, XCD-aware split, one partial row per workgroup -- as dw_fwd_seg / dw_bwd_weight_seg there, which
// these are copies of with bf16 tensors; csrc/bf16.h states the rounding points).  Every input row is loaded, normalised, activated
// and rounded ONCE per strip; the strip kernels of dw_bf16_strip.h redo that for each of the k rows of every output row (4.5x per
// element for 3x3) and spend 34 vector instructions per output element on it.
// A value is rounded to bf16 where the policy makes it a bf16 tensor: after the consumer-side prologue and where an output is stored
// (its BatchNorm statistics are those of the stored values).
// These kernels are bound by vector-instruction issue, not by HBM, and live on v_pk_fma_f32: two FMAs per instruction need their
// operands in aligned register PAIRS.  A 16-byte fp32 load delivers four consecutive registers and the compiler pairs them by
// itself; four channels unpacked from two bf16 dwords are four unrelated registers and it does not (the first bf16 instantiation
// of this kernel ran 2x the fp32 one per element: 228 scalar FMAs where the fp32 kernel has 108 packed ones).  So the bf16
// kernels carry their four channels as two explicit <2 x float> vectors (pk4) from the load to the store.
typedef float dw_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int dw_u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 dw_bf16x2 __attribute__((ext_vector_type(2)));
struct pk4 { dw_f32x2 lo, hi; };
template <typename T> struct dw_vec { typedef float4 type; };
template <> struct dw_vec<bf16> { typedef pk4 type; };

template <typename V> __device__ __forceinline__ V vzero();
template <> __device__ __forceinline__ float4 vzero<float4>() { return zero4(); }
template <> __device__ __forceinline__ pk4 vzero<pk4>() { return pk4{dw_f32x2{0.f, 0.f}, dw_f32x2{0.f, 0.f}}; }
template <typename V> __device__ __forceinline__ V vfrom(float4 v);
template <> __device__ __forceinline__ float4 vfrom<float4>(float4 v) { return v; }
template <> __device__ __forceinline__ pk4 vfrom<pk4>(float4 v) { return pk4{dw_f32x2{v.x, v.y}, dw_f32x2{v.z, v.w}}; }
__device__ __forceinline__ float4 tof4(float4 v) { return v; }
__device__ __forceinline__ float4 tof4(pk4 v) { return make_float4(v.lo[0], v.lo[1], v.hi[0], v.hi[1]); }
__device__ __forceinline__ pk4 fma4(pk4 a, pk4 b, pk4 c) {
  return pk4{__builtin_elementwise_fma(a.lo, b.lo, c.lo), __builtin_elementwise_fma(a.hi, b.hi, c.hi)};
}
__device__ __forceinline__ pk4 add4(pk4 a, pk4 b) { return pk4{a.lo + b.lo, a.hi + b.hi}; }
__device__ __forceinline__ float4 vsel(bool c, float4 a) { return c ? a : zero4(); }
__device__ __forceinline__ pk4 vsel(bool c, pk4 a) {
  return pk4{dw_f32x2{c ? a.lo[0] : 0.f, c ? a.lo[1] : 0.f}, dw_f32x2{c ? a.hi[0] : 0.f, c ? a.hi[1] : 0.f}};
}
// two bf16 in a dword <-> a float pair (exact both ways for values that ARE bf16)
__device__ __forceinline__ dw_f32x2 dw_unpack(unsigned int d) {
  return dw_f32x2{__builtin_bit_cast(float, d << 16), __builtin_bit_cast(float, d & 0xffff0000u)};
}
__device__ __forceinline__ unsigned int dw_pack_rne(dw_f32x2 v) {      // v_cvt_pk_bf16_f32: round to nearest even
  const dw_bf16x2 h = {(__bf16)v[0], (__bf16)v[1]};
  return __builtin_bit_cast(unsigned int, h);
}
template <typename V, typename T> __device__ __forceinline__ V vld(const T* p);
template <> __device__ __forceinline__ float4 vld<float4, float>(const float* p) { return ld4(p); }
template <> __device__ __forceinline__ pk4 vld<pk4, bf16>(const bf16* p) {
  const dw_u32x2 d = *reinterpret_cast<const dw_u32x2*>(p);
  return pk4{dw_unpack(d[0]), dw_unpack(d[1])};
}
// store the four channels; returns what the tensor now holds (the bf16-rounded values on the mixed path: the statistics that
// follow are those of the stored tensor)
__device__ __forceinline__ float4 vstore(float* p, float4 v, int nt) {
  if (nt) st4_nt(p, v); else st4(p, v);
  return v;
}
__device__ __forceinline__ pk4 vstore(bf16* p, pk4 v, int) {
  const dw_u32x2 d = {dw_pack_rne(v.lo), dw_pack_rne(v.hi)};
  *reinterpret_cast<dw_u32x2*>(p) = d;
  return pk4{dw_unpack(d[0]), dw_unpack(d[1])};
}
// producer prologue: affine in packed form, clamp as one v_med3 per element (ReLU / ReLU6 / none) or the hard-swish family's
// arithmetic of act_apply (HS instantiations), then the rounding of the activated tensor
template <int PRO, bool HS>
__device__ __forceinline__ pk4 dwb_prologue(pk4 v, pk4 sc, pk4 sh, int act, float lo, float hi) {
  if (PRO == 0) return v;
  v = fma4(v, sc, sh);
  if (PRO == 2) {
    if (HS) {
      v.lo = dw_f32x2{act_apply(v.lo[0], act), act_apply(v.lo[1], act)};
      v.hi = dw_f32x2{act_apply(v.hi[0], act), act_apply(v.hi[1], act)};
    } else {
      v.lo = dw_f32x2{__builtin_amdgcn_fmed3f(v.lo[0], lo, hi), __builtin_amdgcn_fmed3f(v.lo[1], lo, hi)};
      v.hi = dw_f32x2{__builtin_amdgcn_fmed3f(v.hi[0], lo, hi), __builtin_amdgcn_fmed3f(v.hi[1], lo, hi)};
    }
  }
  return pk4{dw_unpack(dw_pack_rne(v.lo)), dw_unpack(dw_pack_rne(v.hi))};
}

template <int KS, int TW, int S, int PRO, bool HS>
__global__ __launch_bounds__(256) void dwb_fwd_seg(DwParams p) {
  typedef bf16 T;
  typedef typename dw_vec<T>::type V4;
  const T* const px_t = reinterpret_cast<const T*>(p.x);
  const T* const pw_t = reinterpret_cast<const T*>(p.w);
  T* const py_t = reinterpret_cast<T*>(p.y);
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
  V4 s1[2] = {vzero<V4>(), vzero<V4>()};
  // 5x5: the 25 weight vectors of a thread's channels live in LDS ([tap][channel lane], one ds_read_b128 per tap and
  // row) -- in registers they cost 100 VGPRs on top of the 5-row window and halve the occupancy
  constexpr bool WLDS = KS == 5;
  extern __shared__ __attribute__((aligned(16))) float4 dw_w_lds[];
  if (WLDS) {
    for (int i = t; i < KS * KS * p.c4s; i += 256) {
      const int tap = i / p.c4s, lane = i - tap * p.c4s;
      dw_w_lds[i] = tof4(vld<V4>(pw_t + (size_t)(p.flip ? KS * KS - 1 - tap : tap) * p.C + (cbase4 + lane) * 4));
    }
    __syncthreads();
  }
  if (active) {
    V4 wreg[WLDS ? 1 : KS * KS];
    if (!WLDS) {
#pragma unroll
      for (int i = 0; i < KS * KS; ++i) wreg[i] = vld<V4>(pw_t + (size_t)(p.flip ? KS * KS - 1 - i : i) * p.C + c);
    }
    auto wtap = [&](int i) { return WLDS ? vfrom<V4>(dw_w_lds[i * p.c4s + cl]) : wreg[WLDS ? 0 : i]; };
    V4 sc = vfrom<V4>(make_float4(1.f, 1.f, 1.f, 1.f)), sh = vzero<V4>();
    if (p.scale) { sc = vfrom<V4>(ld4(p.scale + c)); sh = vfrom<V4>(ld4(p.shift + c)); }
    const int act = p.act;
    const float act_lo = act == DL3P_ACT_NONE ? -DL3P_INF : 0.f;
    const float act_hi = (act == DL3P_ACT_NONE || act == DL3P_ACT_RELU) ? DL3P_INF : 6.f;
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
      const T* ximg = px_t + (size_t)n * p.H * p.W * p.ldx + c;
      int coff[SEG];
      bool cok[SEG];
#pragma unroll
      for (int i = 0; i < SEG; ++i) {
        const int ix = ix0 + i * rate;
        cok[i] = ix >= 0 && ix < p.W;
        coff[i] = ix * p.ldx;
      }
      V4 win[KS][SEG];
      dw_u32x2 raw[S][SEG];      // the next rows as LOADED (two dwords): unpacked behind this row's FMAs, so that nothing waits on them here
      // prime the window with the KS rows of the first output row (exec-masked loads, math afterwards)
      {
        const int oy = py + v0 * rate;
#pragma unroll
        for (int ky = 0; ky < KS; ++ky) {
          const int iy = oy * S - p.pad_t + ky * rate;
          const bool yok = iy >= 0 && iy < p.H;
          const T* xrow = ximg + (size_t)iy * p.W * p.ldx;
#pragma unroll
          for (int i = 0; i < SEG; ++i) {
            win[ky][i] = vzero<V4>();
            if (yok && cok[i]) win[ky][i] = vld<V4>(xrow + coff[i]);
          }
        }
#pragma unroll
        for (int ky = 0; ky < KS; ++ky) {
          const int iy = oy * S - p.pad_t + ky * rate;
          const bool yok = iy >= 0 && iy < p.H;
#pragma unroll
          for (int i = 0; i < SEG; ++i) {
            const V4 a = dwb_prologue<PRO, HS>(win[ky][i], sc, sh, act, act_lo, act_hi);
            win[ky][i] = vsel(yok && cok[i], a);
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
          const T* xrow = ximg + (size_t)iy * p.W * p.ldx;
#pragma unroll
          for (int i = 0; i < SEG; ++i) {
            raw[q][i] = dw_u32x2{0u, 0u};
            if (nyok[q] && cok[i]) raw[q][i] = *reinterpret_cast<const dw_u32x2*>(xrow + coff[i]);
          }
        }
        V4 acc[TW];
#pragma unroll
        for (int i = 0; i < TW; ++i) acc[i] = vzero<V4>();
        // keep the LDS weight reads inside the row loop: hoisted, the 25 vectors cost 100 VGPRs (1 wave per SIMD)
        if (WLDS) asm volatile("" ::: "memory");
#pragma unroll
        for (int ky = 0; ky < KS; ++ky)
#pragma unroll
          for (int kx = 0; kx < KS; ++kx) {
            const V4 wv = wtap(ky * KS + kx);
#pragma unroll
            for (int tw = 0; tw < TW; ++tw) acc[tw] = fma4(win[ky][tw * S + kx], wv, acc[tw]);
          }
        T* yrow = py_t + (((size_t)n * p.Ho + oy) * p.Wo + ox0) * p.ldy + c;
#pragma unroll
        for (int tw = 0; tw < TW; ++tw) {
          if (ox0 + tw * rate < p.Wo) {
            V4 vv = acc[tw];
            T* yp = yrow + (size_t)tw * rate * p.ldy;
            if (p.accumulate) vv = add4(vv, vld<V4>(yp));
            vv = vstore(yp, vv, p.nt);
              s1[0] = add4(s1[0], vv);
              s1[1] = fma4(vv, vv, s1[1]);
            
          }
        }
        // slide the window down by S rows
#pragma unroll
        for (int ky = 0; ky + S < KS; ++ky)
#pragma unroll
          for (int i = 0; i < SEG; ++i) win[ky][i] = win[ky + S][i];
#pragma unroll
        for (int q = 0; q < S; ++q)
#pragma unroll
          for (int i = 0; i < SEG; ++i) {
            const V4 a = dwb_prologue<PRO, HS>(pk4{dw_unpack(raw[q][i][0]), dw_unpack(raw[q][i][1])}, sc, sh, act, act_lo, act_hi);
            win[KS - S + q][i] = vsel(nyok[q] && cok[i], a);
          }
      }
    }
  }
  float4 s1f[2] = {tof4(s1[0]), tof4(s1[1])};
  if (p.partials) block_reduce_store<2>(s1f, active, pl, cl, p.c4s, p.px, cbase4, p.C, p.partials + (size_t)bx * 2 * p.C);
}


template <int KS, int TW, int S, int PRO, bool HS>
__global__ __launch_bounds__(256) void dwb_bwd_weight_seg(DwParams p) {
  typedef bf16 T;
  typedef typename dw_vec<T>::type V4;
  const T* const px_t = reinterpret_cast<const T*>(p.x);
  const T* const pdy_t = reinterpret_cast<const T*>(p.dy);
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
  V4 wacc[KS * KS];
#pragma unroll
  for (int i = 0; i < KS * KS; ++i) wacc[i] = vzero<V4>();
  if (active) {
    V4 sc = vfrom<V4>(make_float4(1.f, 1.f, 1.f, 1.f)), sh = vzero<V4>();
    if (p.scale) { sc = vfrom<V4>(ld4(p.scale + c)); sh = vfrom<V4>(ld4(p.shift + c)); }
    const int act = p.act;
    const float act_lo = act == DL3P_ACT_NONE ? -DL3P_INF : 0.f;
    const float act_hi = (act == DL3P_ACT_NONE || act == DL3P_ACT_RELU) ? DL3P_INF : 6.f;
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
      const T* ximg = px_t + (size_t)n * p.H * p.W * p.ldx + c;
      int coff[SEG];
      bool cok[SEG];
#pragma unroll
      for (int i = 0; i < SEG; ++i) {
        const int ix = ix0 + i * rate;
        cok[i] = ix >= 0 && ix < p.W;
        coff[i] = ix * p.ldx;
      }
      V4 win[KS][SEG];
      dw_u32x2 raw[S][SEG];      // the next rows as LOADED (two dwords): unpacked behind this row's FMAs, so that nothing waits on them here
      // prime the window with the KS rows of the first output row (exec-masked loads, math afterwards)
      {
        const int oy = py + v0 * rate;
#pragma unroll
        for (int ky = 0; ky < KS; ++ky) {
          const int iy = oy * S - p.pad_t + ky * rate;
          const bool yok = iy >= 0 && iy < p.H;
          const T* xrow = ximg + (size_t)iy * p.W * p.ldx;
#pragma unroll
          for (int i = 0; i < SEG; ++i) {
            win[ky][i] = vzero<V4>();
            if (yok && cok[i]) win[ky][i] = vld<V4>(xrow + coff[i]);
          }
        }
#pragma unroll
        for (int ky = 0; ky < KS; ++ky) {
          const int iy = oy * S - p.pad_t + ky * rate;
          const bool yok = iy >= 0 && iy < p.H;
#pragma unroll
          for (int i = 0; i < SEG; ++i) {
            const V4 a = dwb_prologue<PRO, HS>(win[ky][i], sc, sh, act, act_lo, act_hi);
            win[ky][i] = vsel(yok && cok[i], a);
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
          const T* xrow = ximg + (size_t)iy * p.W * p.ldx;
#pragma unroll
          for (int i = 0; i < SEG; ++i) {
            raw[q][i] = dw_u32x2{0u, 0u};
            if (nyok[q] && cok[i]) raw[q][i] = *reinterpret_cast<const dw_u32x2*>(xrow + coff[i]);
          }
        }
        // gradient of the raw conv output for this row's strip
        V4 dyv[TW];
        const T* drow = pdy_t + (((size_t)n * p.Ho + oy) * p.Wo + ox0) * p.lddy + c;
#pragma unroll
        for (int tw = 0; tw < TW; ++tw) {
          dyv[tw] = vzero<V4>();
          if (ox0 + tw * rate < p.Wo) dyv[tw] = vld<V4>(drow + (size_t)tw * rate * p.lddy);
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
            const V4 a = dwb_prologue<PRO, HS>(pk4{dw_unpack(raw[q][i][0]), dw_unpack(raw[q][i][1])}, sc, sh, act, act_lo, act_hi);
            win[KS - S + q][i] = vsel(nyok[q] && cok[i], a);
          }
      }
    }
  }
  float4 waccf[KS * KS];
#pragma unroll
  for (int i = 0; i < KS * KS; ++i) waccf[i] = tof4(wacc[i]);
  block_reduce_store<KS * KS>(waccf, active, pl, cl, p.c4s, p.px, cbase4, p.C,
                              p.partials + (size_t)bx * KS * KS * p.C);
}

