// bf16 storage helpers for the mixed-precision path (BASELINE.json configs[4]: MobileNetV3-Large 1024x2048,
// "mixed-precision bf16"; reference switch train.py:37-46 `mixed_precision.set_global_policy`).
//
// Contract of the path (DESIGN.md "bf16"): activations and activation gradients live in HBM as bf16, GEMM
// weights are read from a bf16 mirror of the fp32 master copy, every accumulation, BatchNorm statistic, the softmax /
// loss and the optimiser state are fp32.  A value is rounded to bf16 (round to nearest even, v_cvt_pk_bf16_f32) exactly
// where Keras' mixed_bfloat16 policy makes a layer output a bf16 tensor: when a raw conv output is stored, and when a
// consumer applies the lazy BatchNorm + activation prologue (the activated value is rounded before it is used).
#pragma once
#include "common.h"

typedef __bf16 bf16;
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float bf16_round(float v) { return (float)(bf16)v; }
__device__ __forceinline__ float4 bf16_round4(float4 v) {
  return make_float4(bf16_round(v.x), bf16_round(v.y), bf16_round(v.z), bf16_round(v.w));
}

// 4 consecutive channels: 8-byte accesses for bf16, 16-byte for float
__device__ __forceinline__ float4 ld4(const bf16* p) {
  const bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
  return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
}
__device__ __forceinline__ void st4(bf16* p, float4 v) {
  const bf16x4 o = {(bf16)v.x, (bf16)v.y, (bf16)v.z, (bf16)v.w};
  *reinterpret_cast<bf16x4*>(p) = o;
}

// V consecutive channels (V = 4 or 8) as floats
template <int V> struct fvec { float v[V]; };
template <int V, typename T> __device__ __forceinline__ fvec<V> ldv(const T* p);
template <> __device__ __forceinline__ fvec<4> ldv<4, bf16>(const bf16* p) {
  const float4 a = ld4(p);
  return fvec<4>{{a.x, a.y, a.z, a.w}};
}
template <> __device__ __forceinline__ fvec<8> ldv<8, bf16>(const bf16* p) {
  const bf16x8 v = *reinterpret_cast<const bf16x8*>(p);
  fvec<8> o;
#pragma unroll
  for (int i = 0; i < 8; ++i) o.v[i] = (float)v[i];
  return o;
}
template <> __device__ __forceinline__ fvec<4> ldv<4, float>(const float* p) {
  const float4 a = ld4(p);
  return fvec<4>{{a.x, a.y, a.z, a.w}};
}
template <> __device__ __forceinline__ fvec<8> ldv<8, float>(const float* p) {
  const float4 a = ld4(p), b = ld4(p + 4);
  return fvec<8>{{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w}};
}
template <int V> __device__ __forceinline__ void stv(bf16* p, const fvec<V>& o);
template <> __device__ __forceinline__ void stv<4>(bf16* p, const fvec<4>& o) {
  st4(p, make_float4(o.v[0], o.v[1], o.v[2], o.v[3]));
}
template <> __device__ __forceinline__ void stv<8>(bf16* p, const fvec<8>& o) {
  bf16x8 v;
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = (bf16)o.v[i];
  *reinterpret_cast<bf16x8*>(p) = v;
}
template <int V> __device__ __forceinline__ void stv(float* p, const fvec<V>& o) {
#pragma unroll
  for (int i = 0; i < V; i += 4) st4(p + i, make_float4(o.v[i], o.v[i + 1], o.v[i + 2], o.v[i + 3]));
}
// per-channel coefficients (scale, shift, mean, ...): 16-byte loads -- c is a multiple of 4 and the arrays are 16-byte
// aligned (scalar loads here cost a thread 8 address-unit passes per coefficient vector; with 7 vectors per thread that
// was 4/5 of the BatchNorm-backward kernels on the 64 x 128 maps, where a thread only has one or two pixels to amortise them)
template <int V> __device__ __forceinline__ fvec<V> ldv_f32_or(const float* p, int c, float dflt) {
  fvec<V> o;
  if (p) {
#pragma unroll
    for (int i = 0; i < V; i += 4) {
      const float4 a = ld4(p + c + i);
      o.v[i] = a.x; o.v[i + 1] = a.y; o.v[i + 2] = a.z; o.v[i + 3] = a.w;
    }
  } else {
#pragma unroll
    for (int i = 0; i < V; ++i) o.v[i] = dflt;
  }
  return o;
}
template <int V> __device__ __forceinline__ fvec<V> fzero() {
  fvec<V> o;
#pragma unroll
  for (int i = 0; i < V; ++i) o.v[i] = 0.f;
  return o;
}
// the consumer-side prologue of the bf16 path: act(z * scale + shift) in fp32, rounded to bf16 (the activated tensor
// is a bf16 tensor under the mixed policy)
template <int V> __device__ __forceinline__ fvec<V> prologue_bf16(const fvec<V>& z, const fvec<V>& sc, const fvec<V>& sh,
                                                                  int act) {
  fvec<V> o;
#pragma unroll
  for (int i = 0; i < V; ++i) o.v[i] = bf16_round(act_apply(fmaf(z.v[i], sc.v[i], sh.v[i]), act));
  return o;
}

// (pixel lanes) x (channel lanes of V) decomposition of a 256-thread workgroup, channel lanes fastest
struct LaneSplit { int cs, px, nslab; };
static inline LaneSplit lane_split(int C, int V) {
  const int cv = C / V;
  int best = 1, best_used = 0;
  for (int d = 1; d <= cv && d <= 256; ++d) {
    if (cv % d) continue;
    if (d < 16 && d < cv) continue;
    const int used = (256 / d) * d;
    if (used > best_used || (used == best_used && d > best)) { best = d; best_used = used; }
  }
  LaneSplit s;
  s.cs = best; s.px = 256 / best; s.nslab = cv / best;
  return s;
}
