// Shared pieces of the fused inverted-residual kernels (irb_fwd.hip, irb_bwd.hip): the expand 1x1 conv of a
// MobileNetV2 block (deeplabv3p_mobilenetv2.py:38-74: Conv2D(6*Cin,1) -> BN -> ReLU6 -> DepthwiseConv2D(3) -> BN) is
// RECOMPUTED from the block's few-channel input wherever its output is needed, on the fp32-input matrix pipe, so the
// 6x-expanded tensor never exists in HBM.
//
// Tile convention (all three kernels): one v_mfma_f32_16x16x4_f32 group produces a 16-pixel x 16-channel tile of the
// expand output with the PIXEL on the lane (j = lane & 15) and four consecutive CHANNELS in the four result registers
// (channel = 16*tile + 4*(lane >> 4) + reg): D[co][pixel] = sum_k W[k][co] * x[pixel][k], A = W^T fragment, B = x.
// The reduction index is permuted so that a lane's B operands are CONTIGUOUS input channels: lane group q = lane >> 4
// owns k = q*(K/4) + s, s = 0 .. K/4-1, i.e. one 16-byte (K = 16) load of its pixel's row is all its operands.
#pragma once
#include "common.h"

typedef float irb_f4 __attribute__((ext_vector_type(4)));

struct IrbParams {
  const float* x; int ldx; const float* xs; const float* xh; int xact;   // block input + its lazy prologue
  const float* w1;                                                        // expand kernel [K][C]
  const float* s1; const float* h1; int act1;                             // expand BN scale / shift + activation
  const float* mu1; const float* is1; const float* coef1;                 // ... saved mean / invstd, backward coefficients [3][C]
  const float* wdw;                                                       // depthwise kernel [9][C]
  float* y; int ldy;                                                      // raw depthwise output (forward)
  const float* dy; int lddy;                                              // gradient w.r.t. the raw depthwise output
  float* partials;                                                        // statistic rows [rows][2][C]
  float* slabs;                                                           // weight-gradient slabs [rows][n]
  float* gx; int ldgx; int accumulate;                                    // gradient w.r.t. the block input
  const float* z0; int ldz0; const float* s0; const float* h0; int act0; const float* mu0; const float* is0;
  float* partials0;                                                       // BatchNorm in front of the block (optional)
  int N, H, W, C, Ho, Wo, pad_t, pad_l;
  int nseg, nband, band, ncg, units;
};

// The activation of the expand BatchNorm as a COMPILE-TIME choice where it is ReLU6 (every MobileNetV2 block; one v_med3_f32 and two
// compares instead of the branch-free generic evaluation of common.h: the kernels are bound by vector-instruction issue beside their
// MFMAs); ACT < 0: the run-time code.  irb_inside: act'(u) != 0 for the piecewise-linear activations (TF: 0 at the kinks).
template <int ACT>
__device__ __forceinline__ float irb_act(float u, int act_rt) {
  if constexpr (ACT == DL3P_ACT_RELU6) return __builtin_amdgcn_fmed3f(u, 0.f, 6.f);
  else if constexpr (ACT == DL3P_ACT_NONE) return u;
  else return act_apply(u, act_rt);
}
template <int ACT>
__device__ __forceinline__ float irb_act_grad_mul(float g, float u, int act_rt) {     // g * act'(u)
  if constexpr (ACT == DL3P_ACT_RELU6) return (u > 0.f && u < 6.f) ? g : 0.f;
  else if constexpr (ACT == DL3P_ACT_NONE) return g;
  else return g * act_grad(u, act_rt);
}
template <int ACT>
__device__ __forceinline__ float4 irb_act4(float4 u, int act_rt) {
  return make_float4(irb_act<ACT>(u.x, act_rt), irb_act<ACT>(u.y, act_rt), irb_act<ACT>(u.z, act_rt), irb_act<ACT>(u.w, act_rt));
}
template <int ACT>
__device__ __forceinline__ float4 irb_act_grad_mul4(float4 g, float4 u, int act_rt) {
  return make_float4(irb_act_grad_mul<ACT>(g.x, u.x, act_rt), irb_act_grad_mul<ACT>(g.y, u.y, act_rt),
                     irb_act_grad_mul<ACT>(g.z, u.z, act_rt), irb_act_grad_mul<ACT>(g.w, u.w, act_rt));
}
__device__ __forceinline__ float4 irb_sel4(bool c, float4 v) { return c ? v : make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float4 irb_f4_to_float4(irb_f4 z) { return make_float4(z[0], z[1], z[2], z[3]); }

template <int K>
__device__ __forceinline__ void irb_load_x(const float* p, float (&v)[K / 4]) {
  if constexpr (K == 16) {
    const float4 t = ld4(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  } else if constexpr (K == 24) {
    const float2 a = *reinterpret_cast<const float2*>(p);
    const float2 b = *reinterpret_cast<const float2*>(p + 2);
    const float2 c = *reinterpret_cast<const float2*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y; v[4] = c.x; v[5] = c.y;
  } else {
    static_assert(K == 32, "K in {16, 24, 32}");
    const float4 a = ld4(p), b = ld4(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  }
}

// the lazy prologue of the block input (BatchNorm of the previous project conv: act NONE; a materialised Add: identity).
// XACT: compile-time activation of the prologue (NONE in every MobileNetV2 block), < 0 = run-time code
template <int K, int XACT = -1>
__device__ __forceinline__ void irb_prologue(float (&v)[K / 4], const float (&xs)[K / 4], const float (&xh)[K / 4], int act) {
#pragma unroll
  for (int s = 0; s < K / 4; ++s) v[s] = irb_act<XACT>(fmaf(v[s], xs[s], xh[s]), act);
}

// z = W^T x for one 16-channel tile: acc[i] = z[channel 16*tile + 4*(lane>>4) + i][pixel lane&15]
template <int K>
__device__ __forceinline__ irb_f4 irb_expand(const float (&wf)[K / 4], const float (&xv)[K / 4]) {
  irb_f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < K / 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[s], xv[s], acc, 0, 0, 0);
  return acc;
}

// value of lane + 1 / lane - 1 inside the 16-lane pixel row: DPP row shifts (row_shl:1 reads lane + 1, row_shr:1 reads lane - 1;
// scripts/micro/dpp_rows.hip), no LDS round trip.  The row's last / first lane gets 0: no kernel uses it.
__device__ __forceinline__ float irb_from_next(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x101, 0xf, 0xf, true));
}
__device__ __forceinline__ float irb_from_prev(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));
}
__device__ __forceinline__ float4 irb_from_next4(float4 v) {
  return make_float4(irb_from_next(v.x), irb_from_next(v.y), irb_from_next(v.z), irb_from_next(v.w));
}
__device__ __forceinline__ float4 irb_from_prev4(float4 v) {
  return make_float4(irb_from_prev(v.x), irb_from_prev(v.y), irb_from_prev(v.z), irb_from_prev(v.w));
}

// sum over the 16 pixel lanes of a row (every lane of the row ends with the total)
__device__ __forceinline__ float irb_row_sum(float v) {
  v += __shfl_xor(v, 1, 16);
  v += __shfl_xor(v, 2, 16);
  v += __shfl_xor(v, 4, 16);
  v += __shfl_xor(v, 8, 16);
  return v;
}
__device__ __forceinline__ float4 irb_row_sum4(float4 v) {
  return make_float4(irb_row_sum(v.x), irb_row_sum(v.y), irb_row_sum(v.z), irb_row_sum(v.w));
}

// workgroup b of an XCD-round-robin dispatch -> the b % 8-th contiguous chunk of the work range (speed only)
__device__ __forceinline__ int irb_wg_index(int b, int nb) {
  const int per = nb >> 3;
  return (b & 7) * per + (b >> 3);
}
