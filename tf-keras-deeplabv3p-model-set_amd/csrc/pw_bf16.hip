// Pointwise (1x1) convolution on the bf16 matrix cores (v_mfma_f32_16x16x32_bf16) for the mixed-precision path of
// BASELINE.json configs[4] (MobileNetV3-Large 1024x2048; reference switch train.py:37-46).  Same call sites as
// pwconv.hip: Conv2D(filters,(1,1)) at /root/reference deeplabv3p/models/layers.py:105,134,141,157,209,
// deeplabv3p_mobilenetv3.py:131-141,160-196, deeplabv3p/model.py:75.
//
//   fwd    Y[M,N]  = bf16(act(X*scale+shift)) @ Wt[N,K]^T (+bias)    (+ per-channel (sum, sum^2) of the STORED values)
//   dgrad  GX[M,K] (+)= DY[M,N] @ W[K,N]^T                            (the same kernel: B = W, reduce over N)
//   wgrad  GW[K,N] = bf16(act(X*scale+shift))^T @ DY                  (fp32 out; split over M, slab reduce)
//
// bf16 operands, fp32 accumulation.  At 16x the fp32 MFMA rate these layers are HBM-bound (K, N <= 960 against
// M = 8 192 .. 524 288 rows), so the kernels are laid out to read every activation byte once per column block and
// in 16-byte lanes: a workgroup owns 128 rows x up to 256 columns, the A tile passes registers -> (BN + activation
// prologue, rounding) -> LDS once per 32-deep K-step, B comes from the bf16 weight mirror (L2-resident).
// Operands are swapped in the MFMA (D = Wfrag x Xfrag) so that a lane ends with 4 CONSECUTIVE output channels of one
// pixel (one 8-byte store per accumulator).  The weight gradient needs both operands transposed (the reduction runs
// over rows): tiles are staged row-major and read back with ds_read_b64_tr_b16, the hardware transpose read.
#include "bf16.h"
#include <string.h>
#include <stdlib.h>
#include <type_traits>

int dl3p_bf16_force_kg = -1;      // dl3p_set_option("bf16_kg", 0 | 1 | 2 | 4 | -1): K groups of pwb_gemm by rule / never / pinned / default (DL3P_BF16_KG)

namespace {

constexpr int BK = 32;
constexpr int LP = 48;      // LDS row pitch in bf16 (96 B): ds_read_b128 of 16 rows x 4 k-groups is conflict-free

struct GemmB {
  const void* A; int lda;
  const float* scale; const float* shift; int act;
  const bf16* B; int ldb;           // [Nout][Kred] row-major
  const float* bias;
  void* Y; int ldy;
  float* partials;
  int M, K, N;
  int accumulate;
  int num_m_tiles;
  int dbg;              // DL3P_BF16_DBG ablation bits (timing experiments only): 1 no stores, 2 no A loads, 4 no B loads
  // data-gradient role with the BatchNorm-backward sums of the BatchNorm behind the gradient (bb_z != nullptr, STATS
  // instantiations): the partial rows hold (sum g', sum g' * xhat), g' = y * act'(z*scale+shift), as dl3p_bn_bwd_reduce_bf16
  // forms them from the stored (bf16) gradient
  const bf16* bb_z; int bb_ldz;
  const float* bb_scale; const float* bb_shift; const float* bb_mean; const float* bb_invstd; int bb_act;
};

// statistics of one stored float4 q at (row m, columns n..n+3): forward (sum, sum^2) or, with bb_z, the backward pair
template <bool BNB>
__device__ __forceinline__ void stats_accumulate_b(const GemmB& p, int m, int n, float4 q, float (&ss)[4], float (&sq)[4]) {
  if (BNB) {
    const float4 z = ld4(p.bb_z + (size_t)m * p.bb_ldz + n);
    const float4 sc = ld4(p.bb_scale + n), sh = ld4(p.bb_shift + n), mu = ld4(p.bb_mean + n), is = ld4(p.bb_invstd + n);
    const float qv[4] = {q.x, q.y, q.z, q.w}, zv[4] = {z.x, z.y, z.z, z.w}, scv[4] = {sc.x, sc.y, sc.z, sc.w};
    const float shv[4] = {sh.x, sh.y, sh.z, sh.w}, muv[4] = {mu.x, mu.y, mu.z, mu.w}, isv[4] = {is.x, is.y, is.z, is.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float d = qv[j] * act_grad(fmaf(zv[j], scv[j], shv[j]), p.bb_act);
      ss[j] += d;
      sq[j] = fmaf(d, (zv[j] - muv[j]) * isv[j], sq[j]);
    }
  } else {
    ss[0] += q.x; ss[1] += q.y; ss[2] += q.z; ss[3] += q.w;
    sq[0] = fmaf(q.x, q.x, sq[0]); sq[1] = fmaf(q.y, q.y, sq[1]); sq[2] = fmaf(q.z, q.z, sq[2]); sq[3] = fmaf(q.w, q.w, sq[3]);
  }
}

template <bool F32> struct AType { typedef bf16 type; };
template <> struct AType<true> { typedef float type; };

// raw (as loaded) 8-element chunk of an operand: 4 VGPRs for bf16, 8 for fp32
template <bool F32> struct Raw8 {
  bf16x8 v;
  __device__ __forceinline__ bf16x8 v_or_zero() const { return v; }
};
template <> struct Raw8<true> {
  float4 a, b;
  __device__ __forceinline__ bf16x8 v_or_zero() const { return bf16x8{}; }     // (never taken: fp32 operands are converted)
};
__device__ __forceinline__ Raw8<false> ld_raw(const bf16* p) { Raw8<false> r; r.v = *reinterpret_cast<const bf16x8*>(p); return r; }
__device__ __forceinline__ Raw8<true> ld_raw(const float* p) { Raw8<true> r; r.a = ld4(p); r.b = ld4(p + 4); return r; }
__device__ __forceinline__ float raw_get(const Raw8<false>& r, int e) { return (float)r.v[e]; }
__device__ __forceinline__ float raw_get(const Raw8<true>& r, int e) {
  return e == 0 ? r.a.x : e == 1 ? r.a.y : e == 2 ? r.a.z : e == 3 ? r.a.w : e == 4 ? r.b.x : e == 5 ? r.b.y : e == 6 ? r.b.z : r.b.w;
}

// Workgroup barrier that waits for this wave's LDS traffic only (sb_common.h lds_barrier): __syncthreads() is `s_waitcnt vmcnt(0)
// lgkmcnt(0); s_barrier`, and the vmcnt(0) drains the operand loads just issued for the K-steps ahead -- every K-step of pwb_gemm
// then contained a full memory round trip (~1 us at one wave per SIMD), whatever the prefetch depth.  The tiles handed over at the
// barrier live in LDS; the loads land in registers nobody else reads and the compiler keeps its own vmcnt bookkeeping for them.
__device__ __forceinline__ void lds_barrier_b() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// sum over the 16 lanes of a DPP row (lanes with equal l >> 4), left in all of them: the butterfly of __shfl_xor(v, 1 / 2 / 4 / 8)
// -- same tree, same bits -- as four DPP adds on the VALU instead of four ds_bpermute round trips through the LDS crossbar (the
// statistics epilogue of a 128-column tile reduced 64 values this way: 256 bpermutes per wave)
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_move<0xB1>(v);       // quad_perm [1, 0, 3, 2]
  v += dpp_move<0x4E>(v);       // quad_perm [2, 3, 0, 1]
  v += dpp_move<0x141>(v);      // row_half_mirror: the other quad of the 8-lane half
  v += dpp_move<0x140>(v);      // row_mirror: the other half of the row
  return v;
}

constexpr int KMAX_LDS = 2048;      // per-channel prologue coefficients of up to this many input channels sit in LDS

// NT: 16-column tiles per workgroup; MI: 16-row tiles per wave (BM = 64 MI rows per workgroup); A_F32 / Y_F32: the logits
// tensor (conv_upsample output and its gradient) stays fp32 so that the softmax / loss head is the fp32 one.
// The operand loads of K-step it + 2 are issued while step it is multiplied (two register slots): at 1-2 workgroups of
// work per CU and 0.3 us per K-step a single step of lookahead left the ~2 us HBM latency exposed on every step.
template <int NT, int MI, bool A_F32, bool Y_F32, bool STATS, bool BNB = false, int KG = 1>
// (with the statistics registers 128 x 64-row tiles need 180 VGPRs: under a three-workgroup bound they spilled 12 of them inside the
// K loop -- 33.9 against 18.0 us without statistics on 8192 x 672 -> 112; two workgroups per CU there)
// KG > 1 (few rows, long reduction: the 64 x 128 maps of configs[4] give 128 row tiles for 256 CUs and 20-40 K-steps each): KG groups
// of four waves share one output tile and take every KG-th K-step each, with their own operand tiles in LDS -- one workgroup per CU
// then has 4 KG waves whose stage / barrier / multiply phases overlap on the SIMDs instead of one wave per SIMD running them in
// series; the groups' accumulators are summed through LDS (the dead operand tiles) in a fixed order before group 0 runs the epilogue.
__global__ __launch_bounds__(256 * KG, KG > 1 ? 1 : (BNB ? 2 : ((NT * MI <= 8 && !(STATS && NT * MI > 4)) ? 4 : ((NT * MI <= 8 && !STATS) ? 3 : 2)))) void pwb_gemm(GemmB p) {
  constexpr int BM = 64 * MI, BN = 16 * NT;
  constexpr int NB = (BN * 4 + 255) / 256;      // 16-B chunks of the B tile per thread
  constexpr int PD = 2;                         // K-steps of loads in flight
  constexpr int NTHR = 256 * KG;
  constexpr int GROUP = (BM + BN) * LP;         // bf16 elements of one group's operand tiles
  constexpr int SLOT = 256 * MI * NT * 4;       // floats of one group's accumulators
  static_assert(KG == 1 || (size_t)GROUP * KG * 2 >= (size_t)SLOT * 4 * (KG / 2), "accumulator exchange does not fit the operand tiles");
  typedef typename AType<A_F32>::type TA;
  typedef typename AType<Y_F32>::type TY;
  __shared__ __attribute__((aligned(16))) bf16 oper[KG * GROUP];
  __shared__ __attribute__((aligned(16))) float coef[2 * KMAX_LDS];
  __shared__ float red[STATS ? 2 * 4 * BN : 1];
  const int grp = threadIdx.x >> 8;             // K group (0 when KG == 1)
  bf16* As = oper + grp * GROUP;
  bf16* Bs = As + BM * LP;
  const int t = threadIdx.x & 255, l = t & 63, w = t >> 6, r = l & 15, kg = l >> 4;
  const int n0 = blockIdx.y * BN;
  const int nk = (p.K + BK - 1) / BK;
  const int nkg = (nk + KG - 1) / KG;           // K-steps per group and tile (a group's last one may lie beyond K: zeros)
  const int my_tiles = (p.num_m_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  const int it_total = my_tiles * nkg;
  const TA* Ap = reinterpret_cast<const TA*>(p.A);
  TY* Yp = reinterpret_cast<TY*>(p.Y);
  const int ar = t >> 2, ac = (t & 3) * 8;      // A staging role: rows ar + 64 i; k offset ac
  const bool has_pro = p.scale != nullptr || p.act != DL3P_ACT_NONE;
  const float act_lo = p.act == DL3P_ACT_NONE ? -DL3P_INF : 0.f;
  const float act_hi = (p.act == DL3P_ACT_NONE || p.act == DL3P_ACT_RELU) ? DL3P_INF : 6.f;
  const bool coef_lds = p.scale != nullptr && nkg * KG * BK <= KMAX_LDS;
  if (coef_lds) {
    const int kpad = nkg * KG * BK;
    for (int i = threadIdx.x; i < kpad; i += NTHR) {
      coef[i] = i < p.K ? p.scale[i] : 1.f;
      coef[KMAX_LDS + i] = i < p.K ? p.shift[i] : 0.f;
    }
  }

  Raw8<A_F32> ra[PD][MI];
  bf16x8 rb[PD][NB];
  bool a_ok[PD][MI], b_ok[PD][NB];

  auto prefetch = [&](auto slot_c, int it) {
    constexpr int S = decltype(slot_c)::value;
    const int kt = (it % nkg) * KG + grp, mt = blockIdx.x + (it / nkg) * gridDim.x;
    const int m0 = mt * BM, k0 = kt * BK;
    const int k = k0 + ac;
    const bool kok = k < p.K;                     // K % 8 == 0: a chunk is inside or outside as a whole
    const int kc = kok ? k : 0;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int m = m0 + ar + 64 * i;
      a_ok[S][i] = kok && m < p.M;
      if (!(p.dbg & 2) || it < PD) ra[S][i] = ld_raw(Ap + (size_t)min(m, p.M - 1) * p.lda + kc);
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int idx = t + 256 * i;
      const int br = idx >> 2, bc = (idx & 3) * 8;
      const int n = n0 + br, kk = k0 + bc;
      b_ok[S][i] = idx < BN * 4 && n < p.N && kk < p.K;
      if (!(p.dbg & 4) || it < PD) rb[S][i] = *reinterpret_cast<const bf16x8*>(p.B + (size_t)min(n, p.N - 1) * p.ldb + (kk < p.K ? kk : 0));
    }
  };

  auto stage = [&](auto slot_c, int it) {
    constexpr int S = decltype(slot_c)::value;
    const int kt_ = (it % nkg) * KG + grp;
    const int k = kt_ * BK + ac;
    float sc[8], sh[8];
    if (p.scale) {
      if (coef_lds) {
        const float4 s0 = *reinterpret_cast<const float4*>(&coef[k]), s1 = *reinterpret_cast<const float4*>(&coef[k + 4]);
        const float4 h0 = *reinterpret_cast<const float4*>(&coef[KMAX_LDS + k]), h1 = *reinterpret_cast<const float4*>(&coef[KMAX_LDS + k + 4]);
        sc[0] = s0.x; sc[1] = s0.y; sc[2] = s0.z; sc[3] = s0.w; sc[4] = s1.x; sc[5] = s1.y; sc[6] = s1.z; sc[7] = s1.w;
        sh[0] = h0.x; sh[1] = h0.y; sh[2] = h0.z; sh[3] = h0.w; sh[4] = h1.x; sh[5] = h1.y; sh[6] = h1.z; sh[7] = h1.w;
      } else {
        const int kc = k < p.K ? k : 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) { sc[e] = p.scale[kc + e]; sh[e] = p.shift[kc + e]; }
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) { sc[e] = 1.f; sh[e] = 0.f; }
    }
    // interior tiles (wave-uniform: no M / K / N tail in this step) skip the zero-fill selects; the clamp family of
    // activations is one fma + one clamp per element (the same arithmetic as act_apply, so both paths round alike)
    const int mt_ = blockIdx.x + (it / nkg) * gridDim.x;
    const bool edge = mt_ * BM + BM > p.M || kt_ * BK + BK > p.K || n0 + BN > p.N;
    const bool hsw = p.act >= DL3P_ACT_HSWISH;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      bf16x8 v;
      if (!has_pro && !A_F32) {
        v = ra[S][i].v_or_zero();
      } else if (!hsw) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (bf16)fminf(fmaxf(fmaf(raw_get(ra[S][i], e), sc[e], sh[e]), act_lo), act_hi);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (bf16)act_apply(fmaf(raw_get(ra[S][i], e), sc[e], sh[e]), p.act);
      }
      if (edge && !a_ok[S][i]) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (bf16)0.f;
      }
      *reinterpret_cast<bf16x8*>(&As[(ar + 64 * i) * LP + ac]) = v;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int idx = t + 256 * i;
      if (idx < BN * 4) {
        bf16x8 v = rb[S][i];
        if (edge && !b_ok[S][i]) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = (bf16)0.f;
        }
        *reinterpret_cast<bf16x8*>(&Bs[(idx >> 2) * LP + (idx & 3) * 8]) = v;
      }
    }
  };

  f32x4v acc[MI][NT];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) acc[mi][ni] = (f32x4v){0.f, 0.f, 0.f, 0.f};
  float st_s[STATS ? NT : 1][4], st_q[STATS ? NT : 1][4];
  if (STATS) {
#pragma unroll
    for (int ni = 0; ni < NT; ++ni)
#pragma unroll
      for (int j = 0; j < 4; ++j) { st_s[ni][j] = 0.f; st_q[ni][j] = 0.f; }
  }

  auto step = [&](auto slot_c, int it) {
    stage(slot_c, it);
    lds_barrier_b();
    if (it + PD < it_total) prefetch(slot_c, it + PD);      // the slot just staged is free again
    bf16x8 af[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) af[mi] = *reinterpret_cast<const bf16x8*>(&As[(w * 16 * MI + mi * 16 + r) * LP + kg * 8]);
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) {
      const bf16x8 bfr = *reinterpret_cast<const bf16x8*>(&Bs[(ni * 16 + r) * LP + kg * 8]);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr, af[mi], acc[mi][ni], 0, 0, 0);
    }
    lds_barrier_b();
    if (KG > 1 && it % nkg == nkg - 1) {
      // the groups' partial sums, in a fixed order: (2, 3) -> (0, 1), then 1 -> 0; element e of thread t sits at ex[e * 256 + t]
      float* ex = reinterpret_cast<float*>(oper);
#pragma unroll
      for (int half = KG / 2; half >= 1; half >>= 1) {
        if (grp >= half && grp < 2 * half) {
          float* dst = ex + (grp - half) * SLOT + t;
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NT; ++ni)
#pragma unroll
              for (int j = 0; j < 4; ++j) dst[((mi * NT + ni) * 4 + j) * 256] = acc[mi][ni][j];
        }
        __syncthreads();
        if (grp < half) {
          const float* src = ex + grp * SLOT + t;
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NT; ++ni)
#pragma unroll
              for (int j = 0; j < 4; ++j) acc[mi][ni][j] += src[((mi * NT + ni) * 4 + j) * 256];
        }
        __syncthreads();
      }
      if (grp != 0) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NT; ++ni) acc[mi][ni] = (f32x4v){0.f, 0.f, 0.f, 0.f};
      }
    }
    if (it % nkg == nkg - 1 && grp == 0) {
      // epilogue of this M tile: lane (r, kg) holds channels n = n0 + 16 ni + 4 kg + j of pixel m = m0 + 16 MI w + 16 mi + r
      const int mt = blockIdx.x + (it / nkg) * gridDim.x;
      const int m0 = mt * BM;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const int m = m0 + w * 16 * MI + mi * 16 + r;
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) {
          const int n = n0 + ni * 16 + kg * 4;
          f32x4v v = acc[mi][ni];
          acc[mi][ni] = (f32x4v){0.f, 0.f, 0.f, 0.f};
          if (m < p.M && n < p.N) {
            float4 o = make_float4(v[0], v[1], v[2], v[3]);
            if (p.bias) o = add4(o, ld4(p.bias + n));
            TY* yp = Yp + (size_t)m * p.ldy + n;
            if (p.accumulate) o = add4(o, ld4(yp));
            if (!(p.dbg & 1)) st4(yp, o);
            if (STATS) {
              const float4 q = Y_F32 ? o : bf16_round4(o);      // statistics of the values the consumers will read
              stats_accumulate_b<BNB>(p, m, n, q, st_s[ni], st_q[ni]);
            }
          }
        }
      }
    }
  };

  __syncthreads();                 // the coefficient table
  if (it_total > 0) prefetch(std::integral_constant<int, 0>{}, 0);
  if (it_total > 1) prefetch(std::integral_constant<int, 1>{}, 1);
  for (int it = 0; it < it_total; it += PD) {
    step(std::integral_constant<int, 0>{}, it);
    if (it + 1 < it_total) step(std::integral_constant<int, 1>{}, it + 1);
  }

  if (STATS) {
    // sum over the 16 pixel lanes of a wave, then over the 4 waves; one partial row per workgroup
#pragma unroll
    for (int ni = 0; ni < NT; ++ni)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float s1 = st_s[ni][j], s2 = st_q[ni][j];
        s1 = row16_sum(s1); s2 = row16_sum(s2);
        if (r == 0 && grp == 0) {
          red[(0 * 4 + w) * BN + ni * 16 + kg * 4 + j] = s1;
          red[(1 * 4 + w) * BN + ni * 16 + kg * 4 + j] = s2;
        }
      }
    __syncthreads();
    if (p.partials && grp == 0) {
      for (int i = t; i < 2 * BN; i += 256) {
        const int which = i / BN, nn = i - which * BN;
        if (n0 + nn < p.N)
          p.partials[((size_t)blockIdx.x * 2 + which) * p.N + n0 + nn] =
              red[(which * 4 + 0) * BN + nn] + red[(which * 4 + 1) * BN + nn] + red[(which * 4 + 2) * BN + nn] + red[(which * 4 + 3) * BN + nn];
      }
    }
  }
}

// ------------------------------------------------------------------------------ streaming GEMM: B resident in LDS
// For K <= 640 the whole [BN][K] weight tile of a workgroup fits LDS as bf16.  Then nothing in the K loop is shared
// any more: every WAVE walks its own 32-row tiles, loads its A fragments straight from global memory in MFMA operand
// layout (lane (r, kg) reads the 16 bytes X[m0 + r][32 ks + 8 kg ..]: the same 16 rows x 64 B per instruction the staged
// kernel issued), applies the BatchNorm + activation prologue in registers and multiplies against B fragments read from
// LDS.  No barrier, no A staging pass, no lockstep between waves: the 8 waves of a workgroup drift apart and cover each
// other's memory latency, which is what these HBM-bound layers need (the tiled kernel above spent ~2 us per 32-deep
// K-step in staging + two barriers, with or without its global loads).  Loads run one chunk of CH K-steps ahead: raw
// registers are converted to fragments, re-issued for the next chunk, then the chunk's MFMAs run.
template <int NT, bool A_F32, bool Y_F32, bool STATS, bool BNB = false>
__global__ __launch_bounds__(512, 1) void pwb_stream(GemmB p, int kp, int tiles) {
  constexpr int MI = 2, BN = 16 * NT, NWAVE = 8;
  // K-steps per chunk (raw + converted fragments of a chunk live next to 64 accumulator and, with statistics, 64
  // statistics registers: 256 VGPRs at two waves per SIMD)
  constexpr int SCH = (NT == 8 && (STATS || A_F32)) ? 3 : 5;
  typedef typename AType<A_F32>::type TA;
  typedef typename AType<Y_F32>::type TY;
  extern __shared__ __attribute__((aligned(16))) unsigned char sm_raw[];
  const int nk = (p.K + BK - 1) / BK, kpad = nk * BK;
  bf16* Bs = reinterpret_cast<bf16*>(sm_raw);
  float* coef = reinterpret_cast<float*>(sm_raw + (size_t)BN * kp * 2);     // [2][kpad]
  float* red = coef + 2 * kpad;                                             // [2][NWAVE][BN] (STATS)
  const int t = threadIdx.x, l = t & 63, w = t >> 6, r = l & 15, kg = l >> 4;
  const int n0 = blockIdx.y * BN;
  const TA* Ap = reinterpret_cast<const TA*>(p.A);
  TY* Yp = reinterpret_cast<TY*>(p.Y);
  {
    const int cpr = kpad / 8;                          // 16-B chunks per B row
    for (int idx = t; idx < BN * cpr; idx += 512) {
      const int n = idx / cpr, kc = (idx - n * cpr) * 8;
      bf16x8 v = {};
      if (n0 + n < p.N && kc < p.K) v = *reinterpret_cast<const bf16x8*>(p.B + (size_t)(n0 + n) * p.ldb + kc);
      *reinterpret_cast<bf16x8*>(&Bs[n * kp + kc]) = v;
    }
    for (int i = t; i < kpad; i += 512) {
      coef[i] = (p.scale && i < p.K) ? p.scale[i] : 1.f;
      coef[kpad + i] = (p.scale && i < p.K) ? p.shift[i] : 0.f;
    }
  }
  __syncthreads();
  const bool has_pro = p.scale != nullptr || p.act != DL3P_ACT_NONE;
  const bool hsw = p.act >= DL3P_ACT_HSWISH;
  const float act_lo = p.act == DL3P_ACT_NONE ? -DL3P_INF : 0.f;
  const float act_hi = (p.act == DL3P_ACT_NONE || p.act == DL3P_ACT_RELU) ? DL3P_INF : 6.f;
  const int ncpt = (nk + SCH - 1) / SCH;               // chunks per tile
  const int first = blockIdx.x * NWAVE + w, stride = gridDim.x * NWAVE;
  const int my_tiles = first < tiles ? (tiles - first + stride - 1) / stride : 0;
  const int n_items = my_tiles * ncpt;

  Raw8<A_F32> raw[SCH][MI];
  f32x4v acc[MI][NT];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) acc[mi][ni] = (f32x4v){0.f, 0.f, 0.f, 0.f};
  float st_s[STATS ? NT : 1][4], st_q[STATS ? NT : 1][4];
  if (STATS) {
#pragma unroll
    for (int ni = 0; ni < NT; ++ni)
#pragma unroll
      for (int j = 0; j < 4; ++j) { st_s[ni][j] = 0.f; st_q[ni][j] = 0.f; }
  }

  auto issue = [&](int item) {
    const int tile = first + (item / ncpt) * stride, c = item % ncpt;
#pragma unroll
    for (int s2 = 0; s2 < SCH; ++s2) {
      if (c * SCH + s2 < nk) {                                         // (wave-uniform: short K loops skip the slots they lack)
        const int k = min((c * SCH + s2) * BK + kg * 8, p.K - 8);      // clamped: chunks past K are zeroed below
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          const int m = min(tile * 32 + mi * 16 + r, p.M - 1);
          raw[s2][mi] = ld_raw(Ap + (size_t)m * p.lda + k);
        }
      }
    }
  };

  if (n_items > 0) issue(0);
  for (int item = 0; item < n_items; ++item) {
    const int tile = first + (item / ncpt) * stride, c = item % ncpt;
    const int m0 = tile * 32;
    bf16x8 fr[SCH][MI];
#pragma unroll
    for (int s2 = 0; s2 < SCH; ++s2) {
      if (c * SCH + s2 >= nk) continue;
      const int k = (c * SCH + s2) * BK + kg * 8;
      const bool kok = k < p.K;
      float sc[8], sh[8];
      if (has_pro) {
        const int kc = kok ? k : 0;
        const float4 s0 = *reinterpret_cast<const float4*>(&coef[kc]), s1 = *reinterpret_cast<const float4*>(&coef[kc + 4]);
        const float4 h0 = *reinterpret_cast<const float4*>(&coef[kpad + kc]), h1 = *reinterpret_cast<const float4*>(&coef[kpad + kc + 4]);
        sc[0] = s0.x; sc[1] = s0.y; sc[2] = s0.z; sc[3] = s0.w; sc[4] = s1.x; sc[5] = s1.y; sc[6] = s1.z; sc[7] = s1.w;
        sh[0] = h0.x; sh[1] = h0.y; sh[2] = h0.z; sh[3] = h0.w; sh[4] = h1.x; sh[5] = h1.y; sh[6] = h1.z; sh[7] = h1.w;
      }
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        bf16x8 v;
        if (!has_pro && !A_F32) {
          v = raw[s2][mi].v_or_zero();
        } else if (!has_pro) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = (bf16)raw_get(raw[s2][mi], e);
        } else if (!hsw) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = (bf16)fminf(fmaxf(fmaf(raw_get(raw[s2][mi], e), sc[e], sh[e]), act_lo), act_hi);
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = (bf16)act_apply(fmaf(raw_get(raw[s2][mi], e), sc[e], sh[e]), p.act);
        }
        if (!kok || m0 + mi * 16 + r >= p.M) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = (bf16)0.f;
        }
        fr[s2][mi] = v;
      }
    }
    if (item + 1 < n_items) issue(item + 1);            // the raw registers are free again: next chunk on its way
#pragma unroll
    for (int s2 = 0; s2 < SCH; ++s2) {
      const int ks = c * SCH + s2;
      if (ks < nk) {
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) {
          const bf16x8 bfr = *reinterpret_cast<const bf16x8*>(&Bs[(ni * 16 + r) * kp + ks * BK + kg * 8]);
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr, fr[s2][mi], acc[mi][ni], 0, 0, 0);
        }
      }
    }
    if (c == ncpt - 1) {
      // lane (r, kg) holds channels n = n0 + 16 ni + 4 kg + j of pixel m = m0 + 16 mi + r
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const int m = m0 + mi * 16 + r;
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) {
          const int n = n0 + ni * 16 + kg * 4;
          f32x4v v = acc[mi][ni];
          acc[mi][ni] = (f32x4v){0.f, 0.f, 0.f, 0.f};
          if (m < p.M && n < p.N) {
            float4 o = make_float4(v[0], v[1], v[2], v[3]);
            if (p.bias) o = add4(o, ld4(p.bias + n));
            TY* yp = Yp + (size_t)m * p.ldy + n;
            if (p.accumulate) o = add4(o, ld4(yp));
            st4(yp, o);
            if (STATS) {
              const float4 q = Y_F32 ? o : bf16_round4(o);
              stats_accumulate_b<BNB>(p, m, n, q, st_s[ni], st_q[ni]);
            }
          }
        }
      }
    }
  }

  if (STATS) {
#pragma unroll
    for (int ni = 0; ni < NT; ++ni)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float s1 = st_s[ni][j], s2 = st_q[ni][j];
        s1 = row16_sum(s1); s2 = row16_sum(s2);
        if (r == 0) {
          red[(0 * NWAVE + w) * BN + ni * 16 + kg * 4 + j] = s1;
          red[(1 * NWAVE + w) * BN + ni * 16 + kg * 4 + j] = s2;
        }
      }
    __syncthreads();
    if (p.partials) {
      for (int i = t; i < 2 * BN; i += 512) {
        const int which = i / BN, nn = i - which * BN;
        if (n0 + nn < p.N) {
          float a = 0.f;
#pragma unroll
          for (int q = 0; q < NWAVE; ++q) a += red[(which * NWAVE + q) * BN + nn];
          p.partials[((size_t)blockIdx.x * 2 + which) * p.N + n0 + nn] = a;
        }
      }
    }
  }
}

struct StreamPlan { bool ok; int nt, kp, gy; size_t lds; };
StreamPlan stream_plan_b(int K, int N, bool stats) {
  StreamPlan pl = {};
  const int kpad = ceil_div(K, BK) * BK;
  pl.kp = kpad + 8;
  if (kpad > 640) return pl;
  int nt = 2;
  while (nt < 8 && 16 * nt < N) nt *= 2;
  auto bytes = [&](int ntv) { return (size_t)16 * ntv * pl.kp * 2 + (size_t)2 * kpad * 4 + (stats ? (size_t)2 * 8 * 16 * ntv * 4 : 0); };
  while (nt > 2 && bytes(nt) > 150 * 1024) nt /= 2;
  if (bytes(nt) > 150 * 1024) return pl;
  // narrow column blocks re-read A once per block: keep the staged kernel when that is more than 3 passes
  if (ceil_div(N, 16 * nt) > 3) return pl;
  pl.ok = true; pl.nt = nt; pl.gy = ceil_div(N, 16 * nt); pl.lds = bytes(nt);
  return pl;
}

template <int NT, bool A_F32, bool Y_F32, bool STATS, bool BNB = false>
void launch_stream_one(const GemmB& p, const StreamPlan& pl, dim3 grid, int tiles, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)pwb_stream<NT, A_F32, Y_F32, STATS, BNB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL((pwb_stream<NT, A_F32, Y_F32, STATS, BNB>), grid, dim3(512), pl.lds, st, p, pl.kp, tiles);
}
template <bool A_F32, bool Y_F32, bool STATS, bool BNB = false>
void launch_stream(const GemmB& p, const StreamPlan& pl, dim3 grid, int tiles, hipStream_t st) {
  switch (pl.nt) {
    case 2: launch_stream_one<2, A_F32, Y_F32, STATS, BNB>(p, pl, grid, tiles, st); break;
    case 4: launch_stream_one<4, A_F32, Y_F32, STATS, BNB>(p, pl, grid, tiles, st); break;
    default: launch_stream_one<8, A_F32, Y_F32, STATS, BNB>(p, pl, grid, tiles, st); break;
  }
}

// M <= 64 rows (convs behind a global pooling: ASPP image pooling, squeeze-excite): one wave per output column, lanes
// stride the reduction in 16-byte chunks
template <bool A_F32, bool Y_F32>
__global__ __launch_bounds__(64) void pwb_tiny(GemmB p) {
  typedef typename AType<A_F32>::type TA;
  typedef typename AType<Y_F32>::type TY;
  const int n = blockIdx.x, l = threadIdx.x;
  const TA* Ap = reinterpret_cast<const TA*>(p.A);
  TY* Yp = reinterpret_cast<TY*>(p.Y);
  const bool has_pro = p.scale != nullptr || p.act != DL3P_ACT_NONE;
  float s1 = 0.f, s2 = 0.f;
  for (int m = 0; m < p.M; ++m) {
    float acc = 0.f;
    for (int k = l * 8; k < p.K; k += 512) {
      fvec<8> a = ldv<8>(Ap + (size_t)m * p.lda + k);
      const fvec<8> wv = ldv<8>(p.B + (size_t)n * p.ldb + k);
      if (has_pro) a = prologue_bf16<8>(a, ldv_f32_or<8>(p.scale, k, 1.f), ldv_f32_or<8>(p.shift, k, 0.f), p.act);
      else if (A_F32) {
#pragma unroll
        for (int e = 0; e < 8; ++e) a.v[e] = bf16_round(a.v[e]);
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) acc = fmaf(a.v[e], wv.v[e], acc);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (l == 0) {
      if (p.bias) acc += p.bias[n];
      TY* yp = Yp + (size_t)m * p.ldy + n;
      if (p.accumulate) acc += (float)*yp;
      *yp = (TY)acc;
      const float q = Y_F32 ? acc : bf16_round(acc);
      s1 += q; s2 = fmaf(q, q, s2);
    }
  }
  if (p.partials && l == 0) { p.partials[n] = s1; p.partials[p.N + n] = s2; }
}

// ------------------------------------------------------------------------------ weight gradient
struct WgradB {
  const bf16* X; int ldx; const float* scale; const float* shift; int act;
  const void* DY; int lddy;
  float* slabs;
  int M, K, N;
  int ktiles, ntiles, mrows;      // rows of M per chunk (multiple of 32)
};

__device__ __forceinline__ bf16x4 tr_read(const bf16* lds_ptr) {
  typedef short s16x4 __attribute__((ext_vector_type(4)));
  const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)lds_ptr);
  return *reinterpret_cast<const bf16x4*>(&v);
}

// MS rows of M per step (one barrier pair per step: at 32 rows a step was 8 MFMAs per wave against two barriers and a
// staging pass, and the kernel sat at 1/4 of its HBM floor whatever the load lookahead)
template <int NW, bool DY_F32>
__global__ __launch_bounds__(256, 2) void pwb_wgrad(WgradB p) {
  constexpr int TK = 64, TN = 16 * NW, MS = 128;
  constexpr int XP = TK + 8, DP = TN + 8;           // pitches (bf16): 8-byte aligned rows for the transposed reads
  constexpr int NX = MS * TK / 8 / 256;             // 16-B chunks of the X tile per thread (4)
  constexpr int ND = MS * TN / 8 / 256;             // ... of the DY tile (NW / 2 * 2)
  typedef typename AType<DY_F32>::type TD;
  __shared__ __attribute__((aligned(16))) bf16 Xs[MS * XP];
  __shared__ __attribute__((aligned(16))) bf16 Ds[MS * DP];
  const int t = threadIdx.x, l = t & 63, w = t >> 6;
  const int tile = blockIdx.x, kt = tile / p.ntiles, nt = tile - kt * p.ntiles;
  const int k0 = kt * TK, n0 = nt * TN;
  const int mbeg = blockIdx.y * p.mrows, mend = min(p.M, mbeg + p.mrows);
  const TD* Dp = reinterpret_cast<const TD*>(p.DY);
  // staging roles: X tile rows xm + 32 i (i < NX), the thread's 8 channels are fixed for the kernel
  const int xm = t >> 3, xk = k0 + (t & 7) * 8;
  const bool xk_ok = xk < p.K;
  const fvec<8> sc = ldv_f32_or<8>(xk_ok ? p.scale : nullptr, xk, 1.f), sh = ldv_f32_or<8>(xk_ok ? p.shift : nullptr, xk, 0.f);
  const bool has_pro = p.scale != nullptr || p.act != DL3P_ACT_NONE;
  const bool hsw = p.act >= DL3P_ACT_HSWISH;
  const float act_lo = p.act == DL3P_ACT_NONE ? -DL3P_INF : 0.f;
  const float act_hi = (p.act == DL3P_ACT_NONE || p.act == DL3P_ACT_RELU) ? DL3P_INF : 6.f;
  constexpr int DCPR = TN / 8;                      // DY chunks per row
  const int dm = t / DCPR, dc = (t - dm * DCPR) * 8;   // DY tile rows dm + (256 / DCPR) i
  constexpr int DROWS = 256 / DCPR;
  const bool dn_ok = n0 + dc < p.N;
  f32x4v acc[NW];
#pragma unroll
  for (int ni = 0; ni < NW; ++ni) acc[ni] = (f32x4v){0.f, 0.f, 0.f, 0.f};
  // transposed-read addresses: lane (g = l >> 4, i = l & 15 = 4q + pp) supplies row 8g + q (+4), columns 4pp..4pp+3
  const int g = l >> 4, q = (l & 15) >> 2, pp = l & 3;
  Raw8<false> rx[NX];
  Raw8<DY_F32> rd[ND];
  auto prefetch = [&](int m0) {
#pragma unroll
    for (int i = 0; i < NX; ++i)
      rx[i] = ld_raw(p.X + (size_t)min(m0 + xm + 32 * i, p.M - 1) * p.ldx + (xk_ok ? xk : 0));
#pragma unroll
    for (int i = 0; i < ND; ++i)
      rd[i] = ld_raw(Dp + (size_t)min(m0 + dm + DROWS * i, p.M - 1) * p.lddy + (dn_ok ? n0 + dc : 0));
  };
  if (mbeg < mend) prefetch(mbeg);
  for (int m0 = mbeg; m0 < mend; m0 += MS) {
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      bf16x8 v;
      if (!has_pro) {
        v = rx[i].v;
      } else if (!hsw) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (bf16)fminf(fmaxf(fmaf((float)rx[i].v[e], sc.v[e], sh.v[e]), act_lo), act_hi);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (bf16)act_apply(fmaf((float)rx[i].v[e], sc.v[e], sh.v[e]), p.act);
      }
      if (!xk_ok || m0 + xm + 32 * i >= mend) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (bf16)0.f;
      }
      *reinterpret_cast<bf16x8*>(&Xs[(xm + 32 * i) * XP + (t & 7) * 8]) = v;
    }
#pragma unroll
    for (int i = 0; i < ND; ++i) {
      bf16x8 dv;
      const bool ok = dn_ok && m0 + dm + DROWS * i < mend;
#pragma unroll
      for (int e = 0; e < 8; ++e) dv[e] = ok ? (bf16)raw_get(rd[i], e) : (bf16)0.f;
      *reinterpret_cast<bf16x8*>(&Ds[(dm + DROWS * i) * DP + dc]) = dv;
    }
    __syncthreads();
    if (m0 + MS < mend) prefetch(m0 + MS);
    // per 32-row sub-step: A operand (16 k rows x 32 m) of this wave, B operands (32 m x 16 n) per column tile
#pragma unroll
    for (int ms = 0; ms < MS / 32; ++ms) {
      const bf16x4 a0 = tr_read(&Xs[(32 * ms + 8 * g + q) * XP + w * 16 + 4 * pp]);
      const bf16x4 a1 = tr_read(&Xs[(32 * ms + 8 * g + 4 + q) * XP + w * 16 + 4 * pp]);
      const bf16x8 af = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
#pragma unroll
      for (int ni = 0; ni < NW; ++ni) {
        const bf16x4 b0 = tr_read(&Ds[(32 * ms + 8 * g + q) * DP + ni * 16 + 4 * pp]);
        const bf16x4 b1 = tr_read(&Ds[(32 * ms + 8 * g + 4 + q) * DP + ni * 16 + 4 * pp]);
        const bf16x8 bfr = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
        acc[ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bfr, acc[ni], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  // D[row = 4 (l >> 4) + j -> k][col = l & 15 -> n]
  float* slab = p.slabs + (size_t)blockIdx.y * p.K * p.N;
#pragma unroll
  for (int ni = 0; ni < NW; ++ni) {
    const int n = n0 + ni * 16 + (l & 15);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = k0 + w * 16 + 4 * g + j;
      if (k < p.K && n < p.N) slab[(size_t)k * p.N + n] = acc[ni][j];
    }
  }
}

// tiny M: gw[k][n] = sum_m a[m][k] dy[m][n], gb[n] = sum_m dy[m][n]; one thread per (k, n)
template <bool DY_F32>
__global__ __launch_bounds__(256) void pwb_tiny_wgrad(WgradB p, float* gw, float* gb) {
  typedef typename AType<DY_F32>::type TD;
  const TD* Dp = reinterpret_cast<const TD*>(p.DY);
  const long long total = (long long)p.K * p.N;
  const bool has_pro = p.scale != nullptr || p.act != DL3P_ACT_NONE;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int k = (int)(i / p.N), n = (int)(i - (long long)k * p.N);
    float acc = 0.f, bsum = 0.f;
    for (int m = 0; m < p.M; ++m) {
      float a = (float)p.X[(size_t)m * p.ldx + k];
      if (has_pro) a = bf16_round(act_apply(fmaf(a, p.scale ? p.scale[k] : 1.f, p.scale ? p.shift[k] : 0.f), p.act));
      const float d = bf16_round((float)Dp[(size_t)m * p.lddy + n]);
      acc = fmaf(a, d, acc);
      bsum += d;
    }
    gw[i] = acc;
    if (gb && k == 0) gb[n] = bsum;
  }
}

// column sums of DY (bias gradient): partial rows, then the row reducer
template <bool DY_F32>
__global__ __launch_bounds__(256) void pwb_colsum(const void* dy, int lddy, int M, int N, float* partial) {
  typedef typename AType<DY_F32>::type TD;
  const TD* Dp = reinterpret_cast<const TD*>(dy);
  const int nl = N < 256 ? N : 256;             // column lanes
  const int rl = 256 / nl;                      // row lanes
  const int c = threadIdx.x % nl, rr = threadIdx.x / nl;
  __shared__ float sm[256];
  for (int c0 = 0; c0 < N; c0 += nl) {
    float a = 0.f;
    if (rr < rl && c0 + c < N)
      for (long long m = (long long)blockIdx.x * rl + rr; m < M; m += (long long)gridDim.x * rl)
        a += bf16_round((float)Dp[(size_t)m * lddy + c0 + c]);
    __syncthreads();
    sm[threadIdx.x] = (rr < rl) ? a : 0.f;
    __syncthreads();
    if (rr == 0 && c0 + c < N) {
      float s = 0.f;
      for (int q = 0; q < rl; ++q) s += sm[q * nl + c];
      partial[(size_t)blockIdx.x * N + c0 + c] = s;
    }
  }
}

int check_mat_b(const char* fn, const void* ptr, int ld, int cols, bool f32) {
  DL3P_CHECK_ARG(ptr != nullptr, "%s: null pointer", fn);
  DL3P_CHECK_ARG(cols > 0 && cols % 8 == 0, "%s: channel count %d must be a positive multiple of 8 (bf16 path)", fn, cols);
  DL3P_CHECK_ARG(ld % 8 == 0 && ld >= cols && aligned16(ptr), "%s: bad layout (ld=%d)", fn, ld);
  (void)f32;
  return DL3P_OK;
}

int pick_nt_b(int N, bool stats) {
  const int cap = stats ? 8 : 16;
  int nt = 2;
  while (nt < cap && 16 * nt < N) nt *= 2;
  return nt;
}

template <int MI, bool A_F32, bool Y_F32, bool STATS, bool BNB = false>
void launch_gemm_b(const GemmB& p, int nt, dim3 grid, hipStream_t st) {
  switch (nt) {
    case 2: hipLaunchKernelGGL((pwb_gemm<2, MI, A_F32, Y_F32, STATS, BNB>), grid, dim3(256), 0, st, p); break;
    case 4: hipLaunchKernelGGL((pwb_gemm<4, MI, A_F32, Y_F32, STATS, BNB>), grid, dim3(256), 0, st, p); break;
    case 8: hipLaunchKernelGGL((pwb_gemm<8, MI, A_F32, Y_F32, STATS, BNB>), grid, dim3(256), 0, st, p); break;
    default:
      // 256 columns per workgroup: 64-row tiles only (128 accumulator registers at 128 rows spill next to the load ring)
      if constexpr (!STATS && MI == 1) hipLaunchKernelGGL((pwb_gemm<16, 1, A_F32, Y_F32, false>), grid, dim3(256), 0, st, p);
      break;
  }
}

int gemm_b(const char* fn, GemmB p, bool a_f32, bool y_f32, int* rows_out, hipStream_t st) {
  DL3P_CHECK_ARG(!(a_f32 && y_f32), "%s: fp32 in and out is the fp32 library's job", fn);
  if (p.M <= 64) {
    if (rows_out) *rows_out = 1;
    if (a_f32) hipLaunchKernelGGL((pwb_tiny<true, false>), dim3(p.N), dim3(64), 0, st, p);
    else if (y_f32) hipLaunchKernelGGL((pwb_tiny<false, true>), dim3(p.N), dim3(64), 0, st, p);
    else hipLaunchKernelGGL((pwb_tiny<false, false>), dim3(p.N), dim3(64), 0, st, p);
    DL3P_CHECK_LAUNCH(fn);
    return DL3P_OK;
  }
  static const int dbg = getenv("DL3P_BF16_DBG") ? atoi(getenv("DL3P_BF16_DBG")) : 0;
  p.dbg = dbg;
  const bool stats = p.partials != nullptr;
  static const int no_stream = getenv("DL3P_BF16_NOSTREAM") ? atoi(getenv("DL3P_BF16_NOSTREAM")) : 0;
  const StreamPlan sp = stream_plan_b(p.K, p.N, stats);
  // (the streaming kernel pays a [BN][K] weight load per workgroup and runs one workgroup of 8 waves per CU: it wins from
  // ~2^16 rows up -- 131072 x 304 -> 256: 67 us against 91 us tiled -- and loses on the 128 x 256 and 64 x 128 maps)
  if (sp.ok && !no_stream && p.M >= (1 << 16) && !(p.bb_z && sp.nt == 8)) {      // (128 columns + the backward sums spill)
    const int tiles = ceil_div(p.M, 32);
    const int per_cu = sp.lds > 75 * 1024 ? 1 : 2;
    int gx = ceil_div(tiles, 8 * 2);                       // at least two tiles per wave
    const int gx_max = (DL3P_NUM_CUS * per_cu) / sp.gy > 0 ? (DL3P_NUM_CUS * per_cu) / sp.gy : 1;
    if (gx > gx_max) gx = gx_max;
    if (gx < 1) gx = 1;
    if (rows_out) *rows_out = gx;
    const dim3 grid(gx, sp.gy);
    if (a_f32) launch_stream<true, false, false>(p, sp, grid, tiles, st);
    else if (y_f32) launch_stream<false, true, false>(p, sp, grid, tiles, st);
    else if (stats && p.bb_z) launch_stream<false, false, true, true>(p, sp, grid, tiles, st);
    else if (stats) launch_stream<false, false, true>(p, sp, grid, tiles, st);
    else launch_stream<false, false, false>(p, sp, grid, tiles, st);
    DL3P_CHECK_LAUNCH(fn);
    return DL3P_OK;
  }
  int nt = pick_nt_b(p.N, stats);
  int gy = ceil_div(p.N, 16 * nt);
  // 64-row tiles whenever 128-row tiles would leave the chip under two workgroups per CU (the 64 x 128 maps: M = 8192)
  int mi = (nt == 16 || (long long)ceil_div(p.M, 128) * gy < 2LL * DL3P_NUM_CUS) ? 1 : 2;
  static const int force_mi = getenv("DL3P_BF16_MI") ? atoi(getenv("DL3P_BF16_MI")) : 0;     // A/B knob
  if (force_mi && nt != 16) mi = force_mi;
  // few row tiles and a long reduction: K groups inside the workgroup (pwb_gemm, KG > 1)
  static const int env_kg = getenv("DL3P_BF16_KG") ? atoi(getenv("DL3P_BF16_KG")) : 0;       // 0: the rule; 1: never; 2 / 4: wherever possible
  const int force_kg = dl3p_bf16_force_kg >= 0 ? dl3p_bf16_force_kg : env_kg;                 // dl3p_set_option("bf16_kg", ...) first
  static const int force_kg_nt = getenv("DL3P_BF16_KG_NT") ? atoi(getenv("DL3P_BF16_KG_NT")) : 0;
  int kgroups = 1;
  if (force_kg != 1 && !a_f32 && !y_f32 && p.N >= 32) {       // (with or without the backward sums: the two data gradients stay bit-identical)
    const int nk = ceil_div(p.K, BK);
    int knt = nt > 8 ? 8 : nt;
    if (force_kg_nt) knt = force_kg_nt;
    const long long wgs = (long long)ceil_div(p.M, 64) * ceil_div(p.N, 16 * knt);
    if (force_kg > 1) kgroups = force_kg;
    else if (wgs <= 2LL * DL3P_NUM_CUS && nk >= 8) kgroups = 2;
    if (kgroups == 4 && knt > 4) knt = 4;       // sixteen waves: 128 VGPRs each
    if (kgroups > 1) { nt = knt; gy = ceil_div(p.N, 16 * nt); mi = 1; }
  }
  p.num_m_tiles = ceil_div(p.M, 64 * mi);
  int gx_max = (DL3P_NUM_CUS * 4) / gy;
  if (gx_max < 8) gx_max = 8;
  if (gx_max > DL3P_MAX_STAT_ROWS) gx_max = DL3P_MAX_STAT_ROWS;
  int gx = p.num_m_tiles;
  if (gx > gx_max) gx = ceil_div(p.num_m_tiles, ceil_div(p.num_m_tiles, gx_max));
  if (rows_out) *rows_out = gx;
  const dim3 grid(gx, gy);
#define DL3P_GB(MIV)                                                                \
  do {                                                                              \
    if (a_f32) launch_gemm_b<MIV, true, false, false>(p, nt, grid, st);             \
    else if (y_f32) launch_gemm_b<MIV, false, true, false>(p, nt, grid, st);        \
    else if (stats && p.bb_z) launch_gemm_b<MIV, false, false, true, true>(p, nt, grid, st); \
    else if (stats) launch_gemm_b<MIV, false, false, true>(p, nt, grid, st);        \
    else launch_gemm_b<MIV, false, false, false>(p, nt, grid, st);                  \
  } while (0)
  if (kgroups > 1) {
#define DL3P_GK(NTV, KGV)                                                                                                          \
    do {                                                                                                                           \
      if (stats && p.bb_z) hipLaunchKernelGGL((pwb_gemm<NTV, 1, false, false, true, true, KGV>), grid, dim3(256 * KGV), 0, st, p); \
      else if (stats) hipLaunchKernelGGL((pwb_gemm<NTV, 1, false, false, true, false, KGV>), grid, dim3(256 * KGV), 0, st, p);     \
      else hipLaunchKernelGGL((pwb_gemm<NTV, 1, false, false, false, false, KGV>), grid, dim3(256 * KGV), 0, st, p);              \
    } while (0)
    if (kgroups == 4) { if (nt == 2) DL3P_GK(2, 4); else DL3P_GK(4, 4); }
    else { if (nt == 2) DL3P_GK(2, 2); else if (nt == 4) DL3P_GK(4, 2); else DL3P_GK(8, 2); }
#undef DL3P_GK
    DL3P_CHECK_LAUNCH(fn);
    return DL3P_OK;
  }
  if (mi == 1) DL3P_GB(1); else DL3P_GB(2);
#undef DL3P_GB
  DL3P_CHECK_LAUNCH(fn);
  return DL3P_OK;
}

struct WgradPlan { int nw, ktiles, ntiles, mchunks, mrows; };
WgradPlan wgrad_plan_b(int M, int K, int N) {
  WgradPlan pl;
  pl.nw = N > 64 ? 8 : 4;
  pl.ktiles = ceil_div(K, 64);
  pl.ntiles = ceil_div(N, 16 * pl.nw);
  const int tiles = pl.ktiles * pl.ntiles;
  int mchunks = (4 * DL3P_NUM_CUS) / tiles;
  const int most = ceil_div(M, 512);
  if (mchunks > most) mchunks = most;
  if (mchunks > 512) mchunks = 512;
  if (mchunks < 1) mchunks = 1;
  pl.mrows = ceil_div(ceil_div(M, mchunks), 128) * 128;
  pl.mchunks = ceil_div(M, pl.mrows);
  return pl;
}

}  // namespace

extern "C" int dl3p_pwconv_fwd_bf16(const void* x, int ldx, int x_is_f32, const float* in_scale, const float* in_shift,
                                    int in_act, const void* wt, const float* bias, void* y, int ldy, int y_is_f32,
                                    float* stat_partials, int* rows_out, int M, int K, int N, void* stream) {
  int rc = check_mat_b("dl3p_pwconv_fwd_bf16", x, ldx, K, x_is_f32);
  if (rc) return rc;
  rc = check_mat_b("dl3p_pwconv_fwd_bf16", y, ldy, N, y_is_f32);
  if (rc) return rc;
  DL3P_CHECK_ARG(wt && aligned16(wt) && M > 0, "dl3p_pwconv_fwd_bf16: bad arguments");
  DL3P_CHECK_ARG(!stat_partials || !(x_is_f32 || y_is_f32), "dl3p_pwconv_fwd_bf16: statistics need bf16 in and out");
  GemmB p = {};
  p.A = x; p.lda = ldx; p.scale = in_scale; p.shift = in_shift; p.act = in_act;
  p.B = (const bf16*)wt; p.ldb = K; p.bias = bias; p.Y = y; p.ldy = ldy; p.partials = stat_partials;
  p.M = M; p.K = K; p.N = N;
  return gemm_b("dl3p_pwconv_fwd_bf16", p, x_is_f32 != 0, y_is_f32 != 0, rows_out, (hipStream_t)stream);
}

extern "C" int dl3p_pwconv_bwd_data_bf16(const void* dy, int lddy, int dy_is_f32, const void* w, void* gx, int ldgx,
                                         int accumulate, int M, int K, int N, void* stream) {
  int rc = check_mat_b("dl3p_pwconv_bwd_data_bf16", dy, lddy, N, dy_is_f32);
  if (rc) return rc;
  rc = check_mat_b("dl3p_pwconv_bwd_data_bf16", gx, ldgx, K, false);
  if (rc) return rc;
  DL3P_CHECK_ARG(w && aligned16(w) && M > 0, "dl3p_pwconv_bwd_data_bf16: bad arguments");
  GemmB p = {};
  p.A = dy; p.lda = lddy; p.act = DL3P_ACT_NONE;
  p.B = (const bf16*)w; p.ldb = N;          // W[K][N]: output column k, reduction n contiguous
  p.Y = gx; p.ldy = ldgx; p.accumulate = accumulate;
  p.M = M; p.K = N; p.N = K;
  return gemm_b("dl3p_pwconv_bwd_data_bf16", p, dy_is_f32 != 0, false, nullptr, (hipStream_t)stream);
}

// data gradient of a layer whose input is act(BN(z)) + the BatchNorm-backward partial sums of that BN (bf16 gradient and z;
// the separate dl3p_bn_bwd_reduce_bf16 pass over them disappears).  More than 64 rows: the few-row kernel carries no sums.
extern "C" int dl3p_pwconv_bwd_data_bn_bf16(const void* dy, int lddy, const void* w, void* gx, int ldgx, int accumulate,
                                            int M, int K, int N, const void* z, int ldz, const float* scale,
                                            const float* shift, int act, const float* save_mean,
                                            const float* save_invstd, float* partials, int* rows_out, void* stream) {
  const char* fn = "dl3p_pwconv_bwd_data_bn_bf16";
  int rc = check_mat_b(fn, dy, lddy, N, false);
  if (rc) return rc;
  rc = check_mat_b(fn, gx, ldgx, K, false);
  if (rc) return rc;
  rc = check_mat_b(fn, z, ldz, K, false);
  if (rc) return rc;
  DL3P_CHECK_ARG(w && aligned16(w) && M > 64 && scale && shift && save_mean && save_invstd && partials && rows_out && K % 4 == 0,
                 "%s: bad arguments (M > 64 rows, all BatchNorm vectors, K %% 4 == 0)", fn);
  GemmB p = {};
  p.A = dy; p.lda = lddy; p.act = DL3P_ACT_NONE;
  p.B = (const bf16*)w; p.ldb = N;
  p.Y = gx; p.ldy = ldgx; p.accumulate = accumulate;
  p.M = M; p.K = N; p.N = K;
  p.partials = partials;
  p.bb_z = (const bf16*)z; p.bb_ldz = ldz; p.bb_scale = scale; p.bb_shift = shift; p.bb_mean = save_mean;
  p.bb_invstd = save_invstd; p.bb_act = act;
  return gemm_b(fn, p, false, false, rows_out, (hipStream_t)stream);
}

extern "C" size_t dl3p_pwconv_bwd_weight_workspace_bf16(int M, int K, int N) {
  if (M <= 0 || K <= 0 || N <= 0) return 0;
  const WgradPlan pl = wgrad_plan_b(M, K, N);
  return ((size_t)pl.mchunks * K * N + (size_t)DL3P_NUM_CUS * 2 * N) * sizeof(float);
}

static int pwb_bwd_weight_impl(const void* x, int ldx, const float* in_scale, const float* in_shift, int in_act,
                               const void* dy, int lddy, int dy_is_f32, float* gw, float* gb, float* workspace,
                               size_t workspace_bytes, int M, int K, int N, int* rows_out, void* stream) {
  int rc = check_mat_b("dl3p_pwconv_bwd_weight_bf16", x, ldx, K, false);
  if (rc) return rc;
  rc = check_mat_b("dl3p_pwconv_bwd_weight_bf16", dy, lddy, N, dy_is_f32);
  if (rc) return rc;
  DL3P_CHECK_ARG((gw || rows_out) && M > 0, "dl3p_pwconv_bwd_weight_bf16: bad arguments");
  DL3P_CHECK_ARG(!rows_out || (!gb && M > 64),
                 "dl3p_pwconv_bwd_weight_slabs_bf16: no bias gradient and more than 64 rows (use dl3p_pwconv_bwd_weight_bf16)");
  hipStream_t st = (hipStream_t)stream;
  WgradB p = {};
  p.X = (const bf16*)x; p.ldx = ldx; p.scale = in_scale; p.shift = in_shift; p.act = in_act; p.DY = dy; p.lddy = lddy;
  p.M = M; p.K = K; p.N = N;
  if (M <= 64) {
    long long blocks = ceil_div_ll((long long)K * N, 256);
    if (blocks > 4096) blocks = 4096;
    if (dy_is_f32) hipLaunchKernelGGL((pwb_tiny_wgrad<true>), dim3((unsigned)blocks), dim3(256), 0, st, p, gw, gb);
    else hipLaunchKernelGGL((pwb_tiny_wgrad<false>), dim3((unsigned)blocks), dim3(256), 0, st, p, gw, gb);
    DL3P_CHECK_LAUNCH("dl3p_pwconv_bwd_weight_bf16");
    return DL3P_OK;
  }
  DL3P_CHECK_ARG(workspace && workspace_bytes >= dl3p_pwconv_bwd_weight_workspace_bf16(M, K, N),
                 "dl3p_pwconv_bwd_weight_bf16: workspace too small");
  const WgradPlan pl = wgrad_plan_b(M, K, N);
  p.slabs = workspace; p.ktiles = pl.ktiles; p.ntiles = pl.ntiles; p.mrows = pl.mrows;
  const dim3 grid(pl.ktiles * pl.ntiles, pl.mchunks);
  if (pl.nw == 8) {
    if (dy_is_f32) hipLaunchKernelGGL((pwb_wgrad<8, true>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((pwb_wgrad<8, false>), grid, dim3(256), 0, st, p);
  } else {
    if (dy_is_f32) hipLaunchKernelGGL((pwb_wgrad<4, true>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((pwb_wgrad<4, false>), grid, dim3(256), 0, st, p);
  }
  DL3P_CHECK_LAUNCH("dl3p_pwconv_bwd_weight_bf16");
  if (rows_out) {                      // the caller reduces the slabs (dl3p_reduce_rows_batched)
    *rows_out = pl.mchunks;
    return DL3P_OK;
  }
  rc = dl3p_reduce_rows_impl(workspace, pl.mchunks, (size_t)K * N, gw, 0, st);
  if (rc) return rc;
  if (gb) {
    float* part = workspace + (size_t)pl.mchunks * K * N;
    const int blocks = DL3P_NUM_CUS * 2;
    if (dy_is_f32) hipLaunchKernelGGL((pwb_colsum<true>), dim3(blocks), dim3(256), 0, st, dy, lddy, M, N, part);
    else hipLaunchKernelGGL((pwb_colsum<false>), dim3(blocks), dim3(256), 0, st, dy, lddy, M, N, part);
    DL3P_CHECK_LAUNCH("dl3p_pwconv_bwd_weight_bf16");
    rc = dl3p_reduce_rows_impl(part, blocks, (size_t)N, gb, 0, st);
  }
  return rc;
}

extern "C" int dl3p_pwconv_bwd_weight_bf16(const void* x, int ldx, const float* in_scale, const float* in_shift,
                                           int in_act, const void* dy, int lddy, int dy_is_f32, float* gw, float* gb,
                                           float* workspace, size_t workspace_bytes, int M, int K, int N, void* stream) {
  return pwb_bwd_weight_impl(x, ldx, in_scale, in_shift, in_act, dy, lddy, dy_is_f32, gw, gb, workspace, workspace_bytes, M, K,
                             N, nullptr, stream);
}

extern "C" int dl3p_pwconv_bwd_weight_slabs_bf16(const void* x, int ldx, const float* in_scale, const float* in_shift,
                                                 int in_act, const void* dy, int lddy, int dy_is_f32, float* workspace,
                                                 size_t workspace_bytes, int* rows_out, int M, int K, int N, void* stream) {
  DL3P_CHECK_ARG(rows_out != nullptr, "dl3p_pwconv_bwd_weight_slabs_bf16: rows_out is required");
  return pwb_bwd_weight_impl(x, ldx, in_scale, in_shift, in_act, dy, lddy, dy_is_f32, nullptr, nullptr, workspace,
                             workspace_bytes, M, K, N, rows_out, stream);
}
