// Pointwise (1x1) convolution = row-major GEMM on the MFMA f32 path (v_mfma_f32_16x16x4_f32) for gfx950.
//
// Replaces Conv2D(filters,(1,1)) at /root/reference deeplabv3p/models/layers.py:105,134,141,157,209,
// deeplabv3p_mobilenetv2.py:47,63, deeplabv3p_xception.py:81, deeplabv3p/model.py:75 (93-98 % of the
// model's MACs).
//
//   fwd    Y[M,N]  = act(X[M,K]*scale+shift) @ W[K,N] (+bias)      + per-channel (sum, sum^2) partials
//   dgrad  GX[M,K] (+)= DY[M,N] @ W[K,N]^T
//   wgrad  GW[K,N] = act(X*scale+shift)^T @ DY                      (split over M, slab reduce)
//
// fp32 in / fp32 accumulate is exact f32 (bitwise an fmaf chain), which the 1e-3 parity budget needs;
// it runs at 64 FLOP/clk/SIMD so the kernels are MFMA-issue bound and LDS traffic is negligible
// (one 16-B fragment read feeds four MFMAs).  Tiling is for 64-wide waves: a workgroup = 4 waves, each
// wave owns 32 rows x (16*NT) columns as 2 x NT accumulators of 16x16.  The k index inside a
// 16-deep group is permuted (lane quarter q supplies k = 4q+j at step j) so that every A fragment is ONE
// ds_read_b128; both operands use the same permutation so the sum is unchanged.
// Operands are swapped in the MFMA (D = W^T-frag x X-frag) so a lane ends up with 4 CONSECUTIVE output
// channels of one pixel: the epilogue is a single 16-B store per accumulator and the BN statistics are
// a 16-lane shuffle reduction.  Workgroups are persistent over M tiles: statistics stay in registers
// and leave as one partial row per workgroup (deterministic, no atomics); the global->LDS staging of
// the next tile is issued before the current tile's MFMAs (register double buffering) and applies the
// producer's BN+activation on the way in, so normalised activations are never materialised in HBM.
#include "common.h"
#include <limits.h>
#include <type_traits>
#include <string.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define BK 32
#define APITCH 36  // 32 + 4: rows 4 apart land 16 banks apart -> ds_read_b128 conflict-free

#include "pw_gemm.h"

// B_KN: B is [Kred][Nout] row-major (forward: the Keras kernel as stored);
// !B_KN: B is [Nout][Kred] row-major (dgrad: the same kernel read as its transpose).
#ifndef DL3P_GEMM_PIN_B
#define DL3P_GEMM_PIN_B 1      // 0 builds the unpinned loop for A/B runs (scripts/micro/build_variant.sh)
#endif
template <int NT, bool B_KN, bool STATS, int MI, int BKT, bool BNB = false, bool GA = false>
__global__ __launch_bounds__(256, 2) void pw_gemm_kernel(GemmParams p_in) {
  GemmParams p = p_in;
  int block_y = blockIdx.y;
#ifndef DL3P_NO_SPLITK
  if constexpr (!B_KN && !STATS && !BNB && !GA) {
    if (p.ksplit > 1) {                     // split-K forward: this workgroup's slice of the reduction and its slab of the output
      const int nbn = (int)gridDim.y / p.ksplit;
      const int z = block_y / nbn;
      block_y -= z * nbn;
      const int k_lo = z * p.kchunk;
      p.A += k_lo;
      if (p.scale) { p.scale += k_lo; p.shift += k_lo; }
      p.B += k_lo;
      p.K = min(p.kchunk, p.K - k_lo);
      p.Y += (size_t)z * p.M * p.ldy;
      p.bias = nullptr;
    }
  }
#endif
  constexpr int AP = BKT + 4;   // A pitch: rows 4 apart land 16 banks apart -> ds_read_b128 conflict-free
  constexpr int KQ = BKT / 4;   // float4 per K-tile row
  constexpr int RP = 256 / KQ;  // A rows staged per pass of the 256 threads
  constexpr int NA = (64 * MI) / RP;
  constexpr int BM = 64 * MI;   // 4 waves x MI tiles of 16 rows
  constexpr int BN = 16 * NT;
  constexpr int BPITCH = B_KN ? (BN + 4) : AP;
  constexpr int BS_FLOATS = B_KN ? BKT * BPITCH : BN * AP;
  constexpr int NB4 = (KQ * BN + 255) / 256;  // float4 per thread for the B tile
  // epilogue transpose buffer (wave-private slices): accumulators go out as whole 256-B row segments
  constexpr int TPP = NT < 4 ? NT : 4;            // 16-column tiles per epilogue pass
  constexpr int NPASS = (NT + TPP - 1) / TPP;
  constexpr int CH = 16 * TPP;
  constexpr int EPITCH = CH + 4;
  constexpr int RW = 16 * MI;                      // rows per wave
  // one dynamic LDS object.  BKT = 64: the epilogue buffer overlays the operand tiles (one extra barrier per
  // M tile) so that two workgroups still fit a CU
  constexpr int AS_FLOATS = BM * AP;
  constexpr int ES_FLOATS = 4 * RW * EPITCH;
  constexpr int OPER_FLOATS = AS_FLOATS + BS_FLOATS;
  constexpr bool OVERLAY = BKT > 32;
  constexpr int RED_OFF = OVERLAY ? (OPER_FLOATS > ES_FLOATS ? OPER_FLOATS : ES_FLOATS) : OPER_FLOATS + ES_FLOATS;
  extern __shared__ __attribute__((aligned(16))) float g_lds[];
  float* As = g_lds;
  float* Bs = g_lds + AS_FLOATS;
  float* Es = OVERLAY ? g_lds : g_lds + OPER_FLOATS;
  float* red = g_lds + RED_OFF;

  const int t = threadIdx.x;
  const int l = t & 63;
  const int w = t >> 6;
  const int l15 = l & 15;
  const int q = l >> 4;
  const int n0 = block_y * BN;
  const int nk = (p.K + BKT - 1) / BKT;
  const int my_tiles = (p.num_m_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  const int it_total = my_tiles * nk;

  // staging roles.  Every global load of the K loop is UNCONDITIONAL on a clamped 32-bit byte offset
  // (rows >= M re-read row M-1, columns >= K re-read the last float4) and invalid lanes are zeroed by a
  // select when the tile is written to LDS: no exec-masked branch per load (13 of them per K-step
  // before), one v_add per address.  Hosts reject operands of 4 GiB or more.
  const int ar = t / KQ;         // A row within a pass of RP rows
  const int akq = (t % KQ) * 4;  // A k offset within the K tile
  const char* Ab = reinterpret_cast<const char*>(p.A);
  const char* Bb = reinterpret_cast<const char*>(p.B);

  float4 ra[NA];
  float4 rb[NB4];
  float4 rsc = make_float4(1.f, 1.f, 1.f, 1.f), rsh = zero4();
  uint32_t a_row[NA];        // byte offset of this thread's A rows in the current M tile
  int g_by[GA ? NA : 1], g_bx[GA ? NA : 1];   // GA: source coordinates of tap (0, 0) of this thread's rows
  uint32_t g_ok = 0;                          // GA: which of the NA loads in flight hit the source tensor
  uint32_t b_off[NB4];           // byte offset of this thread's B float4s at k0 = 0
  bool b_nok[NB4];               // column (B_KN) / row (!B_KN) of the B tile inside the matrix
  int pf_m0 = -1;
#pragma unroll
  for (int i = 0; i < NB4; ++i) {
    const int idx = min(t + 256 * i, KQ * BN - 1);
    if (B_KN) {
      const int kk = idx / (BN / 4), nq = idx - kk * (BN / 4);
      const int n = n0 + nq * 4;
      b_nok[i] = (t + 256 * i < KQ * BN) && n < p.N;
      b_off[i] = (uint32_t)(min(n, p.N - 4)) * 4u;    // + k * ldb * 4 per K-step
    } else {
      const int r = idx / KQ;
      const int n = n0 + r;
      b_nok[i] = (t + 256 * i < KQ * BN) && n < p.N;
      b_off[i] = (uint32_t)min(n, p.N - 1) * (uint32_t)p.ldb * 4u;   // + k * 4 per K-step
    }
  }

  auto prefetch = [&](int it) {
    const int kt = it % nk;
    const int mt = blockIdx.x + (it / nk) * gridDim.x;
    const int m0 = mt * BM;
    const int k0 = kt * BKT;
    if (GA) {
      if (m0 != pf_m0) {
        pf_m0 = m0;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
          const int m = m0 + ar + RP * i;
          const int mc = min(m, p.M - 1);
          const int row = mc / p.g_RW, x = mc - row * p.g_RW;
          const int n = row / p.g_RH, y = row - n * p.g_RH;
          // rows past M get a base far outside the source: every tap fails the bounds check
          g_by[i] = m < p.M ? y * p.g_mul + p.g_ay : -(1 << 20);
          g_bx[i] = x * p.g_mul + p.g_ax;
          a_row[i] = (uint32_t)n * (uint32_t)(p.g_SH * p.g_SW);      // pixel index of the image in the source
        }
      }
      const int k = min(k0 + akq, p.K - 4);
      const int tap = (int)__umulhi((uint32_t)k, p.g_cmagic);
      const int c = k - tap * p.g_C;
      const int ky = (tap * p.g_kwmagic) >> 16, kx = tap - ky * p.g_kw;
      const int dyo = ky * p.g_d, dxo = kx * p.g_d;
      const int par = (1 << p.g_shift) - 1;
      g_ok = 0;
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const int ty = g_by[i] + dyo, tx = g_bx[i] + dxo;
        const int sy = ty >> p.g_shift, sx = tx >> p.g_shift;
        const bool ok = ty >= 0 && tx >= 0 && ((ty | tx) & par) == 0 && sy < p.g_SH && sx < p.g_SW && k0 + akq < p.K;
        const uint32_t off = ok ? ((a_row[i] + (uint32_t)(sy * p.g_SW + sx)) * (uint32_t)p.lda + (uint32_t)c) * 4u : 0u;
        ra[i] = *reinterpret_cast<const float4*>(Ab + off);
        g_ok |= ok ? (1u << i) : 0u;
      }
      if (p.scale) {
        rsc = *reinterpret_cast<const float4*>(p.scale + c);
        rsh = *reinterpret_cast<const float4*>(p.shift + c);
      }
    } else {
    if (m0 != pf_m0) {
      pf_m0 = m0;
#pragma unroll
      for (int i = 0; i < NA; ++i) a_row[i] = (uint32_t)min(m0 + ar + RP * i, p.M - 1) * (uint32_t)p.lda * 4u;
    }
    const uint32_t kb = (uint32_t)min(k0 + akq, p.K - 4) * 4u;
#pragma unroll
    for (int i = 0; i < NA; ++i) ra[i] = *reinterpret_cast<const float4*>(Ab + (a_row[i] + kb));
    if (p.scale) {
      rsc = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(p.scale) + kb);
      rsh = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(p.shift) + kb);
    }
    }
#pragma unroll
    for (int i = 0; i < NB4; ++i) {
      const int idx = min(t + 256 * i, KQ * BN - 1);
      if (B_KN) {
        const int kk = idx / (BN / 4);
        rb[i] = *reinterpret_cast<const float4*>(Bb + (b_off[i] + (uint32_t)min(k0 + kk, p.K - 1) * (uint32_t)p.ldb * 4u));
      } else {
        const int kq = (idx % KQ) * 4;
        rb[i] = *reinterpret_cast<const float4*>(Bb + (b_off[i] + (uint32_t)min(k0 + kq, p.K - 4) * 4u));
      }
    }
  };

  // producer's BatchNormalization + activation on the way into LDS.  relu / relu6 / none are one
  // fma + one v_med3 per element; the hard-swish family takes the general form (wave-uniform branch).
  const float act_lo = p.act == DL3P_ACT_NONE ? -DL3P_INF : 0.f;
  const float act_hi = (p.act == DL3P_ACT_NONE || p.act == DL3P_ACT_RELU) ? DL3P_INF : 6.f;
  auto prologue4 = [&](float4 v) {
    v = fma4(v, rsc, rsh);
    if (p.act >= DL3P_ACT_HSWISH) return act_apply4(v, p.act);
    return make_float4(__builtin_amdgcn_fmed3f(v.x, act_lo, act_hi), __builtin_amdgcn_fmed3f(v.y, act_lo, act_hi),
                       __builtin_amdgcn_fmed3f(v.z, act_lo, act_hi), __builtin_amdgcn_fmed3f(v.w, act_lo, act_hi));
  };

  const bool has_pro = p.scale != nullptr || p.act != DL3P_ACT_NONE;   // data gradient / im2col input: raw operand
  const bool n_edge = n0 + BN > p.N;
  auto stage = [&](int it) {
    const int kt = it % nk;
    const int mt = blockIdx.x + (it / nk) * gridDim.x;
    const int m0 = mt * BM;
    const int k0 = kt * BKT;
    // interior K-steps (the common case, wave-uniform) skip the zero-fill selects of the M / K / N tails
    const bool a_edge = m0 + BM > p.M || k0 + BKT > p.K;
    const bool b_edge = n_edge || k0 + BKT > p.K;
    const bool kok = k0 + akq < p.K;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int r = ar + RP * i;
      float4 v = ra[i];
      if (has_pro) v = prologue4(v);
      // zero rows/cols stay exactly zero (padding of the M and K tails; GA: taps outside the source)
      if (GA) v = ((g_ok >> i) & 1u) ? v : zero4();
      else if (a_edge) v = (kok && m0 + r < p.M) ? v : zero4();
      *reinterpret_cast<float4*>(&As[r * AP + akq]) = v;
    }
#pragma unroll
    for (int i = 0; i < NB4; ++i) {
      const int idx = t + 256 * i;
#if DL3P_GEMM_PIN_B
      // An empty asm that reads rb[i] right before its LDS store.  Without it clang hoists the B stores' address math
      // and, in 64 of the 80 instantiations, ends up with an s_waitcnt vmcnt(5)/(6) INSIDE the next K-step's prefetch
      // burst (right after the barrier): every wave then sits out a full memory latency before its first MFMA.  With
      // the pin none of the 80 has that wait; the decoder GEMMs run 4-12 % faster ("stage phase").
      asm volatile("" :: "v"(rb[i].x), "v"(rb[i].y), "v"(rb[i].z), "v"(rb[i].w));
#endif
      if (idx < KQ * BN) {
        if (B_KN) {
          const int kk = idx / (BN / 4), nq = idx - kk * (BN / 4);
          float4 v = make_float4(rb[i].x, rb[i].y, rb[i].z, rb[i].w);
          if (b_edge) v = (b_nok[i] && k0 + kk < p.K) ? v : zero4();
          *reinterpret_cast<float4*>(&Bs[kk * BPITCH + nq * 4]) = v;
        } else {
          const int r = idx / KQ, kq = (idx % KQ) * 4;
          float4 v = make_float4(rb[i].x, rb[i].y, rb[i].z, rb[i].w);
          if (b_edge) v = (b_nok[i] && k0 + kq < p.K) ? v : zero4();
          *reinterpret_cast<float4*>(&Bs[r * AP + kq]) = v;
        }
      }
    }
  };

  f32x4 acc[MI][NT];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float4 st_s[STATS ? NPASS : 1], st_q[STATS ? NPASS : 1];   // per lane: 4 columns of each epilogue pass
  if (STATS) {
#pragma unroll
    for (int i = 0; i < NPASS; ++i) { st_s[i] = zero4(); st_q[i] = zero4(); }
  }

  if (it_total > 0) prefetch(0);
  for (int it = 0; it < it_total; ++it) {
    stage(it);
    __syncthreads();
    if (it + 1 < it_total) prefetch(it + 1);
#pragma unroll
    for (int g = 0; g < BKT / 16; ++g) {
      const int kc = g * 16 + q * 4;
      float4 a[MI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
        a[mi] = *reinterpret_cast<const float4*>(&As[(w * 16 * MI + mi * 16 + l15) * AP + kc]);
#pragma unroll
      for (int ni = 0; ni < NT; ++ni) {
        float b[4];
        if (B_KN) {
#pragma unroll
          for (int j = 0; j < 4; ++j) b[j] = Bs[(kc + j) * BPITCH + ni * 16 + l15];
        } else {
          const float4 bv = *reinterpret_cast<const float4*>(&Bs[(ni * 16 + l15) * AP + kc]);
          b[0] = bv.x; b[1] = bv.y; b[2] = bv.z; b[3] = bv.w;
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[0], a[mi].x, acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[1], a[mi].y, acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[2], a[mi].z, acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[3], a[mi].w, acc[mi][ni], 0, 0, 0);
        }
      }
    }
    __syncthreads();
    if (it % nk == nk - 1) {
      // epilogue of this M tile.  After the MFMAs a lane holds 4 consecutive channels of pixel l15 per
      // accumulator; stored directly that is 16 rows x 64 B per store instruction (half cache lines,
      // measured: 13.6k cycles per tile, and the next tile's staging waits behind those stores).  The tile
      // is therefore transposed through a wave-private LDS slice and leaves as 4 rows x 256 B per
      // instruction; bias / accumulate / BN statistics are applied on the way out.
      const int mt = blockIdx.x + (it / nk) * gridDim.x;
      const int m0 = mt * BM;
      float* es = Es + w * RW * EPITCH;
      const int rr = l >> 4, cq = l & 15;
#pragma unroll
      for (int ps = 0; ps < NPASS; ++ps) {
        const int ni0 = ps * TPP;
#pragma unroll
        for (int nl = 0; nl < TPP; ++nl) {
          if (ni0 + nl < NT) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
              const f32x4 v = acc[mi][ni0 + nl];
              acc[mi][ni0 + nl] = (f32x4){0.f, 0.f, 0.f, 0.f};
              *reinterpret_cast<float4*>(&es[(mi * 16 + l15) * EPITCH + nl * 16 + q * 4]) = make_float4(v[0], v[1], v[2], v[3]);
            }
          }
        }
        const int n = n0 + ni0 * 16 + cq * 4;
        const bool col_ok = (ni0 * 16 + cq * 4 < BN) && (cq * 4 < CH) && n < p.N && (ni0 + cq / 4 < NT);
        const int row_lim = p.M - (m0 + w * RW);     // valid rows of this wave's slice (wave-uniform)
        if (col_ok) {
          float4 bias4 = zero4();
          if (p.bias) bias4 = ld4(p.bias + n);
          char* yb = reinterpret_cast<char*>(p.Y) + ((uint32_t)(m0 + w * RW + rr) * (uint32_t)p.ldy + (uint32_t)n) * 4u;
          const uint32_t ystep = (uint32_t)p.ldy * 16u;   // 4 rows
          // fused BN-backward statistics: per-channel constants of this lane's 4 columns
          constexpr bool bnb = STATS && BNB;    // (a template flag: the z prefetch registers must not burden the forward)
          float4 bsc = zero4(), bsh = zero4(), bmu = zero4(), bis = zero4();
          float4 zpre[RW / 4];          // all z rows of the pass are requested before the first one is used
          if (bnb) {
            bsc = ld4(p.bb_scale + n); bsh = ld4(p.bb_shift + n); bmu = ld4(p.bb_mean + n); bis = ld4(p.bb_invstd + n);
            const char* zbase = reinterpret_cast<const char*>(p.bb_z) + (uint32_t)n * 4u;
#pragma unroll
            for (int i = 0; i < RW / 4; ++i) {
              const int mrow = min(m0 + w * RW + 4 * i + rr, p.M - 1);       // rows past M re-read the last one (unused)
              zpre[i] = *reinterpret_cast<const float4*>(zbase + (uint32_t)mrow * (uint32_t)p.bb_ldz * 4u);
            }
          }
          auto rows = [&](auto full) {
#pragma unroll
            for (int r0 = 0; r0 < RW; r0 += 4) {
              const int row = r0 + rr;
              if (decltype(full)::value || row < row_lim) {
                float4 o = add4(*reinterpret_cast<const float4*>(&es[row * EPITCH + cq * 4]), bias4);
                float* yp = reinterpret_cast<float*>(yb + (r0 / 4) * ystep);
                if (p.accumulate) o = add4(o, ld4(yp));
                st4(yp, o);
                if (STATS) {
                  if (bnb) {
                    const float4 zv = zpre[r0 / 4];
                    const float4 u = fma4(zv, bsc, bsh);
                    const float4 d = make_float4(o.x * act_grad(u.x, p.bb_act), o.y * act_grad(u.y, p.bb_act),
                                                 o.z * act_grad(u.z, p.bb_act), o.w * act_grad(u.w, p.bb_act));
                    const float4 xh = make_float4((zv.x - bmu.x) * bis.x, (zv.y - bmu.y) * bis.y, (zv.z - bmu.z) * bis.z,
                                                  (zv.w - bmu.w) * bis.w);
                    st_s[ps] = add4(st_s[ps], d);
                    st_q[ps] = fma4(d, xh, st_q[ps]);
                  } else {
                    st_s[ps] = add4(st_s[ps], o);
                    st_q[ps] = fma4(o, o, st_q[ps]);
                  }
                }
              }
            }
          };
          if (row_lim >= RW) rows(std::true_type{});
          else rows(std::false_type{});
        }
      }
      if (OVERLAY) __syncthreads();   // the next stage() overwrites the epilogue buffer
    }
  }

  if (STATS) {
    // reduce over the 4 row groups of the wave, then over the 4 waves; one partial row per workgroup
    const int rr = l >> 4, cq = l & 15;
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      float sv[4] = {st_s[ps].x, st_s[ps].y, st_s[ps].z, st_s[ps].w};
      float qv[4] = {st_q[ps].x, st_q[ps].y, st_q[ps].z, st_q[ps].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float s1 = sv[e], s2 = qv[e];
        s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
        s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
        const int col = ps * CH + cq * 4 + e;
        if (rr == 0 && cq * 4 < CH && col < BN) {
          red[(0 * 4 + w) * BN + col] = s1;
          red[(1 * 4 + w) * BN + col] = s2;
        }
      }
    }
    __syncthreads();
    if (p.partials) {
      for (int i = t; i < 2 * BN; i += 256) {
        const int which = i / BN, nn = i - which * BN;
        if (n0 + nn < p.N) {
          float s = red[(which * 4 + 0) * BN + nn] + red[(which * 4 + 1) * BN + nn] +
                    red[(which * 4 + 2) * BN + nn] + red[(which * 4 + 3) * BN + nn];
          p.partials[((size_t)blockIdx.x * 2 + which) * p.N + n0 + nn] = s;
        }
      }
    }
  }
}

// (A barrier-enforced ping-pong of two half-workgroups -- 512 threads, waves 0-3 multiply while waves 4-7 stage and
// vice versa -- was built and measured twice this round: 510 us and 624 us against 471 us for two free-running
// workgroups per CU on 266256x304x256.  One wave per SIMD cannot keep the matrix pipe issuing back to back through
// its own LDS-read latencies; the free-running pair fills those bubbles.  Removed.)

// ------------------------------------------------------------------------------ forward / dgrad, small K x N
// Same idea as pw_wgrad_small_kernel for Y = act(X*scale+shift) @ W when the whole kernel matrix is a few
// KB (the 129x129 / 257x257 layers): W sits in LDS for the life of the workgroup, every wave walks its own
// 16-row tiles of M with no workgroup barrier, A fragments come straight from global memory (lane
// (row l15, quarter q) loads the float4 X[row][16 kt + 4q ..]: the k-permutation of the big kernel makes
// that exactly its MFMA operand), the 16 x N result is transposed through a wave-private LDS slice and
// leaves as whole rows; BN statistics are kept per lane in that row-major form and reduced once at the end.
template <int KT, int NTN, bool STATS, bool BNB = false>
__global__ __launch_bounds__(256, 2) void pw_small_kernel(GemmParams p) {
  constexpr int KP = 16 * KT, NP = 16 * NTN;
  constexpr int BP = NP + 4, TP = NP + 4;
  extern __shared__ __attribute__((aligned(16))) float sm_lds[];
  float* Bs = sm_lds;                         // [KP][BP], zero padded
  float* sc_s = Bs + KP * BP;
  float* sh_s = sc_s + KP;
  float* bq_s = sh_s + KP;                    // BNB: [scale | shift | mean | invstd][NP] of the BatchNorm whose sums ride along
  float* Tall = bq_s + (BNB ? 4 * NP : 0);    // 4 wave slices of [16][TP]
  const int t = threadIdx.x, l = t & 63, w = t >> 6, l15 = l & 15, q = l >> 4;
  float* Ts = Tall + w * 16 * TP;
  if (p.b_kn) {
    for (int idx = t; idx < KP * NP; idx += 256) {
      const int k = idx / NP, n = idx - k * NP;
      Bs[k * BP + n] = (k < p.K && n < p.N) ? p.B[(size_t)k * p.ldb + n] : 0.f;
    }
  } else {
    for (int idx = t; idx < KP * NP; idx += 256) {
      const int n = idx / KP, k = idx - n * KP;
      Bs[k * BP + n] = (k < p.K && n < p.N) ? p.B[(size_t)n * p.ldb + k] : 0.f;
    }
  }
  for (int i = t; i < KP; i += 256) {
    sc_s[i] = (p.scale && i < p.K) ? p.scale[i] : 1.f;
    sh_s[i] = (p.scale && i < p.K) ? p.shift[i] : 0.f;
  }
  if (BNB) {
    for (int i = t; i < NP; i += 256) {
      const bool in = i < p.N;
      bq_s[i] = in ? p.bb_scale[i] : 1.f;
      bq_s[NP + i] = in ? p.bb_shift[i] : 0.f;
      bq_s[2 * NP + i] = in ? p.bb_mean[i] : 0.f;
      bq_s[3 * NP + i] = in ? p.bb_invstd[i] : 0.f;
    }
  }
  __syncthreads();

  // A fragment loads: clamped 32-bit byte offsets, invalid lanes zeroed by select
  uint32_t a_k[KT];
  bool a_kok[KT];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) {
    const int k = kt * 16 + 4 * q;
    a_kok[kt] = k < p.K;
    a_k[kt] = (uint32_t)min(k, p.K - 4) * 4u;
  }
  // row-major output mapping of a 16 x N tile: float4 f = l + 64 i  ->  (row f / (N/4), column group f % (N/4))
  const int n4 = p.N >> 2, nf = 4 * p.N;
  int yrow[NTN], yl[NTN];
  uint32_t yg[NTN], zg[BNB ? NTN : 1];
#pragma unroll
  for (int i = 0; i < NTN; ++i) {
    const int f = min(l + 64 * i, nf - 1);
    const int r = f / n4, c = f - r * n4;
    yrow[i] = (l + 64 * i < nf) ? r : (1 << 20);
    yl[i] = r * TP + c * 4;
    yg[i] = ((uint32_t)r * (uint32_t)p.ldy + (uint32_t)c * 4u) * 4u;
    if (BNB) zg[i] = ((uint32_t)r * (uint32_t)p.bb_ldz + (uint32_t)c * 4u) * 4u;
  }
  const char* Zb = reinterpret_cast<const char*>(p.bb_z);
  const float act_lo = p.act == DL3P_ACT_NONE ? -DL3P_INF : 0.f;
  const float act_hi = (p.act == DL3P_ACT_NONE || p.act == DL3P_ACT_RELU) ? DL3P_INF : 6.f;

  f32x4 acc[NTN];
#pragma unroll
  for (int b = 0; b < NTN; ++b) acc[b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float4 st_s[STATS ? NTN : 1], st_q[STATS ? NTN : 1];
  if (STATS) {
#pragma unroll
    for (int i = 0; i < NTN; ++i) { st_s[i] = zero4(); st_q[i] = zero4(); }
  }

  const int ntiles = (p.M + 15) >> 4;
  const int nwaves = gridDim.x * 4;
  const int gw = blockIdx.x * 4 + w;
  const char* Ab = reinterpret_cast<const char*>(p.A);
  char* Yb = reinterpret_cast<char*>(p.Y);
  float4 ra[KT];
#define SM_PREFETCH(tile_)                                                                                   \
  {                                                                                                          \
    const uint32_t arow = (uint32_t)min(((tile_) << 4) + l15, p.M - 1) * (uint32_t)p.lda * 4u;              \
    _Pragma("unroll") for (int kt = 0; kt < KT; ++kt) ra[kt] = *reinterpret_cast<const float4*>(Ab + (arow + a_k[kt])); \
  }
  SM_PREFETCH(min(gw, ntiles - 1))
  for (int tile = gw; tile < ntiles; tile += nwaves) {
    const int m0 = tile << 4;
    const bool row_ok = m0 + l15 < p.M;
    // BNB: the z rows this lane's outputs meet are requested now and used after the MFMAs
    float4 zpre[BNB ? NTN : 1];
    if (BNB) {
      const uint32_t zbase = (uint32_t)m0 * (uint32_t)p.bb_ldz * 4u;
#pragma unroll
      for (int i = 0; i < NTN; ++i) {
        const uint32_t off = (yrow[i] < p.M - m0) ? zg[i] : 0u;        // rows past M re-read the tile's first row (unused)
        zpre[i] = *reinterpret_cast<const float4*>(Zb + (zbase + off));
      }
    }
    float4 a[KT];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      const float4 s4 = *reinterpret_cast<const float4*>(&sc_s[kt * 16 + 4 * q]);
      const float4 h4 = *reinterpret_cast<const float4*>(&sh_s[kt * 16 + 4 * q]);
      float4 v = fma4(ra[kt], s4, h4);
      if (p.act >= DL3P_ACT_HSWISH) v = act_apply4(v, p.act);
      else v = make_float4(__builtin_amdgcn_fmed3f(v.x, act_lo, act_hi), __builtin_amdgcn_fmed3f(v.y, act_lo, act_hi),
                           __builtin_amdgcn_fmed3f(v.z, act_lo, act_hi), __builtin_amdgcn_fmed3f(v.w, act_lo, act_hi));
      a[kt] = (row_ok && a_kok[kt]) ? v : zero4();
    }
    SM_PREFETCH(min(tile + nwaves, ntiles - 1))
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      const float av[4] = {a[kt].x, a[kt].y, a[kt].z, a[kt].w};
#pragma unroll
      for (int nt = 0; nt < NTN; ++nt)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(Bs[(kt * 16 + 4 * q + j) * BP + nt * 16 + l15], av[j], acc[nt], 0, 0, 0);
    }
    // lane holds 4 consecutive channels (nt*16 + 4q ..) of pixel l15 -> wave-private transpose -> whole rows
#pragma unroll
    for (int nt = 0; nt < NTN; ++nt) {
      *reinterpret_cast<float4*>(&Ts[l15 * TP + nt * 16 + 4 * q]) = make_float4(acc[nt][0], acc[nt][1], acc[nt][2], acc[nt][3]);
      acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const uint32_t ybase = (uint32_t)m0 * (uint32_t)p.ldy * 4u;
    const int rows_here = p.M - m0;
#pragma unroll
    for (int i = 0; i < NTN; ++i) {
      if (yrow[i] < rows_here) {
        float4 o = *reinterpret_cast<const float4*>(&Ts[yl[i]]);
        if (p.bias) o = add4(o, ld4(p.bias + (yl[i] - yrow[i] * TP)));
        float* yp = reinterpret_cast<float*>(Yb + (ybase + yg[i]));
        if (p.accumulate) o = add4(o, ld4(yp));
#ifdef DL3P_ABLATE_STORES
        if (o.x == 1234.5678f)
#endif
        st4(yp, o);
        if (STATS && BNB) {
          // (sum g', sum g' * xhat) of the BatchNorm behind this gradient, as dl3p_bn_bwd_reduce forms them
          const int cf = yl[i] - yrow[i] * TP;
          const float4 zv = zpre[i];
          const float4 u = fma4(zv, *reinterpret_cast<const float4*>(&bq_s[cf]), *reinterpret_cast<const float4*>(&bq_s[NP + cf]));
          const float4 mu = *reinterpret_cast<const float4*>(&bq_s[2 * NP + cf]), is = *reinterpret_cast<const float4*>(&bq_s[3 * NP + cf]);
          const float4 d = make_float4(o.x * act_grad(u.x, p.bb_act), o.y * act_grad(u.y, p.bb_act),
                                       o.z * act_grad(u.z, p.bb_act), o.w * act_grad(u.w, p.bb_act));
          const float4 xh = make_float4((zv.x - mu.x) * is.x, (zv.y - mu.y) * is.y, (zv.z - mu.z) * is.z, (zv.w - mu.w) * is.w);
          st_s[i] = add4(st_s[i], d);
          st_q[i] = fma4(d, xh, st_q[i]);
        } else if (STATS) {
          st_s[i] = add4(st_s[i], o);
          st_q[i] = fma4(o, o, st_q[i]);
        }
      }
    }
  }
#undef SM_PREFETCH
  if (STATS) {
    // per-lane sums are indexed by (row r, column group c) of the tile pattern: dump them and add the
    // 16 rows x 4 waves of every column group in a fixed order; one partial row per workgroup
    float4* S = reinterpret_cast<float4*>(Tall);
#pragma unroll
    for (int which = 0; which < 2; ++which) {
      __syncthreads();
#pragma unroll
      for (int i = 0; i < NTN; ++i)
        if (l + 64 * i < nf) S[w * 4 * NP + l + 64 * i] = which ? st_q[i] : st_s[i];
      __syncthreads();
      if (p.partials && t < n4) {
        float4 s = zero4();
        for (int ww = 0; ww < 4; ++ww)
          for (int r = 0; r < 16; ++r) s = add4(s, S[ww * 4 * NP + r * n4 + t]);
        st4(p.partials + ((size_t)blockIdx.x * 2 + which) * p.N + 4 * t, s);
      }
    }
  }
}

template <int KT, int NTN, bool BNB = false>
static constexpr size_t pw_small_lds() {
  return sizeof(float) * (size_t)(16 * KT * (16 * NTN + 4) + 2 * 16 * KT + (BNB ? 4 * 16 * NTN : 0) + 4 * 16 * (16 * NTN + 4));
}

template <int KT, int NTN, bool STATS, bool BNB = false>
static void launch_pw_small(const GemmParams& p, int grid, hipStream_t st) {
  constexpr size_t lds = pw_small_lds<KT, NTN, BNB>();
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)pw_small_kernel<KT, NTN, STATS, BNB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  dl3p_launch(pw_small_kernel<KT, NTN, STATS, BNB>, dim3(grid), dim3(256), lds, st, p);
}

struct SmallShape { int kt, ntn; };
// (reduction tiles, output tiles) instantiated for the forward / data-gradient small kernel
static bool pw_small_pick(int K, int N, SmallShape* out) {
  // measured against the tiled kernel (scripts/gemm_sweep.py): these shapes win 5-25 % at M >= 2^17 rows;
  // 2x2 (32x32), 2x16 / 16x2 and the 67600-row layers lose and stay on the tiled kernel
  static const SmallShape list[] = {{1, 2}, {2, 1}, {6, 1}, {1, 6}, {2, 3}, {3, 2}, {6, 2}, {2, 6}, {2, 9}, {9, 2}};
  static const int off = getenv("DL3P_PW_SMALL") ? atoi(getenv("DL3P_PW_SMALL")) == 0 : 0;
  if (off) return false;
  const int kt = ceil_div(K, 16), ntn = ceil_div(N, 16);
  int best = -1, best_tiles = 1 << 30;
  for (int i = 0; i < (int)(sizeof(list) / sizeof(list[0])); ++i)
    if (list[i].kt >= kt && list[i].ntn >= ntn && list[i].kt * list[i].ntn < best_tiles) { best = i; best_tiles = list[i].kt * list[i].ntn; }
  if (best < 0 || best_tiles > 2 * kt * ntn) return false;
  *out = list[best];
  return true;
}

// rows from which the streaming small-K.N kernels take over from the tiled kernel (production: 2^17).  The parity
// tests lower it through dl3p_set_option so that small test shapes reach those kernels too, and reset it for the
// tests that check the production dispatch at the production shapes.
static int g_gemm_force_nt = 0, g_gemm_force_mi = 0, g_gemm_force_pc = 0, g_gemm_use_table = -1;   // see gemm_tuned_lookup / gemm_plan
static int g_split_wgrad = -1;      // dl3p_set_option("split_wgrad", 0 | 1): weight gradients on the split-bf16 kernel (default DL3P_SPLIT_WGRAD, else DL3P_SPLIT_GEMM, else 1)
static int g_splitk_force = -1;        // "splitk": -1 the rule, 0 never, S > 0 that many slices where the shape is served (dl3p_pwconv_fwd_splitk_plan)
static int g_sbw_force_tile = -1, g_sbw_force_pc = 0;      // "split_wgrad_tile" (0..3, -1 none) / "split_wgrad_per_cu": pin its plan (and bypass the verdicts)
static int g_sb_pipe = -1;      // dl3p_set_option("sb_pipe", 0 | 1): the producer / consumer form of the split kernel (default DL3P_SB_PIPE or 0)
static int g_sb_force_wm = 0, g_sb_force_nt = 0;      // dl3p_set_option("sb_wm" / "sb_nt"): pin the split kernel's wide-tile family (gemm_plan_sb)
static int g_sb3 = -1;        // dl3p_set_option("sb3", 0 | 1 | -1): the pinned-schedule split forward (pw_split3.hip) never / wherever it serves the shape / by rule (DL3P_SB3)
static int g_sb_rs = -1;      // dl3p_set_option("sb_rs", 0 | 1 | -1): the row-stationary split kernel (pw_split_rs.hip) never / wherever it serves the shape / by rule (DL3P_SB_RS)
extern int dl3p_bf16_force_kg;      // pw_bf16.hip
static int g_conv_sb = -1;    // dl3p_set_option("conv_sb", 0 | 1 | 2 | -1): dense convs on the split kernels never / by rule / wherever supported / default (DL3P_CONV_SB, else 1)
static int g_wgrad_force_tile = -1, g_wgrad_force_per_cu = 0;                    // see wgrad_pick_tile / wgrad_split
static int g_pw_small_min_rows = -1;
static int pw_small_min_rows() {
  if (g_pw_small_min_rows < 0)
    g_pw_small_min_rows = getenv("DL3P_PW_SMALL_MIN_ROWS") ? atoi(getenv("DL3P_PW_SMALL_MIN_ROWS")) : (1 << 17);
  return g_pw_small_min_rows;
}
extern "C" int dl3p_set_option(const char* name, int value) {
  DL3P_CHECK_ARG(name != nullptr, "dl3p_set_option: null name");
  if (!strcmp(name, "pw_small_min_rows")) {
    g_pw_small_min_rows = value < 0 ? (1 << 17) : value;      // value < 0 restores the production threshold
    return DL3P_OK;
  }
  if (!strcmp(name, "gemm_nt")) { g_gemm_force_nt = (value >= 1 && value <= 8) ? value : 0; return DL3P_OK; }
  if (!strcmp(name, "gemm_mi")) { g_gemm_force_mi = (value == 1 || value == 2) ? value : 0; return DL3P_OK; }
  if (!strcmp(name, "gemm_per_cu")) { g_gemm_force_pc = value > 0 ? value : 0; return DL3P_OK; }
  if (!strcmp(name, "gemm_tuned")) { g_gemm_use_table = value ? 1 : 0; return DL3P_OK; }
  if (!strcmp(name, "sb_pipe")) { g_sb_pipe = value ? 1 : 0; return DL3P_OK; }
  if (!strcmp(name, "split_wgrad")) { g_split_wgrad = value ? 1 : 0; return DL3P_OK; }
  if (!strcmp(name, "split_wgrad_tile")) { g_sbw_force_tile = (value >= 0 && value <= 4) ? value : -1; return DL3P_OK; }
  if (!strcmp(name, "split_wgrad_per_cu")) { g_sbw_force_pc = value > 0 ? value : 0; return DL3P_OK; }
  if (!strcmp(name, "splitk")) { g_splitk_force = value; return DL3P_OK; }
  if (!strcmp(name, "sb_wm")) { g_sb_force_wm = (value >= -1 && value <= 2) ? value : 0; return DL3P_OK; }    // -1: never wide
  if (!strcmp(name, "sb_rs")) { g_sb_rs = value < 0 ? -1 : (value ? 1 : 0); return DL3P_OK; }
  if (!strcmp(name, "sb3")) { g_sb3 = value < 0 ? -1 : (value ? 1 : 0); return DL3P_OK; }
  if (!strcmp(name, "bf16_kg")) { dl3p_bf16_force_kg = (value == 0 || value == 1 || value == 2 || value == 4) ? value : -1; return DL3P_OK; }
  if (!strcmp(name, "conv_sb")) { g_conv_sb = (value >= 0 && value <= 2) ? value : -1; return DL3P_OK; }
  if (!strcmp(name, "sb_nt")) { g_sb_force_nt = (value == 8 || value == 12 || value == 16) ? value : 0; return DL3P_OK; }
  if (!strcmp(name, "dw_per_cu")) { dl3p_dw_force_per_cu = value > 0 ? value : 0; return DL3P_OK; }
  if (!strcmp(name, "dw_want")) { dl3p_dw_force_want = value > 0 ? value : 0; return DL3P_OK; }
  if (!strcmp(name, "dw_maxth")) { dl3p_dw_force_maxth = value > 0 ? value : 0; return DL3P_OK; }
  if (!strcmp(name, "dw_tw")) { dl3p_dw_force_tw = (value == 2 || value == 4) ? value : 0; return DL3P_OK; }
  if (!strcmp(name, "dw_tuned")) { dl3p_dw_use_table = value ? 1 : 0; return DL3P_OK; }
  if (!strcmp(name, "wgrad_tile")) { g_wgrad_force_tile = (value >= 0 && value <= 3) ? value : -1; return DL3P_OK; }
  if (!strcmp(name, "wgrad_per_cu")) { g_wgrad_force_per_cu = value > 0 ? value : 0; return DL3P_OK; }
  dl3p_set_error("dl3p_set_option: unknown option '%s'", name);
  return DL3P_EINVAL;
}
// the current value of the knobs that decide how many slabs / partial rows a traced launch writes ("split_wgrad", "conv_sb", "sb_rs",
// "sb_pipe"): an executor records them when it traces its plans and pins them again before an eager replay;
// INT_MIN for an unknown name
extern "C" int dl3p_get_option(const char* name) {
  if (!name) return INT_MIN;
  if (g_sb_pipe < 0) g_sb_pipe = getenv("DL3P_SB_PIPE") ? atoi(getenv("DL3P_SB_PIPE")) : 0;
  if (!strcmp(name, "split_wgrad")) return g_split_wgrad;
  if (!strcmp(name, "conv_sb")) return g_conv_sb;
  if (!strcmp(name, "sb_rs")) return g_sb_rs;
  if (!strcmp(name, "sb3")) return g_sb3;
  if (!strcmp(name, "sb_pipe")) return g_sb_pipe;
  if (!strcmp(name, "splitk")) return g_splitk_force;
  return INT_MIN;
}

static int pw_small_grid(int M) {
  static const int per_cu = getenv("DL3P_PW_SMALL_PER_CU") ? atoi(getenv("DL3P_PW_SMALL_PER_CU")) : 2;
  int g = ceil_div(M, 16) / (4 * 2);          // >= 2 row tiles per wave
  if (g > DL3P_NUM_CUS * per_cu) g = DL3P_NUM_CUS * per_cu;   // two resident workgroups per CU
  if (g > DL3P_MAX_STAT_ROWS) g = DL3P_MAX_STAT_ROWS;
  if (g < 1) g = 1;
  return g;
}

template <bool STATS, bool BNB = false>
static void launch_pw_small_any(const GemmParams& p, SmallShape sh, int grid, hipStream_t st) {
#define DL3P_PS(a, b) if (sh.kt == a && sh.ntn == b) { launch_pw_small<a, b, STATS, BNB>(p, grid, st); return; }
  DL3P_PS(1, 2) DL3P_PS(2, 1) DL3P_PS(6, 1) DL3P_PS(1, 6) DL3P_PS(2, 3) DL3P_PS(3, 2) DL3P_PS(6, 2) DL3P_PS(2, 6)
  DL3P_PS(2, 9) DL3P_PS(9, 2)
#undef DL3P_PS
}

// Measured tile choices for the GEMM shapes of the BASELINE graphs (scripts/tune_gemm.py writes gemm_tuned.h from timings
// on an MI355X; the heuristics below serve every other shape).  role: 0 forward, 1 forward + BatchNorm statistics,
// 2 data gradient, 3 data gradient + fused BatchNorm-backward sums; M rows, K reduction length, N output columns of the
// GEMM as launched.  dl3p_set_option("gemm_nt" / "gemm_mi", v) pins a choice (v = 0: automatic) -- that is how the
// tuner tries the candidates; ("gemm_tuned", 0) ignores the table.
struct GemmTuned { int role, M, K, N, nt, mi, pc; };   // pc: persistent workgroups per CU (0 = by tile width)
struct SbPays { int role, M, K, N, pays; };
#include "gemm_tuned.h"
#include "sb_tuned.h"
static const GemmTuned* gemm_tuned_lookup(int role, int M, int K, int N) {
  if (g_gemm_use_table < 0) g_gemm_use_table = getenv("DL3P_GEMM_TUNED") ? atoi(getenv("DL3P_GEMM_TUNED")) : 1;
  if (!g_gemm_use_table) return nullptr;
  if (role >= 5) {      // the split-bf16 kernel's rows (scripts/tune_split.py): role + 5
    for (size_t i = 0; i < sizeof(g_sb_tuned) / sizeof(g_sb_tuned[0]); ++i) {
      const GemmTuned& e = g_sb_tuned[i];
      if (e.role == role && e.M == M && e.K == K && e.N == N) return &e;
    }
    return nullptr;
  }
  for (size_t i = 0; i < sizeof(g_gemm_tuned) / sizeof(g_gemm_tuned[0]); ++i) {
    const GemmTuned& e = g_gemm_tuned[i];
    if (e.role == role && e.M == M && e.K == K && e.N == N) return &e;
  }
  return nullptr;
}

// choose the columns-per-workgroup (NT tiles of 16) that wastes the fewest MFMA columns
static int pick_nt(int N, int M) {
  const int ntiles = ceil_div(N, 16);
  static const int cand[] = {8, 7, 6, 5, 4, 3, 2, 1};
  int best = 1;
  float best_cost = 1e30f;
  static const int nt_max = getenv("DL3P_GEMM_NT_MAX") ? atoi(getenv("DL3P_GEMM_NT_MAX")) : 8;
  static const int quant = getenv("DL3P_GEMM_QUANT") ? atoi(getenv("DL3P_GEMM_QUANT")) : 1;
  // Small grids (Xception at batch 4: M = 4356, N = 728 -> 414 workgroups of 64x128 on 256 CUs): what counts is how
  // many workgroups the busiest CU has to run, times the cost of one (fixed part ~2 column blocks + nt).  112-column
  // tiles (nt = 7) exist for this regime only: 483 workgroups of 7/8 the work instead of 414.
  const long long mt64 = ceil_div(M, 64);
  const bool small = quant && mt64 * ceil_div(ntiles, 8) <= 6LL * DL3P_NUM_CUS;
  for (int c : cand) {
    if (c > nt_max || (c == 7 && !small)) continue;
    float cost;
    if (small) {
      cost = (float)ceil_div_ll(mt64 * ceil_div(ntiles, c), DL3P_NUM_CUS) * (2.f + (float)c);
    } else {
      // MFMA columns actually computed, plus the A-tile re-reads/staging that every column block repeats
      cost = (float)(ceil_div(ntiles, c) * c) * (1.f + 1.5f / (float)c);
    }
    if (cost < best_cost) { best = c; best_cost = cost; }
  }
  return best;
}

// grid: persistent workgroups over M tiles.  Small maps (M = N*33*33) give only ~137 tiles of 128 rows,
// which quantises badly over 256 CUs; 64-row tiles (MI = 1) are used whenever 128-row tiles would leave
// the chip under two rounds of work.
static void gemm_grid(int M, int N, int nt, int* gx, int* gy, int* num_m_tiles, int* mi_out, bool bn_sums = false, int force_mi = 0,
                      int force_pc = 0) {
  const int nb = ceil_div(N, 16 * nt);
  int mi = 2;
  if ((long long)ceil_div(M, 128) * nb < 4LL * DL3P_NUM_CUS) mi = 1;
  // Long GEMMs with wide column blocks (the decoder layers: 266256 rows, 256 / 304 columns): 64-row tiles leave room for
  // THREE resident workgroups per CU (137 VGPRs, 44 KB of LDS each) instead of two of 128 rows.  The SQ counters show
  // the two-workgroup version's waves parked on their barriers / load waits 21 % of the time and, sharing one matrix
  // pipe, in step with each other (profiles/r02_gemm_wave_state_counters.txt); a third workgroup fills those gaps:
  // 266256x304->256 forward 492 -> 455 us, data gradient 462 -> 428; 256->256 400 -> 376 / 348 -> 319 (same box).
  static const int long_rows = getenv("DL3P_GEMM_LONG_ROWS") ? atoi(getenv("DL3P_GEMM_LONG_ROWS")) : 60000;
  static const int long_nt = getenv("DL3P_GEMM_LONG_NT") ? atoi(getenv("DL3P_GEMM_LONG_NT")) : 2;
  // (not for the data gradient with the fused BatchNorm sums: its z-prefetch registers cap it at two workgroups per CU
  // either way -- forced under 168 VGPRs it spills 7-26 registers and is no faster -- and at 64 rows with two it is 6-15 %
  // slower: 537 -> 615 us in the step)
  if (M >= long_rows && nt >= long_nt && !bn_sums) mi = 1;
  int per_cu = nt <= 1 ? 6 : (nt == 2 ? 5 : (nt <= 4 ? 3 : 2));
  if (mi == 1 && per_cu < 3) per_cu = 3;
  static const int e_mi = getenv("DL3P_GEMM_MI") ? atoi(getenv("DL3P_GEMM_MI")) : 0;
  static const int e_pc = getenv("DL3P_GEMM_PER_CU") ? atoi(getenv("DL3P_GEMM_PER_CU")) : 0;
  if (force_mi) { mi = force_mi; if (mi == 1 && per_cu < 3) per_cu = 3; }
  if (force_pc) per_cu = force_pc;
  if (e_mi) mi = e_mi;
  if (e_pc) per_cu = e_pc;
  const int bm = 64 * mi;
  const int mt = ceil_div(M, bm);
  int gx_max = (DL3P_NUM_CUS * per_cu) / nb;
  if (gx_max < 8) gx_max = 8;
  if (gx_max > DL3P_MAX_STAT_ROWS) gx_max = DL3P_MAX_STAT_ROWS;
  int g = mt;
  if (mt > gx_max) {
    const int per = ceil_div(mt, gx_max);
    g = ceil_div(mt, per);
  }
  *gx = g; *gy = nb; *num_m_tiles = mt; *mi_out = mi;
}

// tile choice of one GEMM launch: the tuned table, a pinned option, or the heuristics
static void gemm_plan(int role, int M, int K, int N, int* nt, int* gx, int* gy, int* num_m_tiles, int* mi) {
  int force_mi = 0, force_pc = 0;
  *nt = pick_nt(N, M);
  if (role == 3 && N <= 768 && K <= 320) {
    // the data gradient with the fused BatchNorm sums: up to 64-column blocks of 64-row tiles are the widest that fit
    // three resident workgroups per CU (147 VGPRs; wider ones need 180-256) -- what the tuner picks for 9 shapes in 10
    // of this role with a short reduction (the project convs of the inverted residuals: K = 24..256), 20-34 % faster
    // than the widest-block choice on the 17424-row layers.  Long reductions (Xception: K = 728..2048 on 4356 rows) pay
    // more for re-staging A per column block than the third workgroup returns: they keep the wide blocks.
    int best = 4, waste = 1 << 30;
    for (int c = 4; c >= 2; --c) {
      const int w = ceil_div(N, 16 * c) * 16 * c - N;
      if (w < waste) { waste = w; best = c; }
    }
    if (ceil_div(N, 16) < best) best = ceil_div(N, 16);
    *nt = best;
    force_mi = 1;
  }
  if (const GemmTuned* e = gemm_tuned_lookup(role, M, K, N)) { *nt = e->nt; force_mi = e->mi; force_pc = e->pc; }
  if (g_gemm_force_nt) *nt = g_gemm_force_nt;
  if (g_gemm_force_mi) force_mi = g_gemm_force_mi;
  if (g_gemm_force_pc) force_pc = g_gemm_force_pc;
  gemm_grid(M, N, *nt, gx, gy, num_m_tiles, mi, role == 3, force_mi, force_pc);
}

template <int NT, bool B_KN, bool STATS, int MI, int BKT, bool BNB = false, bool GA = false>
static void launch_gemm_one(const GemmParams& p, dim3 grid, hipStream_t st) {
  constexpr int BM = 64 * MI, BN = 16 * NT, AP = BKT + 4;
  constexpr int BS = B_KN ? BKT * (BN + 4) : BN * AP;
  constexpr int TPP = NT < 4 ? NT : 4;
  constexpr int ES = 4 * 16 * MI * (16 * TPP + 4);
  constexpr int OPER = BM * AP + BS;
  constexpr int RED = STATS ? 2 * 4 * BN : 0;
  constexpr size_t lds = sizeof(float) * (size_t)((BKT > 32 ? (OPER > ES ? OPER : ES) : OPER + ES) + RED);
  static bool attr_set = false;
  if (!attr_set) {
    if (lds > 64 * 1024)
      (void)hipFuncSetAttribute((const void*)pw_gemm_kernel<NT, B_KN, STATS, MI, BKT, BNB, GA>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  dl3p_launch(pw_gemm_kernel<NT, B_KN, STATS, MI, BKT, BNB, GA>, grid, dim3(256), lds, st, p);
}

template <bool B_KN, bool STATS, int MI, int BKT, bool BNB = false, bool GA = false>
static void launch_gemm_mi(const GemmParams& p, int nt, dim3 grid, hipStream_t st) {
  switch (nt) {
    case 1: launch_gemm_one<1, B_KN, STATS, MI, BKT, BNB, GA>(p, grid, st); break;
    case 2: launch_gemm_one<2, B_KN, STATS, MI, BKT, BNB, GA>(p, grid, st); break;
    case 3: launch_gemm_one<3, B_KN, STATS, MI, BKT, BNB, GA>(p, grid, st); break;
    case 4: launch_gemm_one<4, B_KN, STATS, MI, BKT, BNB, GA>(p, grid, st); break;
    case 5: launch_gemm_one<5, B_KN, STATS, MI, BKT, BNB, GA>(p, grid, st); break;
    case 6: launch_gemm_one<6, B_KN, STATS, MI, BKT, BNB, GA>(p, grid, st); break;
    case 7: launch_gemm_one<7, B_KN, STATS, MI, BKT, BNB, GA>(p, grid, st); break;
    default: launch_gemm_one<8, B_KN, STATS, MI, BKT, BNB, GA>(p, grid, st); break;
  }
}

template <bool B_KN, bool STATS, bool BNB = false, bool GA = false>
static void launch_gemm(const GemmParams& p_in, int nt, int mi, dim3 grid, hipStream_t st) {
  GemmParams p = p_in;
  // (BKT = 64 -- half the barriers and staging passes per MFMA -- was measured neutral on the decoder layers and is
  // not instantiated: the loop is bound by matrix-pipe sharing between the two resident workgroups)
  if (mi == 1) launch_gemm_mi<B_KN, STATS, 1, 32, BNB, GA>(p, nt, grid, st);
  else launch_gemm_mi<B_KN, STATS, 2, 32, BNB, GA>(p, nt, grid, st);
}

static int check_mat(const char* fn, const void* ptr, int ld, int cols) {
  DL3P_CHECK_ARG(ptr != nullptr, "%s: null pointer", fn);
  DL3P_CHECK_ARG(cols > 0 && cols % 4 == 0, "%s: channel count %d must be a positive multiple of 4", fn, cols);
  DL3P_CHECK_ARG(ld % 4 == 0 && ld >= cols && aligned16(ptr), "%s: bad layout (ld=%d)", fn, ld);
  return DL3P_OK;
}

static int pwconv_fwd_impl(const char* fn, const float* x, int ldx, const float* in_scale, const float* in_shift,
                           int in_act, const float* w, bool w_kn, const float* bias, float* y, int ldy,
                           float* stat_partials, int* rows_out, int M, int K, int N, void* stream) {
  int rc = check_mat(fn, x, ldx, K);
  if (rc) return rc;
  rc = check_mat(fn, y, ldy, N);
  if (rc) return rc;
  DL3P_CHECK_ARG(w && aligned16(w) && M > 0, "%s: bad arguments", fn);
  DL3P_CHECK_ARG((unsigned long long)M * (unsigned long long)(ldx > ldy ? ldx : ldy) * 4ull < (1ull << 32),
                 "%s: operands of 4 GiB or more are not supported (M=%d)", fn, M);
  GemmParams p = {};
  p.A = x; p.lda = ldx; p.scale = in_scale; p.shift = in_shift; p.act = in_act;
  p.B = w; p.ldb = w_kn ? N : K; p.bias = bias; p.Y = y; p.ldy = ldy; p.partials = stat_partials;
  p.M = M; p.K = K; p.N = N;
  hipStream_t st = (hipStream_t)stream;
  if (!w_kn && dl3p_pw_tiny_applies(M)) {
    if (rows_out) *rows_out = 1;
    dl3p_pw_tiny_nt(x, ldx, in_scale, in_shift, in_act, w, K, bias, y, ldy, 0, stat_partials, M, K, N, st);
    DL3P_CHECK_LAUNCH(fn);
    return DL3P_OK;
  }
  SmallShape sh;
  if (M >= pw_small_min_rows() && pw_small_pick(K, N, &sh)) {
    p.b_kn = w_kn ? 1 : 0;
    const int g = pw_small_grid(M);
    if (rows_out) *rows_out = g;
    if (stat_partials) launch_pw_small_any<true>(p, sh, g, st);
    else launch_pw_small_any<false>(p, sh, g, st);
    DL3P_CHECK_LAUNCH(fn);
    return DL3P_OK;
  }
  int nt, gx, gy, mi;
  if (w_kn) {
    nt = pick_nt(N, M);
    gemm_grid(M, N, nt, &gx, &gy, &p.num_m_tiles, &mi);
  } else {
    gemm_plan(stat_partials ? 1 : 0, M, K, N, &nt, &gx, &gy, &p.num_m_tiles, &mi);
  }
  if (rows_out) *rows_out = gx;
  if (w_kn) {
    if (stat_partials) launch_gemm<true, true>(p, nt, mi, dim3(gx, gy), st);
    else launch_gemm<true, false>(p, nt, mi, dim3(gx, gy), st);
  } else {
    if (stat_partials) launch_gemm<false, true>(p, nt, mi, dim3(gx, gy), st);
    else launch_gemm<false, false>(p, nt, mi, dim3(gx, gy), st);
  }
  DL3P_CHECK_LAUNCH(fn);
  return DL3P_OK;
}

extern "C" int dl3p_pwconv_fwd(const float* x, int ldx, const float* in_scale, const float* in_shift, int in_act,
                               const float* w, const float* bias, float* y, int ldy, float* stat_partials,
                               int* rows_out, int M, int K, int N, void* stream) {
  return pwconv_fwd_impl("dl3p_pwconv_fwd", x, ldx, in_scale, in_shift, in_act, w, true, bias, y, ldy, stat_partials,
                         rows_out, M, K, N, stream);
}

// the same product with the kernel handed over transposed, wt[N][K]: the B tile then sits in LDS as [n][k] and its
// MFMA fragments are one ds_read_b128 instead of four ds_read_b32 (the layout the data-gradient GEMM gets for free)
extern "C" int dl3p_pwconv_fwd_wt(const float* x, int ldx, const float* in_scale, const float* in_shift, int in_act,
                                  const float* wt, const float* bias, float* y, int ldy, float* stat_partials,
                                  int* rows_out, int M, int K, int N, void* stream) {
  return pwconv_fwd_impl("dl3p_pwconv_fwd_wt", x, ldx, in_scale, in_shift, in_act, wt, false, bias, y, ldy,
                         stat_partials, rows_out, M, K, N, stream);
}

// ------------------------------------------------------------------------------ split-K forward
// Few rows and a long reduction (Xception's / ResNet50's ASPP at a 33 x 33 map: 4356 x 2048 -> 256) give the tiled kernel 34 row
// tiles for 256 CUs: with 32-column blocks (its best, 552 workgroups) every workgroup re-reads its 64 x 2048 slice of A for a
// quarter of the output width and the launch runs at a third of the fp32 matrix rate.  Here the REDUCTION is cut into slices of
// >= 256: 64 x 128 tiles x ~5 slices fill the chip with workgroups that each stream a 64 x 416 panel of A once; the slices leave
// slabs [S][M][N] in a workspace and splitk_finish_kernel adds them in slice order (+ bias) and takes the BatchNorm statistic rows
// from the finished output.  Deterministic; the sum is associated differently from the one-launch kernel (equal to rounding).
// (The same slices on the split-bf16 kernel were measured at 52.8-55.3 us against 59.0 here for 4356 x 2048 -> 256, 39.4-39.9 against
// 41.2 at K = 1280 -- both forms are bound by the latency of their 13 K-steps per tile, not by the matrix pipe -- and are not built in.)
__global__ __launch_bounds__(256) void splitk_finish_kernel(const float* __restrict__ slabs, int S, const float* __restrict__ bias,
                                                            float* __restrict__ y, int ldy, float* __restrict__ partials, int M,
                                                            int N, int rows_per_wg) {
  __shared__ float red[2][256 * 4];
  const int c4n = N / 4;                       // float4 columns per row; 256 % c4n == 0 (host)
  const int rpp = 256 / c4n;                   // rows per pass of the workgroup
  const int c4 = threadIdx.x % c4n, rg = threadIdx.x / c4n;
  const int r0 = blockIdx.x * rows_per_wg;
  float4 s = zero4(), ss = zero4();
  float4 b4 = zero4();
  if (bias) b4 = ld4(bias + c4 * 4);
  for (int r = r0 + rg; r < min(r0 + rows_per_wg, M); r += rpp) {
    float4 acc = ld4(slabs + ((size_t)r * N + c4 * 4));
    for (int z = 1; z < S; ++z) acc = add4(acc, ld4(slabs + (((size_t)z * M + r) * N + c4 * 4)));
    acc = add4(acc, b4);
    st4(y + (size_t)r * ldy + c4 * 4, acc);
    s = add4(s, acc);
    ss.x = fmaf(acc.x, acc.x, ss.x); ss.y = fmaf(acc.y, acc.y, ss.y); ss.z = fmaf(acc.z, acc.z, ss.z); ss.w = fmaf(acc.w, acc.w, ss.w);
  }
  if (!partials) return;
  *reinterpret_cast<float4*>(&red[0][threadIdx.x * 4]) = s;
  *reinterpret_cast<float4*>(&red[1][threadIdx.x * 4]) = ss;
  __syncthreads();
  if (rg == 0) {
    float4 a = zero4(), b = zero4();
    for (int g = 0; g < rpp; ++g) {
      a = add4(a, *reinterpret_cast<const float4*>(&red[0][(g * c4n + c4) * 4]));
      b = add4(b, *reinterpret_cast<const float4*>(&red[1][(g * c4n + c4) * 4]));
    }
    st4(partials + ((size_t)blockIdx.x * 2) * N + c4 * 4, a);
    st4(partials + ((size_t)blockIdx.x * 2 + 1) * N + c4 * 4, b);
  }
}

// -> slices (0: the one-launch kernel); kchunk_out = reduction length of a slice (a multiple of 32)
extern "C" int dl3p_pwconv_fwd_splitk_plan(int M, int K, int N) {
  static const int env = getenv("DL3P_SPLITK") ? atoi(getenv("DL3P_SPLITK")) : -1;      // (A/B switch: 0 = never)
  if (g_splitk_force == 0 || (g_splitk_force < 0 && env == 0)) return 0;
  if (!(N == 128 || N == 256 || N == 512) || K % 4 || K < 512 || M < 1024) return 0;
  if ((unsigned long long)M * (unsigned long long)K * 4ull >= (1ull << 32)) return 0;
  // measured (scripts/micro/splitk_bench.py, 64 x 128 tiles): 4356 x 2048 -> 256: one launch 93.8 us, 5 slices 59.8 (8: 64.6, 10: 62.4,
  // 4: 69.5); 4356 x 1280: 55.2 -> 41.1 (4: 46.7, 8: 46.2); 8712 x 2048: 131.5 -> 102.5;
  // 17424 x 2048 (546 tiles): 226.9 -> 182.7 (4: 183.8, 2: 196.9, 8: 197.5), x 1280: 143.9 -> 124.3-127.9.  Five slices are the best
  // or within 3 % of it at every size measured; rule: five (slices of at least 256) for up to 640 tiles of 64 x 128.
  const int tiles = ceil_div(M, 64) * ceil_div(N, 128);
  if (g_splitk_force < 0 && (tiles > 640 || K < 1024)) return 0;
  int S = g_splitk_force > 0 ? g_splitk_force : 5;
  S = std::min(S, std::min(16, K / 256));
  while (S > 1 && ceil_div(ceil_div(K, S), 32) * 32 * (S - 1) >= K) --S;   // every slice owns at least one column
  return S > 1 ? S : 0;
}
extern "C" size_t dl3p_pwconv_fwd_splitk_workspace(int M, int K, int N) {
  const int S = dl3p_pwconv_fwd_splitk_plan(M, K, N);
  return S ? sizeof(float) * (size_t)S * M * N : 0;
}

extern "C" int dl3p_pwconv_fwd_wt_splitk(const float* x, int ldx, const float* in_scale, const float* in_shift, int in_act,
                                         const float* wt, const float* bias, float* y, int ldy, float* stat_partials,
                                         int* rows_out, void* workspace, size_t workspace_bytes, int M, int K, int N,
                                         void* stream) {
  const char* fn = "dl3p_pwconv_fwd_wt_splitk";
  int rc = check_mat(fn, x, ldx, K);
  if (rc) return rc;
  rc = check_mat(fn, y, ldy, N);
  if (rc) return rc;
  DL3P_CHECK_ARG(wt && aligned16(wt) && M > 0 && workspace && aligned16(workspace), "%s: bad arguments", fn);
  const int S = dl3p_pwconv_fwd_splitk_plan(M, K, N);
  DL3P_CHECK_ARG(S > 1, "%s: M=%d K=%d N=%d is not served (dl3p_pwconv_fwd_splitk_plan; use dl3p_pwconv_fwd_wt)", fn, M, K, N);
  DL3P_CHECK_ARG(workspace_bytes >= sizeof(float) * (size_t)S * M * N, "%s: workspace of %zu bytes, %zu needed", fn, workspace_bytes,
                 sizeof(float) * (size_t)S * M * N);
  DL3P_CHECK_ARG((unsigned long long)M * (unsigned long long)std::max(ldx, ldy) * 4ull < (1ull << 32) &&
                 (unsigned long long)S * M * N * 4ull < (1ull << 32), "%s: operands of 4 GiB or more are not supported", fn);
  GemmParams p = {};
  p.A = x; p.lda = ldx; p.scale = in_scale; p.shift = in_shift; p.act = in_act;
  p.B = wt; p.ldb = K; p.bias = nullptr; p.Y = (float*)workspace; p.ldy = N; p.partials = nullptr;
  p.M = M; p.K = K; p.N = N;
  p.ksplit = S; p.kchunk = ceil_div(ceil_div(K, S), 32) * 32;
  static const int nt_env = getenv("DL3P_SPLITK_NT") ? atoi(getenv("DL3P_SPLITK_NT")) : 8;
  static const int mi_env = getenv("DL3P_SPLITK_MI") ? atoi(getenv("DL3P_SPLITK_MI")) : 1;
  const int nt = std::min(nt_env, ceil_div(N, 16)), mi = mi_env;
  p.num_m_tiles = ceil_div(M, 64 * mi);
  const int nb = ceil_div(N, 16 * nt);
  hipStream_t st = (hipStream_t)stream;
  launch_gemm<false, false>(p, nt, mi, dim3(p.num_m_tiles, nb * S), st);
  DL3P_CHECK_LAUNCH(fn);
  const int rpp = 256 / (N / 4);
  int rows_per_wg = rpp * 4;
  while (ceil_div(M, rows_per_wg) > DL3P_MAX_STAT_ROWS) rows_per_wg += rpp;
  const int wgs = ceil_div(M, rows_per_wg);
  if (rows_out) *rows_out = wgs;
  hipLaunchKernelGGL(splitk_finish_kernel, dim3(wgs), dim3(256), 0, st, (const float*)workspace, S, bias, y, ldy, stat_partials, M, N,
                     rows_per_wg);
  DL3P_CHECK_LAUNCH(fn);
  return DL3P_OK;
}

// dst[off + n*K + k] = src[off + k*N + n] for every (off, K, N) row of `table` (device, int[n][4]): the transposed
// copies of all pointwise kernels in the flat parameter buffer, refreshed once per optimiser step
// 64 x 64 tiles (round 5; 32 x 32 before: 128-byte row segments each way, 2.3 TB/s on Xception's 164 MB of kernels): a wave reads and
// writes 256 contiguous bytes per instruction, all sixteen loads of a thread before the first LDS store; clamped addresses, no load in a
// branch
__global__ __launch_bounds__(256) void transpose_batch_kernel(const float* src, float* dst, const int* table) {
  __shared__ float tile[64][65];
  const int off = table[blockIdx.x * 4], K = table[blockIdx.x * 4 + 1], N = table[blockIdx.x * 4 + 2];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int tk = (K + 63) / 64, tn = (N + 63) / 64;
  for (int tl = blockIdx.y; tl < tk * tn; tl += gridDim.y) {
    const int k0 = (tl / tn) * 64, n0 = (tl % tn) * 64;
    float v[16];
    const int n = n0 + tx, nc = n < N ? n : N - 1;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int k = k0 + ty + 4 * i;
      v[i] = src[off + (size_t)(k < K ? k : K - 1) * N + nc];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) tile[ty + 4 * i][tx] = v[i];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int nn = n0 + ty + 4 * i, k = k0 + tx;
      if (k < K && nn < N) dst[off + (size_t)nn * K + k] = tile[tx][ty + 4 * i];
    }
  }
}

extern "C" int dl3p_transpose_batch(const float* src, float* dst, const int* table, int n_matrices, void* stream) {
  DL3P_CHECK_ARG(src && dst && table && n_matrices > 0, "dl3p_transpose_batch: bad arguments");
  hipLaunchKernelGGL(transpose_batch_kernel, dim3(n_matrices, 96), dim3(256), 0, (hipStream_t)stream, src, dst, table);
  DL3P_CHECK_LAUNCH("dl3p_transpose_batch");
  return DL3P_OK;
}

extern "C" int dl3p_pwconv_bwd_data(const float* dy, int lddy, const float* w, float* gx, int ldgx, int accumulate,
                                    int M, int K, int N, void* stream) {
  int rc = check_mat("dl3p_pwconv_bwd_data", dy, lddy, N);
  if (rc) return rc;
  rc = check_mat("dl3p_pwconv_bwd_data", gx, ldgx, K);
  if (rc) return rc;
  DL3P_CHECK_ARG(w && aligned16(w) && M > 0, "dl3p_pwconv_bwd_data: bad arguments");
  DL3P_CHECK_ARG((unsigned long long)M * (unsigned long long)(lddy > ldgx ? lddy : ldgx) * 4ull < (1ull << 32),
                 "dl3p_pwconv_bwd_data: operands of 4 GiB or more are not supported (M=%d)", M);
  GemmParams p = {};
  p.A = dy; p.lda = lddy; p.act = DL3P_ACT_NONE;
  p.B = w; p.ldb = N;           // W[K][N]: output column k, reduction n contiguous
  p.Y = gx; p.ldy = ldgx; p.accumulate = accumulate;
  p.M = M; p.K = N; p.N = K;    // reduce over N, produce K columns
  if (dl3p_pw_tiny_applies(M)) {
    dl3p_pw_tiny_nt(dy, lddy, nullptr, nullptr, DL3P_ACT_NONE, w, N, nullptr, gx, ldgx, accumulate, nullptr, M, N, K,
                    (hipStream_t)stream);
    DL3P_CHECK_LAUNCH("dl3p_pwconv_bwd_data");
    return DL3P_OK;
  }
  SmallShape sh;
  if (M >= pw_small_min_rows() && pw_small_pick(N, K, &sh)) {
    p.b_kn = 0;
    launch_pw_small_any<false>(p, sh, pw_small_grid(M), (hipStream_t)stream);
    DL3P_CHECK_LAUNCH("dl3p_pwconv_bwd_data");
    return DL3P_OK;
  }
  int nt, gxn, gy, mi;
  gemm_plan(2, M, N, K, &nt, &gxn, &gy, &p.num_m_tiles, &mi);
  launch_gemm<false, false>(p, nt, mi, dim3(gxn, gy), (hipStream_t)stream);
  DL3P_CHECK_LAUNCH("dl3p_pwconv_bwd_data");
  return DL3P_OK;
}

// data gradient of a layer whose input is act(BN(z)): writes gx and, from the finished gradient in the epilogue, the
// BatchNorm-backward partial sums of that BN (the separate dl3p_bn_bwd_reduce pass over gx and z disappears)
extern "C" int dl3p_pwconv_bwd_data_bn(const float* dy, int lddy, const float* w, float* gx, int ldgx, int accumulate,
                                       int M, int K, int N, const float* z, int ldz, const float* scale,
                                       const float* shift, int act, const float* save_mean, const float* save_invstd,
                                       float* partials, int* rows_out, void* stream) {
  int rc = check_mat("dl3p_pwconv_bwd_data_bn", dy, lddy, N);
  if (rc) return rc;
  rc = check_mat("dl3p_pwconv_bwd_data_bn", gx, ldgx, K);
  if (rc) return rc;
  rc = check_mat("dl3p_pwconv_bwd_data_bn", z, ldz, K);
  if (rc) return rc;
  DL3P_CHECK_ARG(w && aligned16(w) && M > 0 && scale && shift && save_mean && save_invstd && partials && rows_out,
                 "dl3p_pwconv_bwd_data_bn: bad arguments");
  DL3P_CHECK_ARG((unsigned long long)M * (unsigned long long)(lddy > ldgx ? (lddy > ldz ? lddy : ldz) : (ldgx > ldz ? ldgx : ldz)) * 4ull < (1ull << 32),
                 "dl3p_pwconv_bwd_data_bn: operands of 4 GiB or more are not supported (M=%d)", M);
  GemmParams p = {};
  p.A = dy; p.lda = lddy; p.act = DL3P_ACT_NONE;
  p.B = w; p.ldb = N;
  p.Y = gx; p.ldy = ldgx; p.accumulate = accumulate;
  p.M = M; p.K = N; p.N = K;    // reduce over N, produce K columns
  p.partials = partials;
  p.bb_z = z; p.bb_ldz = ldz; p.bb_scale = scale; p.bb_shift = shift; p.bb_mean = save_mean; p.bb_invstd = save_invstd;
  p.bb_act = act;
  SmallShape sh;
  static const int small_bnb = getenv("DL3P_PW_SMALL_BNB") ? atoi(getenv("DL3P_PW_SMALL_BNB")) : 1;
  if (small_bnb && M >= pw_small_min_rows() && pw_small_pick(N, K, &sh)) {
    // few channels, many rows: the streaming kernel (the whole kernel matrix in LDS, no workgroup barrier in the row loop)
    p.b_kn = 0;
    const int g = pw_small_grid(M);
    *rows_out = g;
    launch_pw_small_any<true, true>(p, sh, g, (hipStream_t)stream);
    DL3P_CHECK_LAUNCH("dl3p_pwconv_bwd_data_bn");
    return DL3P_OK;
  }
  int nt, gxn, gy, mi;
  gemm_plan(3, M, N, K, &nt, &gxn, &gy, &p.num_m_tiles, &mi);
  *rows_out = gxn;
  launch_gemm<false, true, true>(p, nt, mi, dim3(gxn, gy), (hipStream_t)stream);
  DL3P_CHECK_LAUNCH("dl3p_pwconv_bwd_data_bn");
  return DL3P_OK;
}

// ------------------------------------------------------------------------------ split-bf16 twins (pw_split.hip)
// The same three products on the bf16 matrix pipe with fp32-accurate results: the conv kernel arrives pre-split into three bf16
// planes [3][rows][pitch] (dl3p_split_bf16x3_batch; rows = the GEMM's OUTPUT columns, reduction index contiguous, pitch a
// multiple of 32 with zero padding), the activations are split while their tile is staged.  Shapes the tiled kernel does not serve
// (few rows, or few-channel layers on the streaming kernels) must go through the fp32 entry points: *_sb_supported says which.
void dl3p_launch_gemm_sb(const GemmParams& p, bool stats, bool bnb, bool ga, int nt, int mi, int wm, dim3 grid, hipStream_t st);
int dl3p_wgrad_sb_plan(int M, int K, int N, int max_slabs, int tile, int per_cu, int* kf, int* nw, int* ktiles, int* ntiles, int* mrows);
void dl3p_launch_wgrad_sb(const float* x, int ldx, const float* scale, const float* shift, int act, const float* dy, int lddy,
                          float* slabs, int M, int K, int N, int kf, int nw, int ktiles, int ntiles, int mrows, int splits, hipStream_t st);
static bool split_wgrad_on() {      // follows the switch of the split forward / data-gradient GEMMs unless set itself
  if (g_split_wgrad < 0)
    g_split_wgrad = getenv("DL3P_SPLIT_WGRAD") ? atoi(getenv("DL3P_SPLIT_WGRAD")) : (getenv("DL3P_SPLIT_GEMM") ? atoi(getenv("DL3P_SPLIT_GEMM")) : 1);
  return g_split_wgrad > 0;
}
extern "C" int dl3p_pwconv_sb_pays(int role, int M, int K, int N);
// does this weight gradient run on the split-bf16 kernel?  -> slabs (0: no) + its plan.  The measured verdict / tile of this exact
// launch where there is one (csrc/sb_tuned.h: g_sb_pays role 4, g_sb_tuned role 9 {tile, workgroups per CU}), else the rule
static int wgrad_sb_route(int M, int K, int N, size_t max_slabs, int* kf, int* nw, int* kt, int* nt, int* mrows) {
  if (!split_wgrad_on()) return 0;
  int tile = g_sbw_force_tile, per_cu = g_sbw_force_pc;
  if (tile < 0 && per_cu <= 0) {
    const int pays = dl3p_pwconv_sb_pays(4, M, K, N);
    if (pays == 0 || (pays < 0 && (K < 128 || N < 128 || M < 16384))) return 0;
    if (const GemmTuned* e = gemm_tuned_lookup(9, M, K, N)) { tile = e->nt; per_cu = e->mi; }
  }
  if (max_slabs > (size_t)DL3P_MAX_STAT_ROWS) max_slabs = DL3P_MAX_STAT_ROWS;
  return dl3p_wgrad_sb_plan(M, K, N, (int)max_slabs, tile, per_cu, kf, nw, kt, nt, mrows);
}
void dl3p_launch_wgrad_sb_gx(const float* x, int ldx, const float* scale, const float* shift, int act, const float* dy, int lddy,
                             float* slabs, int M, int K, int N, const int* geo, int kf, int nw, int ktiles, int ntiles, int mrows, int splits,
                             hipStream_t st);
bool dl3p_sb_wide_config(int nt, int mi, int wm);
bool dl3p_sb_rs_supported(int role, int M, int K, int N);
int dl3p_sb_rs_grid(int M);
bool dl3p_launch_gemm_sbr(const GemmParams& p, int mode, int grid, hipStream_t st);
// the row-stationary form (pw_split_rs.hip; gemm_plan_sb reports it as wm = 3): pinned by dl3p_set_option("sb_rs", 1), otherwise by
// the measured table ({2, 1, 103} rows of csrc/sb_tuned.h) or the rule below
static bool sb_rs_route(int role, int M, int K, int N) {
  if (!dl3p_sb_rs_supported(role, M, K, N)) return false;
  if (g_sb_pipe < 0) g_sb_pipe = getenv("DL3P_SB_PIPE") ? atoi(getenv("DL3P_SB_PIPE")) : 0;      // (before its first use: ADVICE r04)
  if (g_sb_force_wm != 0 || g_sb_pipe > 0 || g_gemm_force_nt || g_gemm_force_mi) return false;     // another form is pinned
  if (g_sb_rs < 0) {
    static const int env = getenv("DL3P_SB_RS") ? atoi(getenv("DL3P_SB_RS")) : -1;
    if (env >= 0) return env != 0;
    // the measured table (csrc/sb_tuned.h, {2, 1, 103} rows) decides where it knows the launch; elsewhere the rule measured on the
    // decoder shapes (scripts/micro/sb_rs.py, profiles/r04_split_gemm_row_stationary.txt): long data gradients with a reduction of
    // 225-320 -- with the fused BatchNorm-backward sums 362 against 494-512 us on 266256 x 256 -> 304, 288 against 340 onto 256
    // columns, 175 against 216 at 131072 rows; plain 258 against 345, 213 against 234, 120 against 146.  Forwards tie (247 against
    // 255 at K = 256) or lose (K = 304: 372 against 349), as does everything under ~10^5 rows (one workgroup per CU and 64-row half
    // tiles: tile quantisation)
    static const int fwd = getenv("DL3P_SB_RS_FWD") ? atoi(getenv("DL3P_SB_RS_FWD")) : 0;      // (A/B switch: forwards too)
    if (fwd && role <= 1 && M >= 131072 && K > 224) return true;
    if (const GemmTuned* e = gemm_tuned_lookup(role + 5, M, K, N)) return e->pc == 103;
    return role >= 2 && M >= 131072 && K > 224;
  }
  return g_sb_rs == 1;
}
void dl3p_launch_gemm_sbp(const GemmParams& p, bool stats, bool bnb, int nt, int mi, dim3 grid, hipStream_t st);
// the pinned-schedule form (pw_split3.hip; gemm_plan_sb reports it as wm = 4): forwards onto 256 columns from 65536 rows up whose
// reduction is 8, 10, 12 ... K-steps long.  dl3p_set_option("sb3", 1) takes it wherever it is supported, 0 never, -1 (default) by
// the rule -- unless the measured table knows the launch
bool dl3p_sb3_supported(int role, int M, int K, int N, int pitch, int act, bool has_scale, bool accumulate, bool bias);
int dl3p_sb3_grid(int M);
bool dl3p_launch_gemm_sb3(GemmParams p, bool stats, int grid, hipStream_t st);
bool dl3p_launch_gemm_sb3d(GemmParams p, bool bnb, int grid, hipStream_t st);
static thread_local int t_sb3_veto = 0;        // set while a launch whose prologue / epilogue the form does not serve is planned again
static bool sb3_route(int role, int M, int K, int N) {
  if (g_sb3 < 0) { static const int env = getenv("DL3P_SB3") ? atoi(getenv("DL3P_SB3")) : -1; if (env >= 0) g_sb3 = env; }
  const int pitch = (K + 31) / 32 * 32;
  if (t_sb3_veto || g_sb3 == 0 || !dl3p_sb3_supported(role, M, K, N, pitch, DL3P_ACT_NONE, true, false, false)) return false;
  if (g_sb3 == 1) return true;
  if (g_sb_pipe < 0) g_sb_pipe = getenv("DL3P_SB_PIPE") ? atoi(getenv("DL3P_SB_PIPE")) : 0;
  if (g_sb_force_wm != 0 || g_sb_pipe > 0 || g_gemm_force_nt || g_gemm_force_mi || g_sb_rs == 1) return false;       // another form is pinned
  if (gemm_tuned_lookup(role + 5, M, K, N)) return false;
  return M >= 65536;
}


// tile choice of a split-bf16 launch.  Wide tiles (one workgroup per CU: 128 or 256 rows x up to 256 columns, the A tile split
// once for all of N) where there are enough row tiles to go round; otherwise the 2-workgroups-per-CU tiles of the fp32 kernel.
static void gemm_plan_sb(int role, int M, int K, int N, int* nt, int* gx, int* gy, int* num_m_tiles, int* mi, int* wm) {
  int force_mi = 0, force_pc = 0;
  *nt = pick_nt(N, M);
  *wm = 1;
  if (sb3_route(role, M, K, N)) {
    *wm = 4; *nt = 16; *mi = 2; *gx = dl3p_sb3_grid(M); *gy = 1; *num_m_tiles = ceil_div(M, 128);
    return;
  }
  if (sb_rs_route(role, M, K, N)) {
    *wm = 3; *nt = 2; *mi = 1; *gx = dl3p_sb_rs_grid(M); *gy = 1; *num_m_tiles = ceil_div(M, 64);
    return;
  }
  if (g_sb_pipe < 0) g_sb_pipe = getenv("DL3P_SB_PIPE") ? atoi(getenv("DL3P_SB_PIPE")) : 0;     // measured slower than the symmetric form: opt-in
  if (g_sb_pipe && g_sb_force_wm <= 0) {
    // producer / consumer form: one 512-thread workgroup per CU, 128 (or 64) rows x up to 128 columns; *wm = 0 marks it
    if (*nt > 8) *nt = 8;
    if (g_gemm_force_nt) *nt = g_gemm_force_nt > 8 ? 8 : g_gemm_force_nt;
    *mi = g_gemm_force_mi ? g_gemm_force_mi : (M >= 4096 ? 2 : 1);
    const int nb = ceil_div(N, 16 * *nt), mt = ceil_div(M, 64 * *mi);
    int gxm = DL3P_NUM_CUS / nb;
    if (gxm < 1) gxm = 1;
    if (gxm > DL3P_MAX_STAT_ROWS) gxm = DL3P_MAX_STAT_ROWS;
    int g = mt;
    if (mt > gxm) g = ceil_div(mt, ceil_div(mt, gxm));
    *gx = g; *gy = nb; *num_m_tiles = mt; *wm = 0;
    return;
  }
  // measured (scripts/micro/sb_gemm.py, profiles/r03_split_gemm.txt): 128-row tiles with the widest column block win on every
  // shape with a few thousand rows or more (the fp32 kernel's 64-row / three-workgroup choice for long GEMMs loses here: two A
  // register sets); with the fused BatchNorm sums 128 x 64, the widest that does not spill.  The one-workgroup-per-CU wide tiles
  // (sb_wm) pay on long forwards onto 256-column layers only (below); elsewhere they tie or lose and stay opt-in.
  if (M >= 4096) force_mi = 2;
  if (role == 3 && N > 64) { *nt = 4; force_mi = 2; }
  int wide_nt = 0, wide_mi = 2, wide_wm = 1;
  bool measured = false;
  bool wide_bnb = false;
  if (const GemmTuned* e = gemm_tuned_lookup(role + 5, M, K, N)) {      // roles 5..8 (csrc/sb_tuned.h); pc > 100: wide family, wm = pc - 100
    measured = true;
    if (e->pc > 100) { wide_nt = e->nt; wide_mi = e->mi; wide_wm = e->pc - 100; wide_bnb = true; }
    else { wide_nt = 0; *nt = e->nt; force_mi = e->mi; force_pc = e->pc; }
  }
  // long forwards onto 256-column layers: 128 rows x 256 columns, 512 threads, one workgroup per CU (the A tile is split once for
  // all of N): 306 against 335 us on 266256 x 304 -> 256, 255 against 278 on K = 256, 98 against 107 on 74498 rows -- since the
  // operand requests stopped being drained at every stage (pw_split.hip, the note in step()); before that the wide tiles tied
  if (!measured && role <= 1 && N % 256 == 0 && M >= 65536) { wide_nt = 16; wide_mi = 1; wide_wm = 2; }
  // the long decoder data gradients with the fused BatchNorm sums: 256 rows x 128 columns, 512 threads (334 against 368 us on
  // 266256 x 256 -> 256, 469 against 495 onto 304 columns; the 256-column tiles lose here -- the z tile of the sums comes on top)
  if (!measured && role == 3 && N >= 256 && M >= 131072) { wide_nt = 8; wide_mi = 2; wide_wm = 2; wide_bnb = true; }
  if (g_sb_force_wm > 0) { wide_nt = g_sb_force_nt ? g_sb_force_nt : 16; wide_wm = g_sb_force_wm; wide_mi = g_gemm_force_mi ? g_gemm_force_mi : 2; }
  if (g_sb_force_wm < 0) wide_nt = 0;
  if (wide_nt && dl3p_sb_wide_config(wide_nt, wide_mi, wide_wm) && (role != 3 || wide_bnb || g_sb_force_wm > 0)) {     // (role 3 takes the wide family only where measured -- above -- or pinned)
    *nt = wide_nt; *mi = wide_mi; *wm = wide_wm;
    const int bm = 64 * wide_mi * wide_wm, nb = ceil_div(N, 16 * wide_nt);
    const int mt = ceil_div(M, bm);
    int gxm = DL3P_NUM_CUS / nb;
    if (gxm < 1) gxm = 1;
    if (gxm > DL3P_MAX_STAT_ROWS) gxm = DL3P_MAX_STAT_ROWS;
    int g = mt;
    if (mt > gxm) g = ceil_div(mt, ceil_div(mt, gxm));
    *gx = g; *gy = nb; *num_m_tiles = mt;
    return;
  }
  if (g_gemm_force_nt) *nt = g_gemm_force_nt;
  if (g_gemm_force_mi) force_mi = g_gemm_force_mi;
  if (g_gemm_force_pc) force_pc = g_gemm_force_pc;
  if (*nt > 8) *nt = 8;
  gemm_grid(M, N, *nt, gx, gy, num_m_tiles, mi, role == 3, force_mi, force_pc);
  // 128-row tiles with the fused BatchNorm sums spill from 80 columns up (two A register sets + the z prefetch)
  if (role == 3 && *mi == 2 && *nt > 4) gemm_grid(M, N, *nt, gx, gy, num_m_tiles, mi, true, 1, force_pc);
}

extern "C" int dl3p_pwconv_sb_pays(int role, int M, int K, int N) {
  // measured verdict for this exact launch (csrc/sb_tuned.h): 1 the split kernel is faster than the fp32-input MFMA kernel,
  // 0 it is not, -1 never measured (or the tables are switched off) -- the caller's threshold rule decides
  if (g_gemm_use_table < 0) g_gemm_use_table = getenv("DL3P_GEMM_TUNED") ? atoi(getenv("DL3P_GEMM_TUNED")) : 1;
  if (!g_gemm_use_table) return -1;
  for (size_t i = 0; i < sizeof(g_sb_pays) / sizeof(g_sb_pays[0]); ++i) {
    const SbPays& e = g_sb_pays[i];
    if (e.role == role && e.M == M && e.K == K && e.N == N) return e.pays;
  }
  return -1;
}

extern "C" int dl3p_pwconv_sb_supported(int role, int M, int K, int N) {
  // role 0 / 1 forward, 2 / 3 data gradient: (M, K, N) as launched (K = reduction length)
  SmallShape sh;
  if (M <= 0 || K < 4 || N < 4 || K % 4 || N % 4 || dl3p_pw_tiny_applies(M)) return 0;
  if (M >= pw_small_min_rows() && pw_small_pick(K, N, &sh)) return 0;
  return 1;
}

static int check_sb(const char* fn, const void* wsp, int pitch, int K) {
  DL3P_CHECK_ARG(wsp && aligned16(wsp) && pitch % 32 == 0 && pitch >= K, "%s: the split kernel must be [3][rows][pitch], pitch a multiple of 32 >= %d (got %d)", fn, K, pitch);
  return DL3P_OK;
}

extern "C" int dl3p_pwconv_fwd_sb(const float* x, int ldx, const float* in_scale, const float* in_shift, int in_act,
                                  const void* wsp, int pitch, const float* bias, float* y, int ldy, float* stat_partials,
                                  int* rows_out, int M, int K, int N, void* stream) {
  const char* fn = "dl3p_pwconv_fwd_sb";
  int rc = check_mat(fn, x, ldx, K);
  if (rc) return rc;
  rc = check_mat(fn, y, ldy, N);
  if (rc) return rc;
  rc = check_sb(fn, wsp, pitch, K);
  if (rc) return rc;
  DL3P_CHECK_ARG(M > 0 && dl3p_pwconv_sb_supported(stat_partials ? 1 : 0, M, K, N), "%s: shape M=%d K=%d N=%d is not served by the tiled kernel", fn, M, K, N);
  DL3P_CHECK_ARG((unsigned long long)M * (unsigned long long)(ldx > ldy ? ldx : ldy) * 4ull < (1ull << 32),
                 "%s: operands of 4 GiB or more are not supported (M=%d)", fn, M);
  GemmParams p = {};
  p.A = x; p.lda = ldx; p.scale = in_scale; p.shift = in_shift; p.act = in_act;
  p.Bsp = (const unsigned short*)wsp; p.bsp_pitch = pitch; p.bsp_plane = (long long)N * pitch;
  p.bias = bias; p.Y = y; p.ldy = ldy; p.partials = stat_partials;
  p.M = M; p.K = K; p.N = N;
  int nt, gx, gy, mi, wm;
  gemm_plan_sb(stat_partials ? 1 : 0, M, K, N, &nt, &gx, &gy, &p.num_m_tiles, &mi, &wm);
  if (wm == 4 && !dl3p_sb3_supported(stat_partials ? 1 : 0, M, K, N, pitch, in_act, in_scale != nullptr, false, bias != nullptr)) {
    static const bool dbg = getenv("DL3P_SB3_DEBUG") != nullptr;
    if (dbg) fprintf(stderr, "dl3p_pwconv_fwd_sb: pinned form vetoed (M=%d K=%d N=%d pitch=%d act=%d scale=%d bias=%d stats=%d)\n", M, K, N, pitch, in_act, in_scale != nullptr, bias != nullptr, stat_partials != nullptr);
    t_sb3_veto = 1;          // (an activation / bias / pitch the pinned form does not serve: the tiled kernels take the launch)
    gemm_plan_sb(stat_partials ? 1 : 0, M, K, N, &nt, &gx, &gy, &p.num_m_tiles, &mi, &wm);
    t_sb3_veto = 0;
  }
#ifdef DL3P_SB_ABLATE
  { const char* e = getenv("DL3P_SB_ABLATE"); p.stagger = e ? atoi(e) : 0; if (p.stagger == 100 && stat_partials) p.B = stat_partials + (size_t)DL3P_MAX_STAT_ROWS * 2 * N; }      // (ablation build: stamps behind the partial rows)
#endif
  if (rows_out) *rows_out = gx;
  if (wm == 4 && getenv("DL3P_SB3_DEBUG")) fprintf(stderr, "dl3p_pwconv_fwd_sb: pinned form M=%d K=%d N=%d grid %d\n", M, K, N, gx);
  if (wm == 4) DL3P_CHECK_ARG(dl3p_launch_gemm_sb3(p, stat_partials != nullptr, gx, (hipStream_t)stream), "%s: no pinned-schedule instantiation for activation %d", fn, in_act);
  else if (wm == 3) DL3P_CHECK_ARG(dl3p_launch_gemm_sbr(p, stat_partials ? 1 : 0, gx, (hipStream_t)stream), "%s: no row-stationary instantiation for K=%d", fn, K);
  else if (wm == 0) dl3p_launch_gemm_sbp(p, stat_partials != nullptr, false, nt, mi, dim3(gx, gy), (hipStream_t)stream);
  else dl3p_launch_gemm_sb(p, stat_partials != nullptr, false, false, nt, mi, wm, dim3(gx, gy), (hipStream_t)stream);
  DL3P_CHECK_LAUNCH(fn);
  return DL3P_OK;
}

// gx[M][K] (+)= dy[M][N] . W[K][N]^T with W pre-split as [3][K][pitch >= N]; z != NULL: also the BatchNorm-backward partial sums
// of dl3p_pwconv_bwd_data_bn
extern "C" int dl3p_pwconv_bwd_data_sb(const float* dy, int lddy, const void* wsp, int pitch, float* gx, int ldgx, int accumulate,
                                       int M, int K, int N, const float* z, int ldz, const float* scale, const float* shift,
                                       int act, const float* save_mean, const float* save_invstd, float* partials,
                                       int* rows_out, void* stream) {
  const char* fn = "dl3p_pwconv_bwd_data_sb";
  int rc = check_mat(fn, dy, lddy, N);
  if (rc) return rc;
  rc = check_mat(fn, gx, ldgx, K);
  if (rc) return rc;
  rc = check_sb(fn, wsp, pitch, N);
  if (rc) return rc;
  const bool bnb = z != nullptr;
  if (bnb) {
    rc = check_mat(fn, z, ldz, K);
    if (rc) return rc;
    DL3P_CHECK_ARG(scale && shift && save_mean && save_invstd && partials && rows_out, "%s: bad BatchNorm arguments", fn);
  }
  DL3P_CHECK_ARG(M > 0 && dl3p_pwconv_sb_supported(bnb ? 3 : 2, M, N, K), "%s: shape M=%d K=%d N=%d is not served by the tiled kernel", fn, M, K, N);
  const int ldm = lddy > ldgx ? (lddy > ldz ? lddy : ldz) : (ldgx > ldz ? ldgx : ldz);
  DL3P_CHECK_ARG((unsigned long long)M * (unsigned long long)ldm * 4ull < (1ull << 32), "%s: operands of 4 GiB or more are not supported (M=%d)", fn, M);
  GemmParams p = {};
  p.A = dy; p.lda = lddy; p.act = DL3P_ACT_NONE;
  p.Bsp = (const unsigned short*)wsp; p.bsp_pitch = pitch; p.bsp_plane = (long long)K * pitch;
  p.Y = gx; p.ldy = ldgx; p.accumulate = accumulate;
  p.M = M; p.K = N; p.N = K;    // reduce over N, produce K columns
  if (bnb) {
    p.partials = partials;
    p.bb_z = z; p.bb_ldz = ldz; p.bb_scale = scale; p.bb_shift = shift; p.bb_mean = save_mean; p.bb_invstd = save_invstd; p.bb_act = act;
  }
  int nt, gxn, gy, mi, wm;
  gemm_plan_sb(bnb ? 3 : 2, M, N, K, &nt, &gxn, &gy, &p.num_m_tiles, &mi, &wm);
  if (rows_out) *rows_out = gxn;
  if (wm == 3) DL3P_CHECK_ARG(dl3p_launch_gemm_sbr(p, bnb ? 2 : 0, gxn, (hipStream_t)stream), "%s: no row-stationary instantiation for K=%d", fn, N);
  else if (wm == 0) dl3p_launch_gemm_sbp(p, bnb, bnb, nt, mi, dim3(gxn, gy), (hipStream_t)stream);
  else dl3p_launch_gemm_sb(p, bnb, bnb, false, nt, mi, wm, dim3(gxn, gy), (hipStream_t)stream);
  DL3P_CHECK_LAUNCH(fn);
  return DL3P_OK;
}

bool dl3p_sb_rs_fold_supported(int M, int K, int N, int act);
bool dl3p_sb3d_supported(int M, int kout, int nred, int pitch, int f_act, int bb_act, bool bnb, bool accumulate);
static bool sb3d_takes(int M, int K, int N, int pitch, int bn_act, int front_act, bool bnb, bool accumulate) {
  if (g_sb3 < 0) { static const int env = getenv("DL3P_SB3") ? atoi(getenv("DL3P_SB3")) : -1; if (env >= 0) g_sb3 = env; }
  // OPT-IN by rule (DL3P_SB3_DGRAD=1; dl3p_set_option("sb3", 1) takes it wherever it is supported): measured on MI355X the pinned form
  // is 7-12 % faster than the row-stationary kernel without the fused sums (263-300 against 281-340 us on 262144-266256 rows) and
  // level with it with them (346-396 against 343-426), and the headline step does not move (11.87 ms either way): this launch moves
  // 1.09-1.36 GB (g, z, dz, gx and the front layer's z) -- 240-300 us at the 4.5-5 TB/s such kernels reach -- so it is bound by HBM,
  // not by the matrix pipe (scripts/micro/sb3d_bench.py)
  static const int sb3d = getenv("DL3P_SB3_DGRAD") ? atoi(getenv("DL3P_SB3_DGRAD")) : 0;
  if (g_sb3 == 0 || !dl3p_sb3d_supported(M, K, N, pitch, bn_act, front_act, bnb, accumulate)) return false;
  return g_sb3 == 1 || (sb3d && M >= 65536);
}
extern "C" int dl3p_pwconv_bwd_data_sb_apply_supported(int M, int K, int N, int bn_act, int with_sums) {
  // (M, K, N) as dl3p_pwconv_bwd_data_sb: K output columns, N the reduction = channels of the folded BatchNorm.  Two kernels serve
  // it: the pinned-schedule form (256 x 256, no accumulation: the call falls back where the caller accumulates) and the
  // row-stationary one
  if (sb3d_takes(M, K, N, (N + 31) / 32 * 32, bn_act, DL3P_ACT_RELU, with_sums != 0, false) && M >= 131072) return 1;
  if (M < 131072 || !dl3p_sb_rs_fold_supported(M, N, K, bn_act)) return 0;
  return 1;
}

extern "C" int dl3p_pwconv_bwd_data_sb_apply(const float* g, int ldg, const float* z_out, int ldz_out, const float* bn_scale,
                                             const float* bn_shift, int bn_act, const float* bn_mean, const float* bn_invstd,
                                             const float* bn_coef, float* dz, int lddz, const void* wsp, int pitch, float* gx,
                                             int ldgx, int accumulate, int M, int K, int N, const float* z, int ldz,
                                             const float* scale, const float* shift, int act, const float* save_mean,
                                             const float* save_invstd, float* partials, int* rows_out, void* stream) {
  const char* fn = "dl3p_pwconv_bwd_data_sb_apply";
  int rc = check_mat(fn, g, ldg, N);
  if (rc) return rc;
  rc = check_mat(fn, z_out, ldz_out, N);
  if (rc) return rc;
  rc = check_mat(fn, dz, lddz, N);
  if (rc) return rc;
  rc = check_mat(fn, gx, ldgx, K);
  if (rc) return rc;
  rc = check_sb(fn, wsp, pitch, N);
  if (rc) return rc;
  DL3P_CHECK_ARG(bn_scale && bn_shift && bn_mean && bn_invstd && bn_coef, "%s: bad BatchNorm-apply arguments", fn);
  const bool bnb = z != nullptr;
  if (bnb) {
    rc = check_mat(fn, z, ldz, K);
    if (rc) return rc;
    DL3P_CHECK_ARG(scale && shift && save_mean && save_invstd && partials && rows_out, "%s: bad BatchNorm arguments", fn);
  }
  const bool take3 = sb3d_takes(M, K, N, pitch, bn_act, act, bnb, accumulate != 0);
  DL3P_CHECK_ARG(take3 || dl3p_pwconv_bwd_data_sb_apply_supported(M, K, N, bn_act, bnb), "%s: shape M=%d K=%d N=%d act %d is not served", fn, M, K, N, bn_act);
  int ldm = ldg > ldgx ? ldg : ldgx;
  if (ldz > ldm) ldm = ldz;
  if (ldz_out > ldm) ldm = ldz_out;
  if (lddz > ldm) ldm = lddz;
  DL3P_CHECK_ARG((unsigned long long)M * (unsigned long long)ldm * 4ull < (1ull << 32), "%s: operands of 4 GiB or more are not supported (M=%d)", fn, M);
  GemmParams p = {};
  p.A = g; p.lda = ldg; p.act = DL3P_ACT_NONE;
  p.f_z = z_out; p.f_ldz = ldz_out; p.f_scale = bn_scale; p.f_shift = bn_shift; p.f_mean = bn_mean; p.f_invstd = bn_invstd;
  p.f_coef = bn_coef; p.f_act = bn_act; p.f_dz = dz; p.f_lddz = lddz;
  p.Bsp = (const unsigned short*)wsp; p.bsp_pitch = pitch; p.bsp_plane = (long long)K * pitch;
  p.Y = gx; p.ldy = ldgx; p.accumulate = accumulate;
  p.M = M; p.K = N; p.N = K;    // reduce over N, produce K columns
  if (bnb) {
    p.partials = partials;
    p.bb_z = z; p.bb_ldz = ldz; p.bb_scale = scale; p.bb_shift = shift; p.bb_mean = save_mean; p.bb_invstd = save_invstd; p.bb_act = act;
  }
  // the pinned-schedule form (pw_split3.hip) where it serves the launch: 256 output columns over a reduction of 256,
  // no accumulation; dl3p_set_option("sb3", 0) / DL3P_SB3_DGRAD=0 keep the row-stationary kernel
  if (take3) {
    const int g3 = dl3p_sb3_grid(M);
    if (rows_out) *rows_out = g3;
    DL3P_CHECK_ARG(dl3p_launch_gemm_sb3d(p, bnb, g3, (hipStream_t)stream), "%s: no pinned-schedule instantiation for activation %d", fn, bn_act);
    DL3P_CHECK_LAUNCH(fn);
    return DL3P_OK;
  }
  p.num_m_tiles = ceil_div(M, 64);
  const int gxn = dl3p_sb_rs_grid(M);
  if (rows_out) *rows_out = gxn;
  DL3P_CHECK_ARG(dl3p_launch_gemm_sbr(p, bnb ? 2 : 0, gxn, (hipStream_t)stream), "dl3p_pwconv_bwd_data_sb_apply: no row-stationary instantiation for K=%d", N);
  DL3P_CHECK_LAUNCH(fn);
  return DL3P_OK;
}

// ------------------------------------------------------------------------------ weight gradient
// One workgroup = one 64(k) x 64(n) tile of GW over one slice of M; slices are summed by
// dl3p_reduce_rows (fixed order -> deterministic).  Both operands are staged in their natural
// [m][channel] layout (pitch 68: rows 4 apart land 16 banks apart) and read as ds_read_b32 fragments.
#define WPITCH 68
struct WgradParams {
  const float* X; int ldx; const float* scale; const float* shift; int act;
  const float* DY; int lddy;
  float* slabs;
  int M, K, N;
  int ktiles, ntiles, mchunk;
  // implicit-GEMM gather of X (GX instantiations; see GemmParams): row m = (n, y, x) over g_RH x g_RW output pixels,
  // column k = tap * g_C + c, element = input [N][g_SH][g_SW][ldx] at (y * g_mul + g_ay + ky * g_d, ...), zero outside
  int g_RH, g_RW, g_SH, g_SW, g_C, g_kw, g_mul, g_ay, g_ax, g_d;
  float g_invRW, g_invRH;   // 1 / g_RW, 1 / g_RH for divmod_small
  // BNA instantiations: DY is the gradient g of act(BN(z)), not of z.  dz = c0 * (g * act'(z*scale+shift) - c1 - xhat * c2)
  // (what dl3p_bn_bwd_apply writes) is formed while the tile is staged and, by the workgroups of the first k tile, written
  // to DZ for the data gradient that follows: the apply pass over (g, z, dz) and its launch disappear.
  const float* Z; int ldz;
  const float* b_scale; const float* b_shift; const float* b_mean; const float* b_invstd; const float* b_coef; int b_act;
  float* DZ; int lddz;
};

// q = a / d, *r = a % d for 0 <= a < 2^24 (exact in float) and 0 < d < 2^14: one multiply by the reciprocal and one
// correction step instead of the ~30-instruction 32-bit division (the weight-gradient gather decodes every staged row)
__device__ __forceinline__ int divmod_small(int a, int d, float inv, int* r) {
  int q = (int)((float)a * inv);
  int rem = a - q * d;
  if (rem < 0) { --q; rem += d; }
  if (rem >= d) { ++q; rem -= d; }
  *r = rem;
  return q;
}

// Tile = (64 KW) x (16 NW) of GW: wave w owns k rows [16 KW w, 16 KW (w+1)) and all NW column tiles.  Larger
// tiles re-read X (N / TN times) and DY (K / TK times) less often -- at 64 x 64 the 304 x 256 decoder layer
// pulls 2.7 GB through L2 for 0.6 GB of operands.  Loads are unconditional on clamped offsets, zeroed by select.
template <int KW, int NW, bool GX = false, bool BNA = false>
__global__ __launch_bounds__(256, 2) void pw_wgrad_kernel(WgradParams p) {
  constexpr int TK = 64 * KW, TN = 16 * NW;
  constexpr int XP = TK + 4, DP = TN + 4;      // pitches: rows 4 apart land 16 banks apart
  constexpr int XQ = TK / 4, DQ = TN / 4;      // float4 per staged row
  constexpr int NX = (32 * XQ) / 256, ND = (32 * DQ + 255) / 256;
  __shared__ __attribute__((aligned(16))) float Xs[32 * XP];
  __shared__ __attribute__((aligned(16))) float Ds[32 * DP];
  const int t = threadIdx.x, l = t & 63, w = t >> 6, l15 = l & 15, q = l >> 4;
  const int tile = blockIdx.x;
  const int kt = tile / p.ntiles, nt = tile - kt * p.ntiles;
  const int k0 = kt * TK, n0 = nt * TN;
  const int m_begin = blockIdx.y * p.mchunk;
  const int m_end = min(p.M, m_begin + p.mchunk);
  // per-thread staging constants
  int xr[NX], dr[ND];
  uint32_t xo[NX], dof[ND];
  bool xok[NX], dok[ND];
  float4 xsc[NX], xsh[NX];
  int gx_dy[GX ? NX : 1], gx_dx[GX ? NX : 1];
  uint32_t gx_ok = 0;
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    const int idx = t + 256 * i;
    xr[i] = idx / XQ;
    const int c = k0 + (idx - xr[i] * XQ) * 4;
    xok[i] = c < p.K;
    xo[i] = (uint32_t)min(c, p.K - 4) * 4u;
    xsc[i] = make_float4(1.f, 1.f, 1.f, 1.f); xsh[i] = zero4();
    if (GX) {
      // this thread's k columns never change: tap offsets and channel of each, once
      const int k = min(c, p.K - 4);
      const int tap = k / p.g_C, ch = k - tap * p.g_C;
      const int ky = tap / p.g_kw, kx = tap - ky * p.g_kw;
      gx_dy[i] = p.g_ay + ky * p.g_d;
      gx_dx[i] = p.g_ax + kx * p.g_d;
      xo[i] = (uint32_t)ch * 4u;
      if (p.scale) { xsc[i] = ld4(p.scale + ch); xsh[i] = ld4(p.shift + ch); }
    } else if (p.scale) { xsc[i] = ld4(p.scale + min(c, p.K - 4)); xsh[i] = ld4(p.shift + min(c, p.K - 4)); }
  }
  // BNA: dz = bA * g * act'(z * bsc + bsh) - bC * z + bD per channel (bA = c0, bC = c0 * invstd * c2, bD = bC * mean - c0 * c1)
  float4 bA[BNA ? ND : 1], bC[BNA ? ND : 1], bD[BNA ? ND : 1], bsc[BNA ? ND : 1], bsh[BNA ? ND : 1];
#pragma unroll
  for (int i = 0; i < ND; ++i) {
    const int idx = min(t + 256 * i, 32 * DQ - 1);
    dr[i] = idx / DQ;
    const int c = n0 + (idx - dr[i] * DQ) * 4;
    dok[i] = (t + 256 * i < 32 * DQ) && c < p.N;
    dof[i] = (uint32_t)min(c, p.N - 4) * 4u;
    if (BNA) {
      const int cc = min(c, p.N - 4);
      const float4 one = make_float4(1.f, 1.f, 1.f, 1.f);
      bsc[i] = p.b_scale ? ld4(p.b_scale + cc) : one;
      bsh[i] = p.b_shift ? ld4(p.b_shift + cc) : zero4();
      const float4 mu = ld4(p.b_mean + cc), is = ld4(p.b_invstd + cc);
      const float4 c0 = ld4(p.b_coef + cc), c1 = ld4(p.b_coef + p.N + cc), c2 = ld4(p.b_coef + 2 * p.N + cc);
      bA[i] = c0;
      bC[i] = mul4(mul4(c0, is), c2);
      bD[i] = make_float4(bC[i].x * mu.x - c0.x * c1.x, bC[i].y * mu.y - c0.y * c1.y, bC[i].z * mu.z - c0.z * c1.z,
                          bC[i].w * mu.w - c0.w * c1.w);
    }
  }
  const float act_lo = p.act == DL3P_ACT_NONE ? -DL3P_INF : 0.f;
  const float act_hi = (p.act == DL3P_ACT_NONE || p.act == DL3P_ACT_RELU) ? DL3P_INF : 6.f;
  const char* Xb = reinterpret_cast<const char*>(p.X);
  const char* Db = reinterpret_cast<const char*>(p.DY);
  const char* Zb = reinterpret_cast<const char*>(p.Z);
  char* DZb = reinterpret_cast<char*>(p.DZ);
  const bool write_dz = BNA && p.DZ != nullptr && kt == 0;
  float4 rx[NX], rd[ND], rz[BNA ? ND : 1];
  auto gather_x = [&](int m0_) {
    gx_ok = 0;
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int m = m0_ + xr[i];
      const int mc = min(m, m_end - 1);
      int x, y;
      const int row = divmod_small(mc, p.g_RW, p.g_invRW, &x);
      const int n = divmod_small(row, p.g_RH, p.g_invRH, &y);
      const int sy = y * p.g_mul + gx_dy[i], sx = x * p.g_mul + gx_dx[i];
      const bool ok = m < m_end && sy >= 0 && sx >= 0 && sy < p.g_SH && sx < p.g_SW;
      const uint32_t off = ok ? (((uint32_t)n * (uint32_t)(p.g_SH * p.g_SW) + (uint32_t)(sy * p.g_SW + sx)) * (uint32_t)p.ldx) * 4u + xo[i] : 0u;
      rx[i] = *reinterpret_cast<const float4*>(Xb + off);
      gx_ok |= ok ? (1u << i) : 0u;
    }
  };
#define WT_PREFETCH(m0_)                                                                                              \
  {                                                                                                                   \
    if (GX) gather_x(m0_);                                                                                            \
    else _Pragma("unroll") for (int i = 0; i < NX; ++i)                                                               \
      rx[i] = *reinterpret_cast<const float4*>(Xb + ((uint32_t)min((m0_) + xr[i], m_end - 1) * (uint32_t)p.ldx * 4u + xo[i]));   \
    _Pragma("unroll") for (int i = 0; i < ND; ++i)                                                                    \
      rd[i] = *reinterpret_cast<const float4*>(Db + ((uint32_t)min((m0_) + dr[i], m_end - 1) * (uint32_t)p.lddy * 4u + dof[i])); \
    if (BNA) _Pragma("unroll") for (int i = 0; i < ND; ++i)                                                           \
      rz[i] = *reinterpret_cast<const float4*>(Zb + ((uint32_t)min((m0_) + dr[i], m_end - 1) * (uint32_t)p.ldz * 4u + dof[i])); \
  }
  f32x4 acc[KW][NW];
#pragma unroll
  for (int a = 0; a < KW; ++a)
#pragma unroll
    for (int b = 0; b < NW; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (m_begin < m_end) WT_PREFETCH(m_begin)
  for (int m0 = m_begin; m0 < m_end; m0 += 32) {
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      float4 v = fma4(rx[i], xsc[i], xsh[i]);
      if (p.act >= DL3P_ACT_HSWISH) v = act_apply4(v, p.act);
      else v = make_float4(__builtin_amdgcn_fmed3f(v.x, act_lo, act_hi), __builtin_amdgcn_fmed3f(v.y, act_lo, act_hi),
                           __builtin_amdgcn_fmed3f(v.z, act_lo, act_hi), __builtin_amdgcn_fmed3f(v.w, act_lo, act_hi));
      const bool ok = xok[i] && (GX ? ((gx_ok >> i) & 1u) != 0 : m0 + xr[i] < m_end);
      *reinterpret_cast<float4*>(&Xs[xr[i] * XP + (t + 256 * i - xr[i] * XQ) * 4]) = ok ? v : zero4();
    }
#pragma unroll
    for (int i = 0; i < ND; ++i) {
      const int idx = t + 256 * i;
      if (idx < 32 * DQ) {
        const bool ok = dok[i] && m0 + dr[i] < m_end;
        float4 v = make_float4(rd[i].x, rd[i].y, rd[i].z, rd[i].w);
        if (BNA) {
          const float4 z = rz[i];
          const float4 u = fma4(z, bsc[i], bsh[i]);
          const int act = p.b_act;
          v = make_float4(fmaf(bA[i].x, v.x * act_grad(u.x, act), fmaf(-bC[i].x, z.x, bD[i].x)),
                          fmaf(bA[i].y, v.y * act_grad(u.y, act), fmaf(-bC[i].y, z.y, bD[i].y)),
                          fmaf(bA[i].z, v.z * act_grad(u.z, act), fmaf(-bC[i].z, z.z, bD[i].z)),
                          fmaf(bA[i].w, v.w * act_grad(u.w, act), fmaf(-bC[i].w, z.w, bD[i].w)));
          if (write_dz && ok) st4(reinterpret_cast<float*>(DZb + ((uint32_t)(m0 + dr[i]) * (uint32_t)p.lddz * 4u + dof[i])), v);
        }
        *reinterpret_cast<float4*>(&Ds[dr[i] * DP + (idx - dr[i] * DQ) * 4]) = ok ? v : zero4();
      }
    }
    __syncthreads();
    if (m0 + 32 < m_end) WT_PREFETCH(m0 + 32)
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      float a[KW][4];
#pragma unroll
      for (int kw = 0; kw < KW; ++kw)
#pragma unroll
        for (int j = 0; j < 4; ++j) a[kw][j] = Xs[(g * 16 + q * 4 + j) * XP + (w * KW + kw) * 16 + l15];
#pragma unroll
      for (int ni = 0; ni < NW; ++ni) {
        float b[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = Ds[(g * 16 + q * 4 + j) * DP + ni * 16 + l15];
#pragma unroll
        for (int kw = 0; kw < KW; ++kw)
#pragma unroll
          for (int j = 0; j < 4; ++j)   // D[n][k]: lane ends with 4 consecutive n for k = l15
            acc[kw][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[j], a[kw][j], acc[kw][ni], 0, 0, 0);
      }
    }
    __syncthreads();
  }
#undef WT_PREFETCH
  float* slab = p.slabs + (size_t)blockIdx.y * p.K * p.N;
#pragma unroll
  for (int kw = 0; kw < KW; ++kw) {
    const int k = k0 + (w * KW + kw) * 16 + l15;
#pragma unroll
    for (int ni = 0; ni < NW; ++ni) {
      const int n = n0 + ni * 16 + q * 4;
      if (k < p.K && n < p.N)
        st4(slab + (size_t)k * p.N + n, make_float4(acc[kw][ni][0], acc[kw][ni][1], acc[kw][ni][2], acc[kw][ni][3]));
    }
  }
}

// ------------------------------------------------------------------------------ weight gradient, small K x N
// High-resolution layers have tiny kernels (16x96, 24x144, 32x192 ...) and millions of rows: the weight
// gradient is a pure stream over X and DY.  Here every WAVE owns the whole K x N gradient (KT x NTN
// accumulators) and walks its own 16-row tiles of M: it loads 16 whole rows of X and DY (contiguous
// 16-B lanes), writes them to a wave-private LDS slice, reads them back as MFMA fragments and multiplies.
// No workgroup barrier in the loop -- the 16 waves of a CU drift apart and cover each other's latencies --
// every input byte is read exactly once, and the next tile's loads are in flight during the MFMAs.
// The four waves of a workgroup are summed through LDS at the end (fixed order), one slab per workgroup.
template <int KT, int NTN, bool BNA = false>
__global__ __launch_bounds__(256, 2) void pw_wgrad_small_kernel(WgradParams p) {
  constexpr int KP = 16 * KT, NP = 16 * NTN;
  constexpr int XP = KP + 4, DP = NP + 4;          // pitches: rows 4 apart land 16 banks apart
  constexpr int WAVE_FLOATS = 16 * (XP + DP);
  constexpr int CF = BNA ? 5 * NP : 0;             // BNA: per-channel bA, bC, bD, scale, shift of the folded BatchNorm apply
  extern __shared__ __attribute__((aligned(16))) float ws_lds[];
  // layout: [scale KP][shift KP][BNA: 5 x NP coefficients][4 waves x WAVE_FLOATS]; the end-of-kernel reduction reuses it from 0
  float* sc_s = ws_lds;
  float* sh_s = ws_lds + KP;
  float* cf_s = ws_lds + 2 * KP;
  const int t = threadIdx.x, l = t & 63, w = t >> 6, l15 = l & 15, q = l >> 4;
  float* Xs = ws_lds + 2 * KP + CF + w * WAVE_FLOATS;
  float* Ds = Xs + 16 * XP;
  for (int i = t; i < KP; i += 256) {
    sc_s[i] = (p.scale && i < p.K) ? p.scale[i] : 1.f;
    sh_s[i] = (p.scale && i < p.K) ? p.shift[i] : 0.f;
  }
  if (BNA) {
    // dz = bA * g * act'(z * scale + shift) - bC * z + bD  (bA = c0, bC = c0 * invstd * c2, bD = bC * mean - c0 * c1)
    for (int i = t; i < NP; i += 256) {
      const bool in = i < p.N;
      const float c0 = in ? p.b_coef[i] : 0.f, c1 = in ? p.b_coef[p.N + i] : 0.f, c2 = in ? p.b_coef[2 * p.N + i] : 0.f;
      const float bc = in ? c0 * p.b_invstd[i] * c2 : 0.f;
      cf_s[i] = c0;
      cf_s[NP + i] = bc;
      cf_s[2 * NP + i] = in ? bc * p.b_mean[i] - c0 * c1 : 0.f;
      cf_s[3 * NP + i] = (in && p.b_scale) ? p.b_scale[i] : 1.f;
      cf_s[4 * NP + i] = (in && p.b_shift) ? p.b_shift[i] : 0.f;
    }
  }
  // columns K..KP-1 / N..NP-1 of the wave's slices are never loaded: zero them once
  for (int i = l; i < 16 * XP; i += 64) Xs[i] = 0.f;
  for (int i = l; i < 16 * DP; i += 64) Ds[i] = 0.f;
  __syncthreads();

  const int k4 = p.K >> 2, n4 = p.N >> 2;          // float4 per row
  const int nx = 4 * p.K, nd = 4 * p.N;            // float4 per 16-row tile
  // per-lane constants of the i-th load of a tile: row within the tile, LDS offset, global offset
  int xrow[KT], drow[NTN];
  uint32_t xg[KT], dg[NTN], zg[BNA ? NTN : 1], og[BNA ? NTN : 1];
  int xl[KT], dl[NTN];
#pragma unroll
  for (int i = 0; i < KT; ++i) {
    const int f = min(l + 64 * i, nx - 1);
    const int r = f / k4, c = f - r * k4;
    xrow[i] = (l + 64 * i < nx) ? r : 16;          // 16 = never valid
    xl[i] = r * XP + c * 4;
    xg[i] = ((uint32_t)r * (uint32_t)p.ldx + (uint32_t)c * 4u) * 4u;
  }
#pragma unroll
  for (int i = 0; i < NTN; ++i) {
    const int f = min(l + 64 * i, nd - 1);
    const int r = f / n4, c = f - r * n4;
    drow[i] = (l + 64 * i < nd) ? r : 16;
    dl[i] = r * DP + c * 4;
    dg[i] = ((uint32_t)r * (uint32_t)p.lddy + (uint32_t)c * 4u) * 4u;
    if (BNA) {
      zg[i] = ((uint32_t)r * (uint32_t)p.ldz + (uint32_t)c * 4u) * 4u;
      og[i] = ((uint32_t)r * (uint32_t)p.lddz + (uint32_t)c * 4u) * 4u;
    }
  }
  const float act_lo = p.act == DL3P_ACT_NONE ? -DL3P_INF : 0.f;
  const float act_hi = (p.act == DL3P_ACT_NONE || p.act == DL3P_ACT_RELU) ? DL3P_INF : 6.f;

  f32x4 acc[KT][NTN];
#pragma unroll
  for (int a = 0; a < KT; ++a)
#pragma unroll
    for (int b = 0; b < NTN; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int ntiles = (p.M + 15) >> 4;
  const int nwaves = gridDim.x * 4;
  const int gw = blockIdx.x * 4 + w;
  const char* Xb = reinterpret_cast<const char*>(p.X);
  const char* Db = reinterpret_cast<const char*>(p.DY);
  const char* Zb = reinterpret_cast<const char*>(p.Z);
  char* Ob = reinterpret_cast<char*>(p.DZ);
  float4 rx[KT], rd[NTN], rz[BNA ? NTN : 1];
  // (a macro, not a lambda: hipcc keeps a by-reference captured float4[] in scratch here)
#define WS_PREFETCH(tile_)                                                                          \
  {                                                                                                 \
    const int pm0 = (tile_) << 4;                                                                   \
    /* rows past M: the tile is loaded shifted up so every row is in bounds; see `back` below */    \
    const int pback = max(0, pm0 + 16 - p.M);                                                       \
    const uint32_t xb0 = (uint32_t)(pm0 - pback) * (uint32_t)p.ldx * 4u;                            \
    const uint32_t db0 = (uint32_t)(pm0 - pback) * (uint32_t)p.lddy * 4u;                           \
    _Pragma("unroll") for (int i = 0; i < KT; ++i) rx[i] = *reinterpret_cast<const float4*>(Xb + (xb0 + xg[i]));   \
    _Pragma("unroll") for (int i = 0; i < NTN; ++i) rd[i] = *reinterpret_cast<const float4*>(Db + (db0 + dg[i]));  \
    if (BNA) {                                                                                      \
      const uint32_t zb0 = (uint32_t)(pm0 - pback) * (uint32_t)p.ldz * 4u;                          \
      _Pragma("unroll") for (int i = 0; i < NTN; ++i) rz[i] = *reinterpret_cast<const float4*>(Zb + (zb0 + zg[i])); \
    }                                                                                               \
  }
  WS_PREFETCH(min(gw, ntiles - 1))
  for (int tile = gw; tile < ntiles; tile += nwaves) {
    const int m0 = tile << 4;
    const int back = max(0, m0 + 16 - p.M);        // the tile was loaded shifted up by `back` rows
    // stage: rows [0, back) of the shifted tile belong to the previous tile -> zero (X only: 0 * dy = 0)
#pragma unroll
    for (int i = 0; i < KT; ++i) {
      if (xrow[i] < 16) {
        const float4 s4 = *reinterpret_cast<const float4*>(&sc_s[xl[i] - xrow[i] * XP]);
        const float4 h4 = *reinterpret_cast<const float4*>(&sh_s[xl[i] - xrow[i] * XP]);
        float4 v = fma4(rx[i], s4, h4);
        if (p.act >= DL3P_ACT_HSWISH) v = act_apply4(v, p.act);
        else v = make_float4(__builtin_amdgcn_fmed3f(v.x, act_lo, act_hi), __builtin_amdgcn_fmed3f(v.y, act_lo, act_hi),
                             __builtin_amdgcn_fmed3f(v.z, act_lo, act_hi), __builtin_amdgcn_fmed3f(v.w, act_lo, act_hi));
        *reinterpret_cast<float4*>(&Xs[xl[i]]) = xrow[i] >= back ? v : zero4();
      }
    }
#pragma unroll
    for (int i = 0; i < NTN; ++i)
      if (drow[i] < 16) {
        float4 v = make_float4(rd[i].x, rd[i].y, rd[i].z, rd[i].w);
        if (BNA) {
          const int cf = dl[i] - drow[i] * DP;       // channel of this float4
          const float4 a4 = *reinterpret_cast<const float4*>(&cf_s[cf]);
          const float4 c4 = *reinterpret_cast<const float4*>(&cf_s[NP + cf]);
          const float4 d4 = *reinterpret_cast<const float4*>(&cf_s[2 * NP + cf]);
          const float4 s4 = *reinterpret_cast<const float4*>(&cf_s[3 * NP + cf]);
          const float4 h4 = *reinterpret_cast<const float4*>(&cf_s[4 * NP + cf]);
          const float4 z = rz[i];
          const float4 u = fma4(z, s4, h4);
          const int act = p.b_act;
          v = make_float4(fmaf(a4.x, v.x * act_grad(u.x, act), fmaf(-c4.x, z.x, d4.x)),
                          fmaf(a4.y, v.y * act_grad(u.y, act), fmaf(-c4.y, z.y, d4.y)),
                          fmaf(a4.z, v.z * act_grad(u.z, act), fmaf(-c4.z, z.z, d4.z)),
                          fmaf(a4.w, v.w * act_grad(u.w, act), fmaf(-c4.w, z.w, d4.w)));
          // (rows [0, back) of a shifted last tile were written by the tile before it)
          if (p.DZ && drow[i] >= back)
            st4(reinterpret_cast<float*>(Ob + ((uint32_t)(m0 - back) * (uint32_t)p.lddz * 4u + og[i])), v);
        }
        *reinterpret_cast<float4*>(&Ds[dl[i]]) = v;
      }
    WS_PREFETCH(min(tile + nwaves, ntiles - 1))   // unconditional (the last one is a harmless re-read)
    // fragments: reduction index m = 4q + j; lane l15 = channel within the 16-wide tile
    float b[NTN][4];
#pragma unroll
    for (int nt = 0; nt < NTN; ++nt)
#pragma unroll
      for (int j = 0; j < 4; ++j) b[nt][j] = Ds[(4 * q + j) * DP + nt * 16 + l15];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      float a[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) a[j] = Xs[(4 * q + j) * XP + kt * 16 + l15];
#pragma unroll
      for (int nt = 0; nt < NTN; ++nt)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[kt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[nt][j], a[j], acc[kt][nt], 0, 0, 0);
    }
  }
  // sum the four waves (fixed order 0+1+2+3) and write this workgroup's slab
  __syncthreads();
  float4* red = reinterpret_cast<float4*>(ws_lds);
  if (w > 0) {
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int nt = 0; nt < NTN; ++nt)
        red[((w - 1) * KT * NTN + kt * NTN + nt) * 64 + l] =
            make_float4(acc[kt][nt][0], acc[kt][nt][1], acc[kt][nt][2], acc[kt][nt][3]);
  }
  __syncthreads();
  if (w == 0) {
    float* slab = p.slabs + (size_t)blockIdx.x * p.K * p.N;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int nt = 0; nt < NTN; ++nt) {
        float4 v = make_float4(acc[kt][nt][0], acc[kt][nt][1], acc[kt][nt][2], acc[kt][nt][3]);
#pragma unroll
        for (int ww = 0; ww < 3; ++ww) v = add4(v, red[(ww * KT * NTN + kt * NTN + nt) * 64 + l]);
        const int k = kt * 16 + l15, n = nt * 16 + q * 4;
        if (k < p.K && n < p.N) st4(slab + (size_t)k * p.N + n, v);
      }
  }
}

template <int KT, int NTN, bool BNA = false>
static constexpr size_t wgrad_small_lds() {
  constexpr size_t stage = sizeof(float) * (size_t)(2 * 16 * KT + (BNA ? 5 * 16 * NTN : 0) + 4 * 16 * (16 * KT + 4 + 16 * NTN + 4));
  constexpr size_t red = 16 * (size_t)(3 * KT * NTN * 64);
  return stage > red ? stage : red;
}

// workgroups for the small-K.N kernel (all resident at once)
static int wgrad_small_grid(int M, int KT, int NTN) {
  const int tiles = ceil_div(M, 16);
  // measured (kernel + slab reduce): two workgroups per CU stream as fast as four and halve the slabs;
  // below two row tiles per wave the per-wave prologue / reduction dominates
  static const int occ_env = getenv("DL3P_WGRAD_SMALL_PER_CU") ? atoi(getenv("DL3P_WGRAD_SMALL_PER_CU")) : 2;
  const int occ = occ_env, tpw = 2;
  (void)KT; (void)NTN;
  int g = tiles / (4 * tpw);
  if (g > DL3P_NUM_CUS * occ) g = DL3P_NUM_CUS * occ;
  if (g > DL3P_MAX_STAT_ROWS) g = DL3P_MAX_STAT_ROWS;
  if (g < 1) g = 1;
  return g;
}

template <int KT, int NTN, bool BNA = false>
static void launch_wgrad_small(const WgradParams& p, int grid, hipStream_t st) {
  constexpr size_t lds = wgrad_small_lds<KT, NTN, BNA>();
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)pw_wgrad_small_kernel<KT, NTN, BNA>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  dl3p_launch(pw_wgrad_small_kernel<KT, NTN, BNA>, dim3(grid), dim3(256), lds, st, p);
}

// the (KT, NTN) instantiations: kernels of the 513x513 MobileNetV2 / V3 / Xception graphs at OS 2-8
static bool wgrad_small_pick(int K, int N, SmallShape* out) {
  static const SmallShape list[] = {{1, 2}, {2, 1}, {2, 2}, {1, 6}, {2, 3}, {2, 4}, {4, 2}, {6, 2}, {2, 6}, {2, 9}, {9, 2},
                                    {2, 12}, {12, 2}, {4, 4}};
  static const int off = getenv("DL3P_WGRAD_SMALL") ? atoi(getenv("DL3P_WGRAD_SMALL")) == 0 : 0;
  if (off) return false;
  const int kt = ceil_div(K, 16), ntn = ceil_div(N, 16);
  int best = -1, best_tiles = 1 << 30;
  for (int i = 0; i < (int)(sizeof(list) / sizeof(list[0])); ++i)
    if (list[i].kt >= kt && list[i].ntn >= ntn && list[i].kt * list[i].ntn < best_tiles) { best = i; best_tiles = list[i].kt * list[i].ntn; }
  if (best < 0 || best_tiles > 2 * kt * ntn) return false;   // too much padding: use the tiled kernel
  *out = list[best];
  return true;
}

template <bool BNA = false>
static void launch_wgrad_small_any(const WgradParams& p, SmallShape sh, int grid, hipStream_t st) {
#define DL3P_WS(a, b) if (sh.kt == a && sh.ntn == b) { launch_wgrad_small<a, b, BNA>(p, grid, st); return; }
  DL3P_WS(1, 2) DL3P_WS(2, 1) DL3P_WS(2, 2) DL3P_WS(1, 6) DL3P_WS(2, 3) DL3P_WS(2, 4) DL3P_WS(4, 2) DL3P_WS(6, 2)
  DL3P_WS(2, 6) DL3P_WS(2, 9) DL3P_WS(9, 2) DL3P_WS(2, 12) DL3P_WS(12, 2) DL3P_WS(4, 4)
#undef DL3P_WS
}

// tile shape (KW, NW) -> (64 KW) x (16 NW): fewest padded MFMA columns, weighted by the operand re-reads
static void wgrad_pick_tile(int M, int K, int N, int* kw, int* nw) {
  static const int cand[4][2] = {{1, 4}, {2, 4}, {1, 8}, {2, 8}};
  static const int env_force = getenv("DL3P_WGRAD_TILE") ? atoi(getenv("DL3P_WGRAD_TILE")) : -1;
  int force = env_force;
  // role 4 of the measured table: nt = tile index (0: 64x64, 1: 128x64, 2: 64x128, 3: 128x128), mi = workgroups per CU
  if (const GemmTuned* e = gemm_tuned_lookup(4, M, K, N)) force = e->nt;
  if (g_wgrad_force_tile >= 0) force = g_wgrad_force_tile;
  // measured: larger tiles pay only when M is large (decoder layers: 64 x 128 is 8-10 % faster than 64 x 64);
  // on the 17424-row layers they cut the number of workgroups too far
  if (force < 0 && M < 65536) { *kw = 1; *nw = 4; return; }
  float best = 1e30f;
  for (int i = 0; i < 4; ++i) {
    if (force >= 0 && i != force) continue;
    const int tk = 64 * cand[i][0], tn = 16 * cand[i][1];
    const float area = (float)(ceil_div(K, tk) * tk) * (float)(ceil_div(N, tn) * tn);
    const float cost = area * (1.f + 0.5f * (64.f / tk + 64.f / tn));
    if (cost < best) { best = cost; *kw = cand[i][0]; *nw = cand[i][1]; }
  }
}

static void wgrad_split(int M, int K, int N, int* ktiles, int* ntiles, int* splits, int* mchunk) {
  int kw, nw;
  wgrad_pick_tile(M, K, N, &kw, &nw);
  *ktiles = ceil_div(K, 64 * kw);
  *ntiles = ceil_div(N, 16 * nw);
  const int tiles = *ktiles * *ntiles;
  static const int env_per_cu = getenv("DL3P_WGRAD_PER_CU") ? atoi(getenv("DL3P_WGRAD_PER_CU")) : 4;
  int per_cu = env_per_cu;
  if (const GemmTuned* e = gemm_tuned_lookup(4, M, K, N)) per_cu = e->mi;
  if (g_wgrad_force_per_cu) per_cu = g_wgrad_force_per_cu;
  int s = (DL3P_NUM_CUS * per_cu) / tiles;
  if (s < 1) s = 1;
  int max_s = ceil_div(M, 256);          // at least 256 rows per slice
  if (s > max_s) s = max_s;
  if (s > DL3P_MAX_STAT_ROWS) s = DL3P_MAX_STAT_ROWS;
  int chunk = ceil_div(ceil_div(M, s), 32) * 32;
  *splits = ceil_div(M, chunk);
  *mchunk = chunk;
}

template <bool GX = false, bool BNA = false>
static void launch_wgrad_tiled(const WgradParams& p, int splits, hipStream_t st) {
  int kw, nw;
  wgrad_pick_tile(p.M, p.K, p.N, &kw, &nw);
  const dim3 grid(p.ktiles * p.ntiles, splits), block(256);
  if (kw == 1 && nw == 4) dl3p_launch(pw_wgrad_kernel<1, 4, GX, BNA>, grid, block, 0, st, p);
  else if (kw == 2 && nw == 4) dl3p_launch(pw_wgrad_kernel<2, 4, GX, BNA>, grid, block, 0, st, p);
  else if (kw == 1 && nw == 8) dl3p_launch(pw_wgrad_kernel<1, 8, GX, BNA>, grid, block, 0, st, p);
  else dl3p_launch(pw_wgrad_kernel<2, 8, GX, BNA>, grid, block, 0, st, p);
}

// column sums of dy (bias gradient): one partial row per workgroup
__global__ __launch_bounds__(256) void colsum_kernel(const float* dy, int lddy, long long M, int C, int c4s, int px,
                                                     int nbx, float* partials) {
  const int b = blockIdx.x;
  const int slab = b / nbx;
  const int bx = b - slab * nbx;
  const int pl = threadIdx.x / c4s;
  const int cl = threadIdx.x - pl * c4s;
  const bool active = pl < px;
  const int cbase4 = slab * c4s;
  const int c = (cbase4 + cl) * 4;
  float4 acc[1] = {zero4()};
  if (active)
    for (long long m = (long long)bx * px + pl; m < M; m += (long long)nbx * px) acc[0] = add4(acc[0], ld4(dy + (size_t)m * lddy + c));
  block_reduce_store<1>(acc, active, pl, cl, c4s, px, cbase4, C, partials + (size_t)bx * C);
}

extern "C" size_t dl3p_pwconv_bwd_weight_workspace(int M, int K, int N) {
  if (M <= 0 || K <= 0 || N <= 0) return 0;
  int kt, nt, s, mc;
  wgrad_split(M, K, N, &kt, &nt, &s, &mc);
  size_t a = (size_t)s * K * N;
  SmallShape sh;
  if (wgrad_small_pick(K, N, &sh)) a = (size_t)wgrad_small_grid(M, sh.kt, sh.ntn) * K * N;
  else {
    int kf, nw, kt2, nt2, mrows;
    const size_t s2 = (size_t)wgrad_sb_route(M, K, N, DL3P_MAX_STAT_ROWS, &kf, &nw, &kt2, &nt2, &mrows) * K * N;
    if (s2 > a) a = s2;
  }
  size_t b = (size_t)512 * N;  // bias column-sum partial rows
  return (a > b ? a : b) * sizeof(float);
}

// rows_out != NULL: leave the slabs in the workspace for dl3p_reduce_rows_batched (gw / gb unused) and report how many
static int pwconv_bwd_weight_impl(const float* x, int ldx, const float* in_scale, const float* in_shift,
                                  int in_act, const float* dy, int lddy, float* gw, float* gb, float* workspace,
                                  size_t workspace_bytes, int M, int K, int N, int* rows_out, void* stream) {
  int rc = check_mat("dl3p_pwconv_bwd_weight", x, ldx, K);
  if (rc) return rc;
  rc = check_mat("dl3p_pwconv_bwd_weight", dy, lddy, N);
  if (rc) return rc;
  DL3P_CHECK_ARG((gw || rows_out) && workspace && aligned16(workspace) && M > 0, "dl3p_pwconv_bwd_weight: bad arguments");
  DL3P_CHECK_ARG(!rows_out || (!gb && !dl3p_pw_tiny_applies(M)),
                 "dl3p_pwconv_bwd_weight_slabs: no bias gradient and more than %d rows (use dl3p_pwconv_bwd_weight)", 64);
  const size_t need = dl3p_pwconv_bwd_weight_workspace(M, K, N);
  if (workspace_bytes < need) {
    dl3p_set_error("dl3p_pwconv_bwd_weight: workspace %zu < %zu bytes", workspace_bytes, need);
    return DL3P_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  if (dl3p_pw_tiny_applies(M)) {
    DL3P_CHECK_ARG(aligned16(gw) && (!gb || aligned16(gb)), "dl3p_pwconv_bwd_weight: gw/gb must be 16-byte aligned");
    dl3p_pw_tiny_wgrad(x, ldx, in_scale, in_shift, in_act, dy, lddy, gw, gb, M, K, N, st);
    DL3P_CHECK_LAUNCH("dl3p_pwconv_bwd_weight");
    return DL3P_OK;
  }
  WgradParams p = {};
  p.X = x; p.ldx = ldx; p.scale = in_scale; p.shift = in_shift; p.act = in_act;
  p.DY = dy; p.lddy = lddy; p.slabs = workspace; p.M = M; p.K = K; p.N = N;
  int splits;
  SmallShape sh;
  if (M >= 16 && wgrad_small_pick(K, N, &sh) && (unsigned long long)M * (unsigned long long)(ldx > lddy ? ldx : lddy) * 4ull < (1ull << 32)) {
    splits = wgrad_small_grid(M, sh.kt, sh.ntn);
    launch_wgrad_small_any(p, sh, splits, st);
  } else {
    // fp32-accurate on the bf16 matrix pipe (pw_split.hip, pw_wgrad_sb_kernel): both operands split while they are staged
    int kf, nw, kt2, nt2, mrows, s2 = 0;
    s2 = wgrad_sb_route(M, K, N, workspace_bytes / ((size_t)K * N * 4), &kf, &nw, &kt2, &nt2, &mrows);
    if (s2 > 0) {
      splits = s2;
      dl3p_launch_wgrad_sb(x, ldx, in_scale, in_shift, in_act, dy, lddy, workspace, M, K, N, kf, nw, kt2, nt2, mrows, s2, st);
    } else {
      wgrad_split(M, K, N, &p.ktiles, &p.ntiles, &splits, &p.mchunk);
      launch_wgrad_tiled<false>(p, splits, st);
    }
  }
  DL3P_CHECK_LAUNCH("dl3p_pwconv_bwd_weight");
  if (rows_out) { *rows_out = splits; return DL3P_OK; }
  rc = dl3p_reduce_rows_impl(workspace, splits, (size_t)K * N, gw, 0, st);
  if (rc) return rc;
  if (gb) {
    int c4s, px, nslab;
    pick_lanes(N, &c4s, &px, &nslab);
    long long need_b = ceil_div_ll(M, px);
    int nbx = (int)(need_b < 512 ? need_b : 512);
    hipLaunchKernelGGL(colsum_kernel, dim3(nbx * nslab), dim3(256), 0, st, dy, lddy, (long long)M, N, c4s, px, nbx,
                       workspace);
    DL3P_CHECK_LAUNCH("dl3p_pwconv_bwd_weight(colsum)");
    rc = dl3p_reduce_rows_impl(workspace, nbx, (size_t)N, gb, 0, st);
  }
  return rc;
}

extern "C" int dl3p_pwconv_bwd_weight(const float* x, int ldx, const float* in_scale, const float* in_shift,
                                      int in_act, const float* dy, int lddy, float* gw, float* gb, float* workspace,
                                      size_t workspace_bytes, int M, int K, int N, void* stream) {
  return pwconv_bwd_weight_impl(x, ldx, in_scale, in_shift, in_act, dy, lddy, gw, gb, workspace, workspace_bytes, M, K, N,
                                nullptr, stream);
}

extern "C" int dl3p_pwconv_bwd_weight_slabs(const float* x, int ldx, const float* in_scale, const float* in_shift,
                                            int in_act, const float* dy, int lddy, float* workspace, size_t workspace_bytes,
                                            int* rows_out, int M, int K, int N, void* stream) {
  DL3P_CHECK_ARG(rows_out != nullptr, "dl3p_pwconv_bwd_weight_slabs: rows_out is required");
  return pwconv_bwd_weight_impl(x, ldx, in_scale, in_shift, in_act, dy, lddy, nullptr, nullptr, workspace, workspace_bytes, M, K,
                                N, rows_out, stream);
}


// Weight gradient of a conv whose output z goes through BatchNorm (+ activation), with that BatchNorm's backward apply
// folded in: `g` is the gradient of act(BN(z)), `coef` what dl3p_bn_bwd_finalize left.  The kernel forms dz while it stages
// its tiles, multiplies with it, and (dz != nullptr) writes it for the data gradient that follows -- dz must not alias g.
// Served where the kernel reads the gradient operand ONCE: the streaming kernels of the few-channel layers (every wave
// owns the whole K x N gradient) and tiled launches with a single k tile.  With several k tiles every one of them would
// re-form dz from two operands instead of reading one: measured 46 % slower per launch on the 17424 x 64..960 layers
// and a net loss per step (14.64 against 14.32 ms), so those shapes keep dl3p_bn_bwd_apply.
static int wgrad_bn_route(int M, int K, int N) {      // 0: not served, 1: streaming kernel, 2: tiled kernel, one k tile
  SmallShape sh;
  if (M <= 64 || dl3p_pw_tiny_applies(M) || N % 4 || K % 4) return 0;
  if (M >= 16 && wgrad_small_pick(K, N, &sh)) return (sh.kt == 2 && sh.ntn == 12) ? 0 : 1;   // (2, 12) + the fold spills
  int kw, nw;
  wgrad_pick_tile(M, K, N, &kw, &nw);
  return (K <= 64 * kw && !(kw == 2 && nw == 8)) ? 2 : 0;      // (the 128 x 128 tile has no registers left for the fold)
}

extern "C" int dl3p_pwconv_bwd_weight_bn_supported(int M, int K, int N) { return wgrad_bn_route(M, K, N) != 0; }

extern "C" int dl3p_pwconv_bwd_weight_slabs_bn(const float* x, int ldx, const float* in_scale, const float* in_shift,
                                               int in_act, const float* g, int ldg, const float* z, int ldz,
                                               const float* bn_scale, const float* bn_shift, int bn_act,
                                               const float* save_mean, const float* save_invstd, const float* coef,
                                               float* dz, int lddz, float* workspace, size_t workspace_bytes,
                                               int* rows_out, int M, int K, int N, void* stream) {
  const char* fn = "dl3p_pwconv_bwd_weight_slabs_bn";
  int rc = check_mat(fn, x, ldx, K);
  if (rc) return rc;
  rc = check_mat(fn, g, ldg, N);
  if (rc) return rc;
  rc = check_mat(fn, z, ldz, N);
  if (rc) return rc;
  if (dz) {
    rc = check_mat(fn, dz, lddz, N);
    if (rc) return rc;
  }
  DL3P_CHECK_ARG(rows_out && workspace && aligned16(workspace) && save_mean && save_invstd && coef && dz != g,
                 "%s: bad arguments", fn);
  DL3P_CHECK_ARG(dl3p_pwconv_bwd_weight_bn_supported(M, K, N), "%s: shape M=%d K=%d N=%d is not served by the tiled kernel", fn,
                 M, K, N);
  const int ldmax = ldx > ldg ? (ldx > ldz ? ldx : ldz) : (ldg > ldz ? ldg : ldz);
  DL3P_CHECK_ARG((unsigned long long)M * (unsigned long long)(ldmax > lddz ? ldmax : lddz) * 4ull < (1ull << 32),
                 "%s: operands of 4 GiB or more are not supported (M=%d)", fn, M);
  const size_t need = dl3p_pwconv_bwd_weight_workspace(M, K, N);
  if (workspace_bytes < need) {
    dl3p_set_error("%s: workspace %zu < %zu bytes", fn, workspace_bytes, need);
    return DL3P_EWORKSPACE;
  }
  WgradParams p = {};
  p.X = x; p.ldx = ldx; p.scale = in_scale; p.shift = in_shift; p.act = in_act;
  p.DY = g; p.lddy = ldg; p.slabs = workspace; p.M = M; p.K = K; p.N = N;
  p.Z = z; p.ldz = ldz; p.b_scale = bn_scale; p.b_shift = bn_shift; p.b_mean = save_mean; p.b_invstd = save_invstd;
  p.b_coef = coef; p.b_act = bn_act; p.DZ = dz; p.lddz = lddz;
  int splits;
  SmallShape sh;
  if (wgrad_bn_route(M, K, N) == 1 && wgrad_small_pick(K, N, &sh)) {
    splits = wgrad_small_grid(M, sh.kt, sh.ntn);
    launch_wgrad_small_any<true>(p, sh, splits, (hipStream_t)stream);
  } else {
    wgrad_split(M, K, N, &p.ktiles, &p.ntiles, &splits, &p.mchunk);
    launch_wgrad_tiled<false, true>(p, splits, (hipStream_t)stream);
  }
  DL3P_CHECK_LAUNCH(fn);
  *rows_out = splits;
  return DL3P_OK;
}


// ------------------------------------------------------------------------------ dense convolutions as implicit GEMMs
// k x k convolutions with Cin % 4 == 0 (Xception entry_flow_conv1_2 3x3 32->64 and its strided 1x1 shortcuts,
// deeplabv3p_xception.py:119-127,175-183; ResNet50's 3x3 / strided convs) on the tiled GEMM kernels above with the patch
// operand GATHERED while the A tile is staged into LDS (north_star "LDS-staged im2col tiles"): no [M][k*k*Cin] matrix in
// HBM, no im2col / col2im pass.  Four consecutive k are four channels of one tap (Cin % 4 == 0), so the gather keeps the
// 16-byte loads of the pointwise path; taps in the padding are selected to zero after the prologue.
static int conv_gemm_check(const char* fn, int N, int H, int W, int Cin, int Cout, int k, int stride, int rate, int pad_t,
                           int pad_l, int Ho, int Wo) {
  DL3P_CHECK_ARG(dl3p_conv2d_gemm_supported(Cin, Cout, k, stride), "%s: unsupported conv (Cin=%d Cout=%d k=%d stride=%d)", fn,
                 Cin, Cout, k, stride);
  DL3P_CHECK_ARG(N > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0 && rate > 0 && pad_t >= 0 && pad_l >= 0, "%s: bad geometry", fn);
  DL3P_CHECK_ARG((long long)N * H * W < (1ll << 24) && (long long)N * Ho * Wo < (1ll << 24) && H < 16384 && W < 16384,
                 "%s: tensor too large", fn);
  DL3P_CHECK_ARG((Ho - 1) * stride - pad_t < H && (Wo - 1) * stride - pad_l < W, "%s: output larger than the strided input", fn);
  return DL3P_OK;
}

extern "C" int dl3p_conv2d_gemm_supported(int Cin, int Cout, int k, int stride) {
  static const int enabled = getenv("DL3P_CONV_GEMM") ? atoi(getenv("DL3P_CONV_GEMM")) : 1;
  return enabled && Cin > 0 && Cout > 0 && Cin % 4 == 0 && Cout % 4 == 0 && k >= 1 && k <= 7 && (stride == 1 || stride == 2) &&
         (long long)k * k * Cin < 65536;
}

static void conv_gather_fwd(GemmParams* p, int H, int W, int Cin, int k, int stride, int rate, int pad_t, int pad_l, int Ho,
                            int Wo) {
  p->g_RH = Ho; p->g_RW = Wo; p->g_SH = H; p->g_SW = W; p->g_C = Cin; p->g_kw = k;
  p->g_mul = stride; p->g_ay = -pad_t; p->g_ax = -pad_l; p->g_d = rate; p->g_shift = 0;
  p->g_cmagic = (uint32_t)((1ull << 32) / (unsigned)Cin) + 1u;
  p->g_kwmagic = 65536 / k + 1;
}

extern "C" int dl3p_conv2d_gemm_fwd(const float* x, int ldx, const float* in_scale, const float* in_shift, int in_act,
                                    const float* wt, const float* bias, float* y, int ldy, float* stat_partials,
                                    int* rows_out, int N, int H, int W, int Cin, int Cout, int k, int stride, int rate,
                                    int pad_t, int pad_l, int Ho, int Wo, void* stream) {
  const char* fn = "dl3p_conv2d_gemm_fwd";
  int rc = conv_gemm_check(fn, N, H, W, Cin, Cout, k, stride, rate, pad_t, pad_l, Ho, Wo);
  if (rc) return rc;
  rc = check_mat(fn, x, ldx, Cin);
  if (rc) return rc;
  rc = check_mat(fn, y, ldy, Cout);
  if (rc) return rc;
  DL3P_CHECK_ARG(wt && aligned16(wt), "%s: bad kernel pointer", fn);
  const int M = N * Ho * Wo, K = k * k * Cin;
  DL3P_CHECK_ARG((unsigned long long)N * H * W * (unsigned long long)ldx * 4ull < (1ull << 32) &&
                     (unsigned long long)M * (unsigned long long)ldy * 4ull < (1ull << 32),
                 "%s: operands of 4 GiB or more are not supported", fn);
  GemmParams p = {};
  p.A = x; p.lda = ldx; p.scale = in_scale; p.shift = in_shift; p.act = in_act;
  p.B = wt; p.ldb = K; p.bias = bias; p.Y = y; p.ldy = ldy; p.partials = stat_partials;
  p.M = M; p.K = K; p.N = Cout;
  conv_gather_fwd(&p, H, W, Cin, k, stride, rate, pad_t, pad_l, Ho, Wo);
  const int nt = pick_nt(Cout, M);
  int gx, gy, mi;
  gemm_grid(M, Cout, nt, &gx, &gy, &p.num_m_tiles, &mi);
  if (rows_out) *rows_out = gx;
  if (stat_partials) launch_gemm<false, true, false, true>(p, nt, mi, dim3(gx, gy), (hipStream_t)stream);
  else launch_gemm<false, false, false, true>(p, nt, mi, dim3(gx, gy), (hipStream_t)stream);
  DL3P_CHECK_LAUNCH(fn);
  return DL3P_OK;
}

// wd[ci][tap * Cout + co] = w[(tap * Cin + ci) * Cout + co]: the kernel as the [Nout = Cin][Kred = taps * Cout] operand of
// the data-gradient GEMM (one launch per step and conv; the kernels are a few hundred KB)
__global__ void conv_dgrad_weights_kernel(const float* w, float* wd, int taps, int Cin, int Cout) {
  const int total = taps * Cin * Cout;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int co = i % Cout, r = i / Cout;
    const int ci = r % Cin, tap = r / Cin;
    wd[((size_t)ci * taps + tap) * Cout + co] = w[i];
  }
}

extern "C" int dl3p_conv2d_gemm_dgrad_weights(const float* w, float* wd, int k, int Cin, int Cout, void* stream) {
  DL3P_CHECK_ARG(w && wd && k >= 1 && Cin > 0 && Cout > 0, "dl3p_conv2d_gemm_dgrad_weights: bad arguments");
  const int total = k * k * Cin * Cout;
  hipLaunchKernelGGL(conv_dgrad_weights_kernel, dim3(ceil_div(total, 256) < 1024 ? ceil_div(total, 256) : 1024), dim3(256), 0,
                     (hipStream_t)stream, w, wd, k * k, Cin, Cout);
  DL3P_CHECK_LAUNCH("dl3p_conv2d_gemm_dgrad_weights");
  return DL3P_OK;
}

extern "C" int dl3p_conv2d_gemm_bwd_data(const float* dy, int lddy, const float* wd, float* gx, int ldgx, int accumulate,
                                         int N, int H, int W, int Cin, int Cout, int k, int stride, int rate, int pad_t,
                                         int pad_l, int Ho, int Wo, void* stream) {
  const char* fn = "dl3p_conv2d_gemm_bwd_data";
  int rc = conv_gemm_check(fn, N, H, W, Cin, Cout, k, stride, rate, pad_t, pad_l, Ho, Wo);
  if (rc) return rc;
  rc = check_mat(fn, dy, lddy, Cout);
  if (rc) return rc;
  rc = check_mat(fn, gx, ldgx, Cin);
  if (rc) return rc;
  DL3P_CHECK_ARG(wd && aligned16(wd) && (long long)k * k * Cout < 65536, "%s: bad kernel operand", fn);
  const int M = N * H * W, K = k * k * Cout;
  DL3P_CHECK_ARG((unsigned long long)N * Ho * Wo * (unsigned long long)lddy * 4ull < (1ull << 32) &&
                     (unsigned long long)M * (unsigned long long)ldgx * 4ull < (1ull << 32),
                 "%s: operands of 4 GiB or more are not supported", fn);
  GemmParams p = {};
  p.A = dy; p.lda = lddy; p.act = DL3P_ACT_NONE;
  p.B = wd; p.ldb = K; p.Y = gx; p.ldy = ldgx; p.accumulate = accumulate;
  p.M = M; p.K = K; p.N = Cin;
  // rows = input pixels; the tap (ky, kx) of input pixel (iy, ix) is output pixel ((iy + pad_t - ky*rate) / stride, ...)
  p.g_RH = H; p.g_RW = W; p.g_SH = Ho; p.g_SW = Wo; p.g_C = Cout; p.g_kw = k;
  p.g_mul = 1; p.g_ay = pad_t; p.g_ax = pad_l; p.g_d = -rate; p.g_shift = stride == 2 ? 1 : 0;
  p.g_cmagic = (uint32_t)((1ull << 32) / (unsigned)Cout) + 1u;
  p.g_kwmagic = 65536 / k + 1;
  const int nt = pick_nt(Cin, M);
  int gxn, gy, mi;
  gemm_grid(M, Cin, nt, &gxn, &gy, &p.num_m_tiles, &mi);
  launch_gemm<false, false, false, true>(p, nt, mi, dim3(gxn, gy), (hipStream_t)stream);
  DL3P_CHECK_LAUNCH(fn);
  return DL3P_OK;
}

// ---- the same implicit GEMMs on the split-bf16 kernels (pw_split.hip, GA instantiations of pw_gemm_sb_kernel / GX of
// pw_wgrad_sb_kernel): fp32-accurate products on the bf16 matrix pipe, the kernel pre-split by dl3p_split_bf16x3_batch as
// [3][Cout][pitch >= k k Cin] (forward: from the transposed kernel wt) or [3][Cin][pitch >= k k Cout] (data gradient: from
// dl3p_conv2d_gemm_dgrad_weights' wd).  role 0 / 1: forward without / with BatchNorm statistics, 2: data gradient, 4: weight
// gradient; (M, K, N) = the GEMM as launched.
static int conv_sb_mode() {       // 0 off, 1 the measured rule, 2 wherever supported (tests)
  static const int env = getenv("DL3P_CONV_SB") ? atoi(getenv("DL3P_CONV_SB")) : 1;
  return g_conv_sb >= 0 ? g_conv_sb : env;
}

extern "C" int dl3p_conv2d_gemm_sb_supported(int role, int M, int K, int N) {
  if (role < 0 || role > 4 || role == 3 || conv_sb_mode() == 0) return 0;
  if (M < 1024 || K < 32 || N < 16 || K % 4 || N % 4 || K >= 65536) return 0;
  if (role == 4) return split_wgrad_on() && N >= 32;
  return 1;
}

// where it pays (scripts/micro/conv_sb.py, profiles/r04_dense_conv_split.txt)
extern "C" int dl3p_conv2d_gemm_sb_pays(int role, int M, int K, int N) {
  if (!dl3p_conv2d_gemm_sb_supported(role, M, K, N)) return 0;
  if (conv_sb_mode() == 2) return 1;
  // measured (same box, fp32-input kernel -> split kernel, us): the WEIGHT gradient wins wherever both of its operands are long
  // enough to fill the tiles -- 125 -> 111 (Xception entry_flow_conv1_2, 264196 x 288 x 64), 124 -> 101 (ResNet50 stage 2), 115 ->
  // 75 / 119 -> 78 (stages 3 / 4), 626 -> 326 (stage 5 at 8712 rows: three slabs instead of one).  The FORWARD and the DATA
  // gradient are bound by the gather of their A operand, not by the matrix pipe: they win with a long reduction (K >= 1024: 141 ->
  // 126, 175 -> 131, 518 -> 460 forward; 132 -> 123, 163 -> 129, 493 -> 443 data gradient); at K = 576 onto 64 columns both lose
  // (142 -> 147, 123 -> 132), as does conv1_2's data gradient onto 32 columns (163 -> 166).  conv1_2's forward would win (150 ->
  // 127 at batch 4, 170 -> 153 at configs[3]) and stays on the fp32-input kernel all the same: its column means come out 3e-8 of
  // a standard deviation off instead of 5e-9 (same rms error per element, 2e-7) -- as the second layer of a 70-layer network on
  // batch statistics that moved every gradient of the 513 x 513 Xception parity test 1.7x further from float64 (0.0048 -> 0.0058
  // worst) for 0.03 ms of a 21.6 ms step
  if (M < 4096) return 0;
  if (role == 4) return M >= 8192 && K >= 128;
  return K >= 1024;
}

static void gemm_plan_sb_ga(int M, int N, bool stats, int* nt, int* gx, int* gy, int* num_m_tiles, int* mi) {
  *nt = pick_nt(N, M);
  if (*nt > 8) *nt = 8;
  if (g_gemm_force_nt) *nt = g_gemm_force_nt > 8 ? 8 : g_gemm_force_nt;
  // (128-row tiles with the gather's index registers spill from 96 columns up: 64-row tiles there)
  gemm_grid(M, N, *nt, gx, gy, num_m_tiles, mi, false, g_gemm_force_mi ? g_gemm_force_mi : ((M >= 4096 && *nt < 6) ? 2 : 1), g_gemm_force_pc);
  (void)stats;
}

extern "C" int dl3p_conv2d_gemm_fwd_sb(const float* x, int ldx, const float* in_scale, const float* in_shift, int in_act,
                                       const void* wsp, int pitch, const float* bias, float* y, int ldy, float* stat_partials,
                                       int* rows_out, int N, int H, int W, int Cin, int Cout, int k, int stride, int rate,
                                       int pad_t, int pad_l, int Ho, int Wo, void* stream) {
  const char* fn = "dl3p_conv2d_gemm_fwd_sb";
  int rc = conv_gemm_check(fn, N, H, W, Cin, Cout, k, stride, rate, pad_t, pad_l, Ho, Wo);
  if (rc) return rc;
  rc = check_mat(fn, x, ldx, Cin);
  if (rc) return rc;
  rc = check_mat(fn, y, ldy, Cout);
  if (rc) return rc;
  const int M = N * Ho * Wo, K = k * k * Cin;
  rc = check_sb(fn, wsp, pitch, K);
  if (rc) return rc;
  DL3P_CHECK_ARG(dl3p_conv2d_gemm_sb_supported(stat_partials ? 1 : 0, M, K, Cout), "%s: shape M=%d K=%d N=%d is not served by the split kernel", fn, M, K, Cout);
  DL3P_CHECK_ARG((unsigned long long)N * H * W * (unsigned long long)ldx * 4ull < (1ull << 32) &&
                     (unsigned long long)M * (unsigned long long)ldy * 4ull < (1ull << 32),
                 "%s: operands of 4 GiB or more are not supported", fn);
  GemmParams p = {};
  p.A = x; p.lda = ldx; p.scale = in_scale; p.shift = in_shift; p.act = in_act;
  p.Bsp = (const unsigned short*)wsp; p.bsp_pitch = pitch; p.bsp_plane = (long long)Cout * pitch;
  p.bias = bias; p.Y = y; p.ldy = ldy; p.partials = stat_partials;
  p.M = M; p.K = K; p.N = Cout;
  conv_gather_fwd(&p, H, W, Cin, k, stride, rate, pad_t, pad_l, Ho, Wo);
  int nt, gx, gy, mi;
  gemm_plan_sb_ga(M, Cout, stat_partials != nullptr, &nt, &gx, &gy, &p.num_m_tiles, &mi);
  if (rows_out) *rows_out = gx;
  dl3p_launch_gemm_sb(p, stat_partials != nullptr, false, true, nt, mi, 1, dim3(gx, gy), (hipStream_t)stream);
  DL3P_CHECK_LAUNCH(fn);
  return DL3P_OK;
}

extern "C" int dl3p_conv2d_gemm_bwd_data_sb(const float* dy, int lddy, const void* wdsp, int pitch, float* gx, int ldgx, int accumulate,
                                            int N, int H, int W, int Cin, int Cout, int k, int stride, int rate, int pad_t,
                                            int pad_l, int Ho, int Wo, void* stream) {
  const char* fn = "dl3p_conv2d_gemm_bwd_data_sb";
  int rc = conv_gemm_check(fn, N, H, W, Cin, Cout, k, stride, rate, pad_t, pad_l, Ho, Wo);
  if (rc) return rc;
  rc = check_mat(fn, dy, lddy, Cout);
  if (rc) return rc;
  rc = check_mat(fn, gx, ldgx, Cin);
  if (rc) return rc;
  const int M = N * H * W, K = k * k * Cout;
  rc = check_sb(fn, wdsp, pitch, K);
  if (rc) return rc;
  DL3P_CHECK_ARG(dl3p_conv2d_gemm_sb_supported(2, M, K, Cin), "%s: shape M=%d K=%d N=%d is not served by the split kernel", fn, M, K, Cin);
  DL3P_CHECK_ARG((unsigned long long)N * Ho * Wo * (unsigned long long)lddy * 4ull < (1ull << 32) &&
                     (unsigned long long)M * (unsigned long long)ldgx * 4ull < (1ull << 32),
                 "%s: operands of 4 GiB or more are not supported", fn);
  GemmParams p = {};
  p.A = dy; p.lda = lddy; p.act = DL3P_ACT_NONE;
  p.Bsp = (const unsigned short*)wdsp; p.bsp_pitch = pitch; p.bsp_plane = (long long)Cin * pitch;
  p.Y = gx; p.ldy = ldgx; p.accumulate = accumulate;
  p.M = M; p.K = K; p.N = Cin;
  p.g_RH = H; p.g_RW = W; p.g_SH = Ho; p.g_SW = Wo; p.g_C = Cout; p.g_kw = k;
  p.g_mul = 1; p.g_ay = pad_t; p.g_ax = pad_l; p.g_d = -rate; p.g_shift = stride == 2 ? 1 : 0;
  p.g_cmagic = (uint32_t)((1ull << 32) / (unsigned)Cout) + 1u;
  p.g_kwmagic = 65536 / k + 1;
  int nt, gxn, gy, mi;
  gemm_plan_sb_ga(M, Cin, false, &nt, &gxn, &gy, &p.num_m_tiles, &mi);
  dl3p_launch_gemm_sb(p, false, false, true, nt, mi, 1, dim3(gxn, gy), (hipStream_t)stream);
  DL3P_CHECK_LAUNCH(fn);
  return DL3P_OK;
}

// does this dense conv's weight gradient run on the split kernel?  -> slabs (0: no)
static int conv_wgrad_sb_route(int M, int K, int N, size_t max_slabs, int* kf, int* nw, int* kt, int* nt, int* mrows) {
  if (!dl3p_conv2d_gemm_sb_pays(4, M, K, N)) return 0;
  if (max_slabs > (size_t)DL3P_MAX_STAT_ROWS) max_slabs = DL3P_MAX_STAT_ROWS;
  // (the gathered-operand instantiations stop at 128 x 128: a pinned 128 x 256 tile means 128 x 128 here)
  return dl3p_wgrad_sb_plan(M, K, N, (int)max_slabs, g_sbw_force_tile == 4 ? 0 : g_sbw_force_tile, g_sbw_force_pc, kf, nw, kt, nt, mrows);
}

extern "C" size_t dl3p_conv2d_gemm_bwd_weight_workspace(int N, int Ho, int Wo, int Cin, int Cout, int k) {
  if (N <= 0 || Ho <= 0 || Wo <= 0 || Cin <= 0 || Cout <= 0 || k <= 0) return 0;
  int kt, nt, s, mc;
  wgrad_split(N * Ho * Wo, k * k * Cin, Cout, &kt, &nt, &s, &mc);
  size_t a = (size_t)s * k * k * Cin * Cout;
  const size_t b = (size_t)512 * Cout;   // slabs; bias column-sum partial rows
  int kf, nw, kt2, nt2, mrows;
  const size_t a2 = (size_t)conv_wgrad_sb_route(N * Ho * Wo, k * k * Cin, Cout, DL3P_MAX_STAT_ROWS, &kf, &nw, &kt2, &nt2, &mrows) * k * k * Cin * Cout;
  if (a2 > a) a = a2;
  return (a > b ? a : b) * sizeof(float);
}

static int conv2d_gemm_bwd_weight_impl(const float* x, int ldx, const float* in_scale, const float* in_shift, int in_act,
                                      const float* dy, int lddy, float* gw, float* gb, float* workspace,
                                      size_t workspace_bytes, int N, int H, int W, int Cin, int Cout, int k, int stride,
                                      int rate, int pad_t, int pad_l, int Ho, int Wo, int* rows_out, void* stream) {
  const char* fn = "dl3p_conv2d_gemm_bwd_weight";
  int rc = conv_gemm_check(fn, N, H, W, Cin, Cout, k, stride, rate, pad_t, pad_l, Ho, Wo);
  if (rc) return rc;
  rc = check_mat(fn, x, ldx, Cin);
  if (rc) return rc;
  rc = check_mat(fn, dy, lddy, Cout);
  if (rc) return rc;
  DL3P_CHECK_ARG((rows_out || (gw && aligned16(gw))) && workspace && aligned16(workspace) && !(rows_out && gb), "%s: bad arguments", fn);
  const int M = N * Ho * Wo, K = k * k * Cin;
  DL3P_CHECK_ARG((unsigned long long)N * H * W * (unsigned long long)ldx * 4ull < (1ull << 32) &&
                     (unsigned long long)M * (unsigned long long)lddy * 4ull < (1ull << 32),
                 "%s: operands of 4 GiB or more are not supported", fn);
  const size_t need = dl3p_conv2d_gemm_bwd_weight_workspace(N, Ho, Wo, Cin, Cout, k);
  if (workspace_bytes < need) {
    dl3p_set_error("%s: workspace %zu < %zu bytes", fn, workspace_bytes, need);
    return DL3P_EWORKSPACE;
  }
  WgradParams p = {};
  p.X = x; p.ldx = ldx; p.scale = in_scale; p.shift = in_shift; p.act = in_act;
  p.DY = dy; p.lddy = lddy; p.slabs = workspace; p.M = M; p.K = K; p.N = Cout;
  p.g_RH = Ho; p.g_RW = Wo; p.g_SH = H; p.g_SW = W; p.g_C = Cin; p.g_kw = k;
  p.g_mul = stride; p.g_ay = -pad_t; p.g_ax = -pad_l; p.g_d = rate;
  p.g_invRW = 1.f / (float)Wo; p.g_invRH = 1.f / (float)Ho;
  int splits;
  hipStream_t st = (hipStream_t)stream;
  int kf, nw, kt2, nt2, mrows;
  const int s2 = conv_wgrad_sb_route(M, K, Cout, workspace_bytes / ((size_t)K * Cout * 4), &kf, &nw, &kt2, &nt2, &mrows);
  if (s2 > 0) {
    const int geo[10] = {Ho, Wo, H, W, Cin, k, stride, -pad_t, -pad_l, rate};
    splits = s2;
    dl3p_launch_wgrad_sb_gx(x, ldx, in_scale, in_shift, in_act, dy, lddy, workspace, M, K, Cout, geo, kf, nw, kt2, nt2, mrows, s2, st);
  } else {
    wgrad_split(M, K, Cout, &p.ktiles, &p.ntiles, &splits, &p.mchunk);
    launch_wgrad_tiled<true>(p, splits, st);
  }
  DL3P_CHECK_LAUNCH(fn);
  if (rows_out) { *rows_out = splits; return DL3P_OK; }
  rc = dl3p_reduce_rows_impl(workspace, splits, (size_t)K * Cout, gw, 0, st);
  if (rc || !gb) return rc;
  DL3P_CHECK_ARG(aligned16(gb), "%s: gb must be 16-byte aligned", fn);
  int c4s, px, nslab;
  pick_lanes(Cout, &c4s, &px, &nslab);
  const long long need_b = ceil_div_ll(M, px);
  const int nbx = (int)(need_b < 512 ? need_b : 512);
  hipLaunchKernelGGL(colsum_kernel, dim3(nbx * nslab), dim3(256), 0, st, dy, lddy, (long long)M, Cout, c4s, px, nbx, workspace);
  DL3P_CHECK_LAUNCH("dl3p_conv2d_gemm_bwd_weight(colsum)");
  return dl3p_reduce_rows_impl(workspace, nbx, (size_t)Cout, gb, 0, st);
}

extern "C" int dl3p_conv2d_gemm_bwd_weight(const float* x, int ldx, const float* in_scale, const float* in_shift, int in_act,
                                           const float* dy, int lddy, float* gw, float* gb, float* workspace,
                                           size_t workspace_bytes, int N, int H, int W, int Cin, int Cout, int k, int stride,
                                           int rate, int pad_t, int pad_l, int Ho, int Wo, void* stream) {
  return conv2d_gemm_bwd_weight_impl(x, ldx, in_scale, in_shift, in_act, dy, lddy, gw, gb, workspace, workspace_bytes, N, H, W,
                                     Cin, Cout, k, stride, rate, pad_t, pad_l, Ho, Wo, nullptr, stream);
}

extern "C" int dl3p_conv2d_gemm_bwd_weight_slabs(const float* x, int ldx, const float* in_scale, const float* in_shift,
                                                 int in_act, const float* dy, int lddy, float* workspace,
                                                 size_t workspace_bytes, int* rows_out, int N, int H, int W, int Cin, int Cout,
                                                 int k, int stride, int rate, int pad_t, int pad_l, int Ho, int Wo,
                                                 void* stream) {
  DL3P_CHECK_ARG(rows_out != nullptr, "dl3p_conv2d_gemm_bwd_weight_slabs: rows_out is required");
  return conv2d_gemm_bwd_weight_impl(x, ldx, in_scale, in_shift, in_act, dy, lddy, nullptr, nullptr, workspace, workspace_bytes,
                                     N, H, W, Cin, Cout, k, stride, rate, pad_t, pad_l, Ho, Wo, rows_out, stream);
}

// ------------------------------------------------------------------------------ plan query (include/dl3p.h)
// the same decisions the entry points above take, reported instead of launched
extern "C" int dl3p_gemm_plan_query(int role, int M, int K, int N, int* out6) {
  DL3P_CHECK_ARG(out6 && role >= 0 && role <= 9 && M > 0 && K > 0 && N > 0, "dl3p_gemm_plan_query: bad arguments");
  for (int i = 0; i < 6; ++i) out6[i] = 0;
  SmallShape sh;
  if (role == 9) {      // the split-bf16 weight gradient: {4, tile index, 64-row blocks of k per tile, slabs, k tiles x n tiles, from table}; -1: not taken
    int kf, nw, kt, nt, mrows;
    const int s = (dl3p_pw_tiny_applies(M) || (M >= 16 && wgrad_small_pick(K, N, &sh))) ? 0 : wgrad_sb_route(M, K, N, DL3P_MAX_STAT_ROWS, &kf, &nw, &kt, &nt, &mrows);
    if (s <= 0) { out6[0] = -1; return DL3P_OK; }
    out6[0] = 4; out6[1] = nw == 16 ? 4 : (kf == 2 ? 0 : 1) + (nw == 8 ? 0 : 2); out6[2] = kf; out6[3] = s; out6[4] = kt * nt; out6[5] = gemm_tuned_lookup(9, M, K, N) != nullptr;
    return DL3P_OK;
  }
  if (role >= 5) {      // the split-bf16 twin of role - 5: {3, nt, mi, wm (0 = producer / consumer form), workgroups, from table}
    if (!dl3p_pwconv_sb_supported(role - 5, M, K, N)) { out6[0] = -1; return DL3P_OK; }
    int nt, gx, gy, mt, mi, wm;
    gemm_plan_sb(role - 5, M, K, N, &nt, &gx, &gy, &mt, &mi, &wm);
    out6[0] = 3; out6[1] = nt; out6[2] = mi; out6[3] = wm; out6[4] = gx * gy; out6[5] = gemm_tuned_lookup(role, M, K, N) != nullptr;
    return DL3P_OK;
  }
  if (role == 4) {
    if (dl3p_pw_tiny_applies(M)) { out6[0] = 2; return DL3P_OK; }
    if (M >= 16 && wgrad_small_pick(K, N, &sh)) {
      out6[0] = 1; out6[1] = sh.kt; out6[2] = sh.ntn; out6[4] = wgrad_small_grid(M, sh.kt, sh.ntn);
      return DL3P_OK;
    }
    int kw, nw, kt, nt, splits, mchunk;
    wgrad_pick_tile(M, K, N, &kw, &nw);
    wgrad_split(M, K, N, &kt, &nt, &splits, &mchunk);
    out6[1] = (kw == 2 ? 1 : 0) + (nw == 8 ? 2 : 0);
    const GemmTuned* e = gemm_tuned_lookup(4, M, K, N);
    out6[2] = g_wgrad_force_per_cu ? g_wgrad_force_per_cu : (e ? e->mi : 0);
    out6[3] = kt * nt; out6[4] = splits; out6[5] = e != nullptr;
    return DL3P_OK;
  }
  if (dl3p_pw_tiny_applies(M)) { out6[0] = 2; return DL3P_OK; }
  if (M >= pw_small_min_rows() && pw_small_pick(K, N, &sh)) {
    out6[0] = 1; out6[1] = sh.kt; out6[2] = sh.ntn; out6[3] = pw_small_grid(M);
    return DL3P_OK;
  }
  int nt, gx, gy, mt, mi;
  gemm_plan(role, M, K, N, &nt, &gx, &gy, &mt, &mi);
  out6[1] = nt; out6[2] = mi; out6[3] = gx; out6[4] = gy; out6[5] = gemm_tuned_lookup(role, M, K, N) != nullptr;
  return DL3P_OK;
}
