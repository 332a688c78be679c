// fp32-ACCURATE pointwise GEMM on the bf16 matrix pipe (gfx950): every fp32 operand is split into three bf16 pieces,
//   a = a1 + a2 + a3   (a1 = rn_bf16(a), a2 = rn_bf16(a - a1), a3 = rn_bf16(a - a1 - a2); the split is exact: 3 x 8 = 24 bits),
// and the product is accumulated in fp32 from the six cross terms of weight 2^-16 and above,
//   a b ~= a1 b1 + a1 b2 + a2 b1 + a2 b2 + a1 b3 + a3 b1        (dropped: a2 b3 + a3 b2 + a3 b3 <= 2^-23 |a b|),
// on v_mfma_f32_16x16x32_bf16.  A bf16 x bf16 product is exact in fp32 and the pipe accumulates in fp32, so the result carries
// the rounding of an fp32 GEMM (the accumulation) plus one dropped term of the size of an fp32 product rounding per product.
// On this chip the bf16 matrix pipe runs 16x the fp32-input MFMA (MI355X_MICROARCH.md, Matrix cores), so six bf16 MFMAs cost
// 6/16 of the fp32 one: 2.67x the fp32-MFMA ceiling for the compute-bound 1x1 convs (K >= 64) that are 93-98 % of the model's MACs
// (reference call sites: deeplabv3p/models/layers.py:105,157,209-218, deeplabv3p_xception.py:81).
//
// Same tiling, staging roles, prologue (the producer's BatchNorm + activation applied while the tile is staged) and epilogue
// (bias, accumulate, BatchNorm statistics / BatchNorm-backward sums) as pw_gemm_kernel in pwconv.hip; what differs:
//   * the A tile is split by the staging thread (each element once) and lives in LDS as three bf16 planes [rows][32 k], 80-byte
//     row pitch (a 16-lane ds_read_b128 group then covers all 64 banks);
//   * the B operand (the conv kernel) arrives PRE-SPLIT from global memory (dl3p_split_bf16x3 after every optimiser step, like the
//     transposed copies), [plane][Nout][Kpad] with the reduction index contiguous and zero-padded to 32;
//   * one K-step = 32 reduction indices = 6 MFMAs per 16 x 16 accumulator (the fp32 kernel: 8 MFMAs of twice the cycles);
//   * the epilogue buffer overlays the operand tiles (two workgroups per CU keep fitting).
#include "common.h"
#include "pw_gemm.h"
#include "sb_common.h"

// ------------------------------------------------------------------------------ the pre-split B operand
// rows x cols fp32 (row pitch ld) -> dst[plane][rows][pitch] bf16, columns >= cols zero.  table rows: {src offset (floats),
// rows, cols, ld, dst offset (bf16 elements, of plane 0), pitch}; `plane` = elements between the planes of ONE matrix is
// rows * pitch, so the three planes of a matrix are contiguous: [3][rows][pitch].
__global__ __launch_bounds__(256) void split_bf16x3_kernel(const float* src, unsigned short* dst, const long long* table) {
  const long long* e = table + (size_t)blockIdx.x * 6;
  const long long soff = e[0], doff = e[4];
  const int rows = (int)e[1], cols = (int)e[2], ld = (int)e[3], pitch = (int)e[5];
  const int p8 = pitch / 8;       // (pitch is a multiple of 32) eight elements per thread: two 16-byte loads, three 16-byte stores
  const long long plane = (long long)rows * pitch;
  for (long long i = (long long)blockIdx.y * 256 + threadIdx.x; i < (long long)rows * p8; i += (long long)gridDim.y * 256) {
    const int r = (int)(i / p8), c = (int)(i - (long long)r * p8) * 8;
    const long long so = soff + (long long)r * ld + c;
    float v[8];
    if (c + 8 <= cols && (so & 3) == 0) {
      const float4 a = *reinterpret_cast<const float4*>(src + so), b = *reinterpret_cast<const float4*>(src + so + 4);
      v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = c + k < cols ? src[so + k] : 0.f;
    }
    uint4 h, m, l;
    split2(v[0], v[1], h.x, m.x, l.x);
    split2(v[2], v[3], h.y, m.y, l.y);
    split2(v[4], v[5], h.z, m.z, l.z);
    split2(v[6], v[7], h.w, m.w, l.w);
    unsigned short* d = dst + doff + (long long)r * pitch + c;      // 16-byte aligned: doff and pitch are multiples of 8 elements
    *reinterpret_cast<uint4*>(d) = h;
    *reinterpret_cast<uint4*>(d + plane) = m;
    *reinterpret_cast<uint4*>(d + 2 * plane) = l;
  }
}

extern "C" int dl3p_split_bf16x3_batch(const float* src, void* dst, const int64_t* table, int n_matrices, void* stream) {
  DL3P_CHECK_ARG(src && dst && table && n_matrices > 0, "dl3p_split_bf16x3_batch: bad arguments");
  hipLaunchKernelGGL(split_bf16x3_kernel, dim3(n_matrices, 64), dim3(256), 0, (hipStream_t)stream, src,
                     reinterpret_cast<unsigned short*>(dst), reinterpret_cast<const long long*>(table));
  DL3P_CHECK_LAUNCH("dl3p_split_bf16x3_batch");
  return DL3P_OK;
}

// ------------------------------------------------------------------------------ forward / data gradient
// WM: groups of four waves along M (256 * WM threads; the B tile is staged once for all of them).  NT up to 16 (256 output
// columns per workgroup: the A tile of a 256-wide layer is then split ONCE, not once per column block).  Big tiles run one
// workgroup per CU -- what they buy is operand traffic: at 128 x 128 the split kernel re-fetches 40 KB from L2 / Infinity Cache per
// K-step and sits at that bandwidth, not at the matrix pipe.
template <int NT, bool STATS, int MI, bool BNB, bool GA, int WM>
__global__ __launch_bounds__(256 * WM, (WM == 1 && NT <= 8) ? 2 : 1) void pw_gemm_sb_kernel(GemmParams p) {
  constexpr int BKT = SB_BKT, PB = SB_PB;
  constexpr int NTHR = 256 * WM, NW = 4 * WM, RPP = 64 * WM;     // threads, waves, A rows staged per pass
  constexpr int BM = 64 * MI * WM, BN = 16 * NT;
  constexpr int A_PLANE = BM * PB, B_PLANE = BN * PB;          // bf16 elements
  constexpr int OPER_BYTES = 3 * (A_PLANE + B_PLANE) * 2;
  constexpr int NA = MI;                                       // A passes of RPP rows: a thread stages 8 consecutive k of one row
  constexpr int NBC = (3 * BN * 4 + NTHR - 1) / NTHR;          // 16-byte chunks of the B tile per thread
  constexpr int TPP = NT < 4 ? NT : 4;
  constexpr int NPASS = (NT + TPP - 1) / TPP;
  constexpr int CH = 16 * TPP;
  constexpr int EPITCH = CH + 4;
  constexpr int RW = 16 * MI;
  constexpr int ES_BYTES = NW * RW * EPITCH * 4;
  constexpr int RED_OFF = (OPER_BYTES > ES_BYTES ? OPER_BYTES : ES_BYTES);
  extern __shared__ __attribute__((aligned(16))) unsigned char sb_lds[];
  unsigned short* As = reinterpret_cast<unsigned short*>(sb_lds);
  unsigned short* Bs = As + 3 * A_PLANE;
  float* Es = reinterpret_cast<float*>(sb_lds);                // overlays the operand tiles (barrier-separated)
  float* red = reinterpret_cast<float*>(sb_lds + RED_OFF);

  const int t = threadIdx.x;
  const int l = t & 63;
  const int w = t >> 6;
  const int l15 = l & 15;
  const int q = l >> 4;
  const int n0 = blockIdx.y * BN;
  const int nk = (p.K + BKT - 1) / BKT;
  const int my_tiles = (p.num_m_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  const int it_total = my_tiles * nk;

  const int ar = t >> 2;           // A row within a pass of 64 rows
  const int ak8 = (t & 3) * 8;     // first of this thread's 8 k of the K tile
  const char* Ab = reinterpret_cast<const char*>(p.A);
  // the producer's BN scale / shift ride along with the A loads; without one the loads still happen (from A: any valid address)
  // and stage() ignores them -- see the note on unconditional prefetches in step()
  const float* scp = p.scale ? p.scale : p.A;
  const float* shp = p.scale ? p.shift : p.A;

  // A is prefetched TWO K-steps ahead (two register sets, PAR = it & 1): with six bf16 MFMAs per tile the multiply phase of a
  // K-step lasts ~0.7 us, less than an HBM round trip under load (measured neutral to -3 %: the loop is bound by its staging
  // pipeline).  B (L2 hits) stays one step ahead and is issued BEFORE the A loads of the step after next, so that the
  // wait for it does not drain those.
  float4 ra[2][NA][2];
  uint4 rb[NBC];
  float4 rsc[2][2], rsh[2][2];
  uint32_t a_row[2][NA];
  int g_by[2][GA ? NA : 1], g_bx[2][GA ? NA : 1];
  uint32_t g_ok[2] = {0, 0};       // GA: bit (2 i + h) = float4 h of pass i hit the source tensor
  uint32_t b_off[NBC];             // byte offset of this thread's B chunks at k0 = 0
  bool b_ok[NBC];
  int b_lds[NBC];                  // bf16 element offset of the chunk in Bs
  int pf_m0[2] = {-1, -1};
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
    for (int h = 0; h < 2; ++h) { rsc[s2][h] = make_float4(1.f, 1.f, 1.f, 1.f); rsh[s2][h] = zero4(); }
#pragma unroll
  for (int i = 0; i < NBC; ++i) {
    const int idx = t + NTHR * i;
    const int c = idx < 3 * BN * 4 ? idx : 3 * BN * 4 - 1;
    const int plane = c / (BN * 4), rem = c - plane * (BN * 4);
    const int r = rem >> 2, ch = rem & 3;
    const int n = n0 + r;
    b_ok[i] = idx < 3 * BN * 4 && n < p.N;
    b_off[i] = (uint32_t)(((long long)plane * p.bsp_plane + (long long)min(n, p.N - 1) * p.bsp_pitch + ch * 8) * 2);
    b_lds[i] = plane * B_PLANE + r * PB + ch * 8;
  }
  const char* Bb = reinterpret_cast<const char*>(p.Bsp);

  auto prefetch_b = [&](int it) __attribute__((always_inline)) {
    const int k0 = (it % nk) * BKT;
#pragma unroll
    for (int i = 0; i < NBC; ++i) rb[i] = *reinterpret_cast<const uint4*>(Bb + (b_off[i] + (uint32_t)k0 * 2u));
  };

  auto prefetch_a = [&](int it, auto par) __attribute__((always_inline)) {
    constexpr int P = decltype(par)::value;
    const int kt = it % nk;
    const int mt = blockIdx.x + (it / nk) * gridDim.x;
    const int m0 = mt * BM;
    const int k0 = kt * BKT;
    if (GA) {
      if (m0 != pf_m0[P]) {
        pf_m0[P] = m0;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
          const int m = m0 + ar + RPP * i;
          const int mc = min(m, p.M - 1);
          const int row = mc / p.g_RW, x = mc - row * p.g_RW;
          const int n = row / p.g_RH, y = row - n * p.g_RH;
          g_by[P][i] = m < p.M ? y * p.g_mul + p.g_ay : -(1 << 20);
          g_bx[P][i] = x * p.g_mul + p.g_ax;
          a_row[P][i] = (uint32_t)n * (uint32_t)(p.g_SH * p.g_SW);
        }
      }
      g_ok[P] = 0;
      const int par_mask = (1 << p.g_shift) - 1;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int kraw = k0 + ak8 + 4 * h;
        const int k = min(kraw, p.K - 4);
        const int tap = (int)__umulhi((uint32_t)k, p.g_cmagic);
        const int c = k - tap * p.g_C;
        const int ky = (tap * p.g_kwmagic) >> 16, kx = tap - ky * p.g_kw;
        const int dyo = ky * p.g_d, dxo = kx * p.g_d;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
          const int ty = g_by[P][i] + dyo, tx = g_bx[P][i] + dxo;
          const int sy = ty >> p.g_shift, sx = tx >> p.g_shift;
          const bool ok = ty >= 0 && tx >= 0 && ((ty | tx) & par_mask) == 0 && sy < p.g_SH && sx < p.g_SW && kraw < p.K;
          const uint32_t off = ok ? ((a_row[P][i] + (uint32_t)(sy * p.g_SW + sx)) * (uint32_t)p.lda + (uint32_t)c) * 4u : 0u;
          ra[P][i][h] = *reinterpret_cast<const float4*>(Ab + off);
          g_ok[P] |= ok ? (1u << (2 * i + h)) : 0u;
        }
        rsc[P][h] = *reinterpret_cast<const float4*>(scp + c);
        rsh[P][h] = *reinterpret_cast<const float4*>(shp + c);
      }
    } else {
      if (m0 != pf_m0[P]) {
        pf_m0[P] = m0;
#pragma unroll
        for (int i = 0; i < NA; ++i) a_row[P][i] = (uint32_t)min(m0 + ar + RPP * i, p.M - 1) * (uint32_t)p.lda * 4u;
      }
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const uint32_t kb = (uint32_t)min(k0 + ak8 + 4 * h, p.K - 4) * 4u;
#pragma unroll
        for (int i = 0; i < NA; ++i) ra[P][i][h] = *reinterpret_cast<const float4*>(Ab + (a_row[P][i] + kb));
        rsc[P][h] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(scp) + kb);
        rsh[P][h] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(shp) + kb);
      }
    }
  };

  const float act_lo = p.act == DL3P_ACT_NONE ? -DL3P_INF : 0.f;
  const float act_hi = (p.act == DL3P_ACT_NONE || p.act == DL3P_ACT_RELU) ? DL3P_INF : 6.f;
  auto prologue4 = [&](float4 v, float4 sc4, float4 sh4) __attribute__((always_inline)) {
    if (p.scale) v = fma4(v, sc4, sh4);
    if (p.act >= DL3P_ACT_HSWISH) return act_apply4(v, p.act);
    return make_float4(__builtin_amdgcn_fmed3f(v.x, act_lo, act_hi), __builtin_amdgcn_fmed3f(v.y, act_lo, act_hi),
                       __builtin_amdgcn_fmed3f(v.z, act_lo, act_hi), __builtin_amdgcn_fmed3f(v.w, act_lo, act_hi));
  };
  const bool has_pro = p.scale != nullptr || p.act != DL3P_ACT_NONE;

  auto stage = [&](int it, auto par) __attribute__((always_inline)) {
    constexpr int P = decltype(par)::value;
    const int kt = it % nk;
    const int mt = blockIdx.x + (it / nk) * gridDim.x;
    const int m0 = mt * BM;
    const int k0 = kt * BKT;
    const bool a_edge = m0 + BM > p.M || k0 + BKT > p.K;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int r = ar + RPP * i;
      float4 v[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        v[h] = ra[P][i][h];
        if (has_pro) v[h] = prologue4(v[h], rsc[P][h], rsh[P][h]);
        // padding of the M / K tails and (GA) taps outside the source stay exactly zero
        if (GA) v[h] = ((g_ok[P] >> (2 * i + h)) & 1u) ? v[h] : zero4();
        else if (a_edge) v[h] = (k0 + ak8 + 4 * h < p.K && m0 + r < p.M) ? v[h] : zero4();
      }
      uint4 hh, mm, ll;
#ifdef DL3P_SB_ABLATE
      if (p.stagger == 1) {          // ablation: no split arithmetic (results wrong, time only)
        hh = make_uint4(__builtin_bit_cast(uint32_t, v[0].x), __builtin_bit_cast(uint32_t, v[0].z), __builtin_bit_cast(uint32_t, v[1].x), __builtin_bit_cast(uint32_t, v[1].z));
        mm = make_uint4(__builtin_bit_cast(uint32_t, v[0].y), __builtin_bit_cast(uint32_t, v[0].w), __builtin_bit_cast(uint32_t, v[1].y), __builtin_bit_cast(uint32_t, v[1].w));
        ll = hh;
      } else
#endif
      {
      split2(v[0].x, v[0].y, hh.x, mm.x, ll.x);
      split2(v[0].z, v[0].w, hh.y, mm.y, ll.y);
      split2(v[1].x, v[1].y, hh.z, mm.z, ll.z);
      split2(v[1].z, v[1].w, hh.w, mm.w, ll.w);
      }
#ifdef DL3P_SB_ABLATE
      if (p.stagger == 6 && it > 0) continue;      // ablation: no A-tile LDS stores after the first step
#endif
      unsigned short* d = As + r * PB + ak8;
      *reinterpret_cast<uint4*>(d) = hh;
      *reinterpret_cast<uint4*>(d + A_PLANE) = mm;
      *reinterpret_cast<uint4*>(d + 2 * A_PLANE) = ll;
    }
#pragma unroll
    for (int i = 0; i < NBC; ++i) {
      // (the empty asm pins the wait for rb[i] in front of its store, as in pw_gemm_kernel: left alone, the compiler parks an
      // s_waitcnt inside the next K-step's prefetch burst)
      asm volatile("" :: "v"(rb[i].x), "v"(rb[i].y), "v"(rb[i].z), "v"(rb[i].w));
#ifdef DL3P_SB_ABLATE
      if (p.stagger == 5 && it > 0) continue;      // ablation: no B-tile LDS stores after the first step
#endif
      if (t + NTHR * i < 3 * BN * 4) {
        uint4 v = rb[i];
        if (!b_ok[i]) v = make_uint4(0u, 0u, 0u, 0u);
        *reinterpret_cast<uint4*>(Bs + b_lds[i]) = v;
      }
    }
  };

  f32x4 acc[MI][NT];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float4 st_s[STATS ? NPASS : 1], st_q[STATS ? NPASS : 1];
  if (STATS) {
#pragma unroll
    for (int i = 0; i < NPASS; ++i) { st_s[i] = zero4(); st_q[i] = zero4(); }
  }

  auto step = [&](int it, auto par) __attribute__((always_inline)) {
    constexpr int P = decltype(par)::value;
    stage(it, par);
    lds_barrier();
#ifdef DL3P_SB_ABLATE
    const int abl = p.stagger;
#else
    constexpr int abl = 0;
#endif
    // UNCONDITIONAL (the index clamps at the last step, whose loads are then simply repeated): a prefetch under `if` leaves the
    // compiler's s_waitcnt pass unable to count the younger loads in the queue, and every wait of the next stage() degrades to
    // vmcnt(0) -- the two-steps-ahead A prefetch then waits for the loads issued one step ago as well
    if (!(abl == 7 && it > 1)) prefetch_b(min(it + 1, it_total - 1));
    if (!(abl == 3 && it > 1)) prefetch_a(min(it + 2, it_total - 1), par);       // into the register set this step has just consumed
    if (abl != 2) {
      s16x8 xa[MI][3];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
          xa[mi][pl] = *reinterpret_cast<const s16x8*>(As + pl * A_PLANE + (w * 16 * MI + mi * 16 + l15) * PB + q * 8);
#pragma unroll
      for (int ni = 0; ni < NT; ++ni) {
        s16x8 wb[3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) wb[pl] = *reinterpret_cast<const s16x8*>(Bs + pl * B_PLANE + (ni * 16 + l15) * PB + q * 8);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          // smallest terms first; MFMA "A" = weights (row = output channel), "B" = activations (column = pixel): a lane ends with
          // 4 consecutive output channels of one pixel
          acc[mi][ni] = split_mac(acc[mi][ni], wb[0], wb[1], wb[2], xa[mi][0], xa[mi][1], xa[mi][2]);
        }
      }
    }
    lds_barrier();
    if (it % nk == nk - 1 && abl != 4) {
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
        const int row_lim = p.M - (m0 + w * RW);
        if (col_ok) {
          float4 bias4 = zero4();
          if (p.bias) bias4 = ld4(p.bias + n);
          char* yb = reinterpret_cast<char*>(p.Y) + ((uint32_t)(m0 + w * RW + rr) * (uint32_t)p.ldy + (uint32_t)n) * 4u;
          const uint32_t ystep = (uint32_t)p.ldy * 16u;
          constexpr bool bnb = STATS && BNB;
          float4 bsc = zero4(), bsh = zero4(), bmu = zero4(), bis = zero4();
          float4 zpre[RW / 4];
          if (bnb) {
            bsc = ld4(p.bb_scale + n); bsh = ld4(p.bb_shift + n); bmu = ld4(p.bb_mean + n); bis = ld4(p.bb_invstd + n);
            const char* zbase = reinterpret_cast<const char*>(p.bb_z) + (uint32_t)n * 4u;
#pragma unroll
            for (int i = 0; i < RW / 4; ++i) {
              const int mrow = min(m0 + w * RW + 4 * i + rr, p.M - 1);
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
      lds_barrier();   // the next stage() overwrites the epilogue buffer
    }
  };
  if (it_total > 0) {
    prefetch_b(0);
    prefetch_a(0, std::integral_constant<int, 0>{});
    prefetch_a(min(1, it_total - 1), std::integral_constant<int, 1>{});
  }
  // pairs of steps in a branch-free body (see the note at the prefetches), the odd last one after the loop
  int it = 0;
  for (; it + 1 < it_total; it += 2) {
    step(it, std::integral_constant<int, 0>{});
    step(it + 1, std::integral_constant<int, 1>{});
  }
  if (it < it_total) step(it, std::integral_constant<int, 0>{});

  if (STATS) {
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
          red[(0 * NW + w) * BN + col] = s1;
          red[(1 * NW + w) * BN + col] = s2;
        }
      }
    }
    lds_barrier();
    if (p.partials) {
      for (int i = t; i < 2 * BN; i += NTHR) {
        const int which = i / BN, nn = i - which * BN;
        if (n0 + nn < p.N) {
          float s = 0.f;
#pragma unroll
          for (int ww = 0; ww < NW; ++ww) s += red[(which * NW + ww) * BN + nn];
          p.partials[((size_t)blockIdx.x * 2 + which) * p.N + n0 + nn] = s;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------ producer / consumer form
// The same product with the workgroup's eight waves SPECIALISED (512 threads, one workgroup per CU): waves 0-3 only multiply (one
// per SIMD: fragments from LDS, six MFMAs per tile, epilogue), waves 4-7 only stage (global loads two K-steps ahead, the producer's
// BatchNorm + activation, the 3-way split, LDS stores) into the OTHER of two operand buffers -- one barrier per K-step hands a
// buffer over.  On every SIMD a multiplying wave and a staging wave sit side by side: the MFMA holds the vector issue for 8 of
// its 16 cycles, the staging wave's VALU / LDS / memory instructions take the rest, so the staging pipeline that bounded the
// symmetric kernel (283 of 425 us with the MFMAs removed) runs UNDER the multiply phase instead of in front of it,
// and the epilogue of a row tile overlaps the staging of the next one's first K-steps.
template <int NT, bool STATS, int MI, bool BNB>
__global__ __launch_bounds__(512, 1) void pw_gemm_sbp_kernel(GemmParams p) {
  constexpr int BKT = SB_BKT, PB = SBP_PB;
  constexpr int BM = 64 * MI, BN = 16 * NT;
  constexpr int A_PLANE = BM * PB, B_PLANE = BN * PB;          // bf16 elements
  constexpr int STAGE = 3 * (A_PLANE + B_PLANE);               // one operand buffer, bf16 elements
  constexpr int NA = MI;
  constexpr int NBC = (3 * BN * 4 + 255) / 256;
  constexpr int TPP = NT < 4 ? NT : 4;
  constexpr int NPASS = (NT + TPP - 1) / TPP;
  constexpr int CH = 16 * TPP;
  constexpr int EPITCH = CH + 4;
  constexpr int RW = 16 * MI;
  extern __shared__ __attribute__((aligned(16))) unsigned char sbp_lds[];
  unsigned short* Stage0 = reinterpret_cast<unsigned short*>(sbp_lds);
  float* Es = reinterpret_cast<float*>(sbp_lds + 2 * STAGE * 2);       // consumer-wave-private epilogue slices
  float* red = Es;                                                      // the final statistics reduction reuses them

  const int t = threadIdx.x;
  const int l = t & 63;
  const int wv = t >> 6;
  const bool producer = wv >= 4;
  const int w = wv & 3;                 // consumer wave / producer wave index
  const int l15 = l & 15;
  const int q = l >> 4;
  const int n0 = blockIdx.y * BN;
  const int nk = (p.K + BKT - 1) / BKT;
  const int my_tiles = (p.num_m_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  const int it_total = my_tiles * nk;

  if (producer) {
    const int pt = t - 256;            // 0 .. 255
    const int ar = pt >> 2;
    const int ak8 = (pt & 3) * 8;
    const char* Ab = reinterpret_cast<const char*>(p.A);
    const char* Bb = reinterpret_cast<const char*>(p.Bsp);
    const float* scp = p.scale ? p.scale : p.A;      // loaded either way (see pw_gemm_sb_kernel), used only with a producer BN
    const float* shp = p.scale ? p.shift : p.A;
    // DEPTH K-steps of operands in flight per staging wave.  Little's law: the multiply phase wants 40 KB per CU every ~0.75 us; at the
    // ~2.5 us a load takes with every CU pulling, that is ~130 KB in flight per CU -- two steps (56 KB) delivered 4.8 TB/s chip-wide
    // and set the pace of the whole kernel (its time with the MFMAs removed, 331 us, is that rate).  Four register sets of
    // (A: 2 x NA float4, B: NBC x 16 bytes) per thread = 160 KB per CU.
    // (three where four would spill: the 128 x 128 tile).
    constexpr int DEPTH = (NT >= 6 && MI >= 2) ? 3 : 4;
    float4 ra[DEPTH][NA][2];
    uint4 rb[DEPTH][NBC];
    float4 rsc[DEPTH][2], rsh[DEPTH][2];
    uint32_t b_off[NBC];
    bool b_ok[NBC];
    int b_lds[NBC];
#pragma unroll
    for (int s2 = 0; s2 < DEPTH; ++s2)
#pragma unroll
      for (int h = 0; h < 2; ++h) { rsc[s2][h] = make_float4(1.f, 1.f, 1.f, 1.f); rsh[s2][h] = zero4(); }
#pragma unroll
    for (int i = 0; i < NBC; ++i) {
      const int idx = pt + 256 * i;
      const int c = idx < 3 * BN * 4 ? idx : 3 * BN * 4 - 1;
      const int plane = c / (BN * 4), rem = c - plane * (BN * 4);
      const int r = rem >> 2, ch = rem & 3;
      const int n = n0 + r;
      b_ok[i] = idx < 3 * BN * 4 && n < p.N;
      b_off[i] = (uint32_t)(((long long)plane * p.bsp_plane + (long long)min(n, p.N - 1) * p.bsp_pitch + ch * 8) * 2);
      b_lds[i] = 3 * A_PLANE + plane * B_PLANE + r * PB + ch * 8;
    }
#ifdef DL3P_SB_ABLATE
    const int abl = p.stagger;
#else
    constexpr int abl = 0;
#endif
    auto prefetch = [&](int it, auto par) __attribute__((always_inline)) {
      constexpr int P = decltype(par)::value;
      const int kt = it % nk;
      const int m0 = (blockIdx.x + (it / nk) * gridDim.x) * BM;
      const int k0 = kt * BKT;
      if (!(abl == 7 && it >= DEPTH)) {
#pragma unroll
        for (int i = 0; i < NBC; ++i) rb[P][i] = *reinterpret_cast<const uint4*>(Bb + (b_off[i] + (uint32_t)k0 * 2u));
      }
      if (abl == 3 && it >= DEPTH) return;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const uint32_t kb = (uint32_t)min(k0 + ak8 + 4 * h, p.K - 4) * 4u;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
          const uint32_t rowb = (uint32_t)min(m0 + ar + 64 * i, p.M - 1) * (uint32_t)p.lda * 4u;
          ra[P][i][h] = *reinterpret_cast<const float4*>(Ab + (rowb + kb));
        }
        rsc[P][h] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(scp) + kb);
        rsh[P][h] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(shp) + kb);
      }
    };
    const float act_lo = p.act == DL3P_ACT_NONE ? -DL3P_INF : 0.f;
    const float act_hi = (p.act == DL3P_ACT_NONE || p.act == DL3P_ACT_RELU) ? DL3P_INF : 6.f;
    const bool has_pro = p.scale != nullptr || p.act != DL3P_ACT_NONE;
    auto stage = [&](int it, auto par, int slot = 0) __attribute__((always_inline)) {
      constexpr int P = decltype(par)::value;
      unsigned short* buf = Stage0 + (slot & 1) * STAGE;
      const int kt = it % nk;
      const int m0 = (blockIdx.x + (it / nk) * gridDim.x) * BM;
      const int k0 = kt * BKT;
      const bool a_edge = m0 + BM > p.M || k0 + BKT > p.K;
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const int r = ar + 64 * i;
        float4 v[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          v[h] = ra[P][i][h];
          if (has_pro) {
            if (p.scale) v[h] = fma4(v[h], rsc[P][h], rsh[P][h]);
            if (p.act >= DL3P_ACT_HSWISH) v[h] = act_apply4(v[h], p.act);
            else v[h] = make_float4(__builtin_amdgcn_fmed3f(v[h].x, act_lo, act_hi), __builtin_amdgcn_fmed3f(v[h].y, act_lo, act_hi),
                                    __builtin_amdgcn_fmed3f(v[h].z, act_lo, act_hi), __builtin_amdgcn_fmed3f(v[h].w, act_lo, act_hi));
          }
          if (a_edge) v[h] = (k0 + ak8 + 4 * h < p.K && m0 + r < p.M) ? v[h] : zero4();
        }
        uint4 hh, mm, ll;
#ifdef DL3P_SB_ABLATE
        if (abl == 1) {
          hh = make_uint4(__builtin_bit_cast(uint32_t, v[0].x), __builtin_bit_cast(uint32_t, v[0].z), __builtin_bit_cast(uint32_t, v[1].x), __builtin_bit_cast(uint32_t, v[1].z));
          mm = make_uint4(__builtin_bit_cast(uint32_t, v[0].y), __builtin_bit_cast(uint32_t, v[0].w), __builtin_bit_cast(uint32_t, v[1].y), __builtin_bit_cast(uint32_t, v[1].w));
          ll = hh;
        } else
#endif
        {
        split2(v[0].x, v[0].y, hh.x, mm.x, ll.x);
        split2(v[0].z, v[0].w, hh.y, mm.y, ll.y);
        split2(v[1].x, v[1].y, hh.z, mm.z, ll.z);
        split2(v[1].z, v[1].w, hh.w, mm.w, ll.w);
        }
        if (abl == 6 && it > 1) continue;
        unsigned short* d = buf + r * PB + ak8;
        *reinterpret_cast<uint4*>(d) = hh;
        *reinterpret_cast<uint4*>(d + A_PLANE) = mm;
        *reinterpret_cast<uint4*>(d + 2 * A_PLANE) = ll;
      }
#pragma unroll
      for (int i = 0; i < NBC; ++i) {
        asm volatile("" :: "v"(rb[P][i].x), "v"(rb[P][i].y), "v"(rb[P][i].z), "v"(rb[P][i].w));
        if (abl == 5 && it > 1) continue;
        if (pt + 256 * i < 3 * BN * 4) {
          uint4 v = rb[P][i];
          if (!b_ok[i]) v = make_uint4(0u, 0u, 0u, 0u);
          *reinterpret_cast<uint4*>(buf + b_lds[i]) = v;
        }
      }
    };
    using P0 = std::integral_constant<int, 0>;
    // every load below is UNCONDITIONAL and the main loop body branch-free (indices clamp at the last K-step; what is staged past
    // it goes to the buffer nobody reads any more): under `if`, the compiler's s_waitcnt pass cannot count the younger loads in
    // the queue and each stage() waits with vmcnt(0) -- for the requests just issued, not only for its own
    const int last = it_total - 1;
    if (it_total > 0) {
      static_for<DEPTH>([&](auto j) { prefetch(min((int)decltype(j)::value, last), j); });
      stage(0, P0{});
      prefetch(min(DEPTH, last), P0{});
    }
    lds_barrier();
    // iteration i (the multiplying waves work on buffer i & 1): stage K-step i + 1 (register set (i + 1) % DEPTH) into the other
    // buffer and request step i + 1 + DEPTH into the set just consumed; one barrier per iteration, like the multiplying waves
    auto body = [&](int i, auto par) __attribute__((always_inline)) {
      stage(min(i + 1, last), par, i + 1);
      prefetch(min(i + 1 + DEPTH, last), par);
      lds_barrier();
    };
    int i = 0;
    for (; i + DEPTH <= it_total; i += DEPTH)
      static_for<DEPTH>([&](auto j) {
        constexpr int J = decltype(j)::value;
        body(i + J, std::integral_constant<int, (J + 1) % DEPTH>{});
      });
    static_for<DEPTH - 1>([&](auto j) {
      constexpr int J = decltype(j)::value;
      if (i + J < it_total) body(i + J, std::integral_constant<int, (J + 1) % DEPTH>{});
    });
    if (STATS) lds_barrier();       // the statistics reduction's barrier (consumers write `red` in front of it)
    return;
  }

  // ---------------------------------------------------------------- multiplying waves
  f32x4 acc[MI][NT];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float4 st_s[STATS ? NPASS : 1], st_q[STATS ? NPASS : 1];
  if (STATS) {
#pragma unroll
    for (int i = 0; i < NPASS; ++i) { st_s[i] = zero4(); st_q[i] = zero4(); }
  }
  float* es = Es + w * RW * EPITCH;
  lds_barrier();
  for (int it = 0; it < it_total; ++it) {
    const unsigned short* As = Stage0 + (it & 1) * STAGE;
    const unsigned short* Bs = As + 3 * A_PLANE;
#ifdef DL3P_SB_ABLATE
    const int abl = p.stagger;
#else
    constexpr int abl = 0;
#endif
    if (abl != 2) {
      s16x8 xa[MI][3];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
          xa[mi][pl] = *reinterpret_cast<const s16x8*>(As + pl * A_PLANE + (w * 16 * MI + mi * 16 + l15) * PB + q * 8);
#pragma unroll
      for (int ni = 0; ni < NT; ++ni) {
        s16x8 wb[3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) wb[pl] = *reinterpret_cast<const s16x8*>(Bs + pl * B_PLANE + (ni * 16 + l15) * PB + q * 8);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          acc[mi][ni] = split_mac(acc[mi][ni], wb[0], wb[1], wb[2], xa[mi][0], xa[mi][1], xa[mi][2]);
        }
      }
    }
    if (it % nk == nk - 1 && abl != 4) {
      // epilogue of this row tile, through the wave's private LDS slice (no barrier: LDS executes one wave's accesses in order)
      const int m0 = (blockIdx.x + (it / nk) * gridDim.x) * BM;
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
        const int row_lim = p.M - (m0 + w * RW);
        if (col_ok) {
          float4 bias4 = zero4();
          if (p.bias) bias4 = ld4(p.bias + n);
          char* yb = reinterpret_cast<char*>(p.Y) + ((uint32_t)(m0 + w * RW + rr) * (uint32_t)p.ldy + (uint32_t)n) * 4u;
          const uint32_t ystep = (uint32_t)p.ldy * 16u;
          constexpr bool bnb = STATS && BNB;
          float4 bsc = zero4(), bsh = zero4(), bmu = zero4(), bis = zero4();
          float4 zpre[RW / 4];
          if (bnb) {
            bsc = ld4(p.bb_scale + n); bsh = ld4(p.bb_shift + n); bmu = ld4(p.bb_mean + n); bis = ld4(p.bb_invstd + n);
            const char* zbase = reinterpret_cast<const char*>(p.bb_z) + (uint32_t)n * 4u;
#pragma unroll
            for (int i = 0; i < RW / 4; ++i) {
              const int mrow = min(m0 + w * RW + 4 * i + rr, p.M - 1);
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
    }
    lds_barrier();
  }
  if (STATS) {
    const int rr = l >> 4, cq = l & 15;
    // (every multiplying wave is past its last epilogue read of `es` here only for ITSELF: `red` overlays all four slices, so the
    // four waves meet first)
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      float sv[4] = {st_s[ps].x, st_s[ps].y, st_s[ps].z, st_s[ps].w};
      float qv[4] = {st_q[ps].x, st_q[ps].y, st_q[ps].z, st_q[ps].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float s1 = sv[e], s2 = qv[e];
        s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
        s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
        sv[e] = s1; qv[e] = s2;
      }
      st_s[ps] = make_float4(sv[0], sv[1], sv[2], sv[3]);
      st_q[ps] = make_float4(qv[0], qv[1], qv[2], qv[3]);
    }
    // the last loop barrier above separates every wave's final epilogue from these writes
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      const float sv[4] = {st_s[ps].x, st_s[ps].y, st_s[ps].z, st_s[ps].w};
      const float qv[4] = {st_q[ps].x, st_q[ps].y, st_q[ps].z, st_q[ps].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int col = ps * CH + cq * 4 + e;
        if (rr == 0 && cq * 4 < CH && col < BN) {
          red[(0 * 4 + w) * BN + col] = sv[e];
          red[(1 * 4 + w) * BN + col] = qv[e];
        }
      }
    }
    lds_barrier();
    if (p.partials) {
      for (int i = t; i < 2 * BN; i += 256) {
        const int which = i / BN, nn = i - which * BN;
        if (n0 + nn < p.N) {
          const float s = red[(which * 4 + 0) * BN + nn] + red[(which * 4 + 1) * BN + nn] +
                          red[(which * 4 + 2) * BN + nn] + red[(which * 4 + 3) * BN + nn];
          p.partials[((size_t)blockIdx.x * 2 + which) * p.N + n0 + nn] = s;
        }
      }
    }
  }
}

template <int NT, bool STATS, int MI, bool BNB>
static void launch_sbp_one(const GemmParams& p, dim3 grid, hipStream_t st) {
  constexpr int BM = 64 * MI, BN = 16 * NT;
  constexpr int STAGE_B = 3 * (BM + BN) * SBP_PB * 2;
  constexpr int TPP = NT < 4 ? NT : 4;
  constexpr int ES = 4 * 16 * MI * (16 * TPP + 4) * 4;
  constexpr int RED = STATS ? 2 * 4 * BN * 4 : 0;
  constexpr size_t lds = (size_t)2 * STAGE_B + (ES > RED ? ES : RED);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)pw_gemm_sbp_kernel<NT, STATS, MI, BNB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  dl3p_launch(pw_gemm_sbp_kernel<NT, STATS, MI, BNB>, grid, dim3(512), lds, st, p);
}

template <bool STATS, bool BNB>
static void launch_sbp_nt(const GemmParams& p, int nt, int mi, dim3 grid, hipStream_t st) {
#define DL3P_SBP(N_) case N_: if (mi == 1) launch_sbp_one<N_, STATS, 1, BNB>(p, grid, st); else launch_sbp_one<N_, STATS, 2, BNB>(p, grid, st); break;
  switch (nt) {
    DL3P_SBP(1) DL3P_SBP(2) DL3P_SBP(3) DL3P_SBP(4) DL3P_SBP(5) DL3P_SBP(6) DL3P_SBP(7)
    default: if (mi == 1) launch_sbp_one<8, STATS, 1, BNB>(p, grid, st); else launch_sbp_one<8, STATS, 2, BNB>(p, grid, st); break;
  }
#undef DL3P_SBP
}

// producer / consumer form: (nt <= 8, mi) tiles, one 512-thread workgroup per CU
void dl3p_launch_gemm_sbp(const GemmParams& p, bool stats, bool bnb, int nt, int mi, dim3 grid, hipStream_t st) {
  if (bnb) launch_sbp_nt<true, true>(p, nt, mi, grid, st);
  else if (stats) launch_sbp_nt<true, false>(p, nt, mi, grid, st);
  else launch_sbp_nt<false, false>(p, nt, mi, grid, st);
}

template <int NT, bool STATS, int MI, bool BNB, bool GA, int WM>
static void launch_sb_one(const GemmParams& p, dim3 grid, hipStream_t st) {
  constexpr int BM = 64 * MI * WM, BN = 16 * NT;
  constexpr int OPER = 3 * (BM + BN) * SB_PB * 2;
  constexpr int TPP = NT < 4 ? NT : 4;
  constexpr int ES = 4 * WM * 16 * MI * (16 * TPP + 4) * 4;
  constexpr int RED = STATS ? 2 * 4 * WM * BN * 4 : 0;
  constexpr size_t lds = (size_t)(OPER > ES ? OPER : ES) + RED;
  static bool attr_set = false;
  if (!attr_set) {
    if (lds > 64 * 1024)
      (void)hipFuncSetAttribute((const void*)pw_gemm_sb_kernel<NT, STATS, MI, BNB, GA, WM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  dl3p_launch(pw_gemm_sb_kernel<NT, STATS, MI, BNB, GA, WM>, grid, dim3(256 * WM), lds, st, p);
}

template <bool STATS, int MI, bool BNB, bool GA>
static void launch_sb_mi(const GemmParams& p, int nt, dim3 grid, hipStream_t st) {
  switch (nt) {
    case 1: launch_sb_one<1, STATS, MI, BNB, GA, 1>(p, grid, st); break;
    case 2: launch_sb_one<2, STATS, MI, BNB, GA, 1>(p, grid, st); break;
    case 3: launch_sb_one<3, STATS, MI, BNB, GA, 1>(p, grid, st); break;
    case 4: launch_sb_one<4, STATS, MI, BNB, GA, 1>(p, grid, st); break;
    case 5: launch_sb_one<5, STATS, MI, BNB, GA, 1>(p, grid, st); break;
    case 6: launch_sb_one<6, STATS, MI, BNB, GA, 1>(p, grid, st); break;
    case 7: launch_sb_one<7, STATS, MI, BNB, GA, 1>(p, grid, st); break;
    default: launch_sb_one<8, STATS, MI, BNB, GA, 1>(p, grid, st); break;
  }
}

// the wide-tile family (one workgroup per CU): nt in {8, 12, 16}, mi in {1, 2}, wm in {1, 2}; never with the implicit-GEMM gather
template <bool STATS, bool BNB>
static void launch_sb_wide(const GemmParams& p, int nt, int mi, int wm, dim3 grid, hipStream_t st) {
#define DL3P_SBW(N_, M_, W_) if (nt == N_ && mi == M_ && wm == W_) { launch_sb_one<N_, STATS, M_, BNB, false, W_>(p, grid, st); return; }
  DL3P_SBW(16, 2, 1) DL3P_SBW(16, 1, 2) DL3P_SBW(16, 2, 2) DL3P_SBW(12, 2, 1) DL3P_SBW(12, 2, 2) DL3P_SBW(8, 2, 2) DL3P_SBW(16, 1, 1)
#undef DL3P_SBW
  launch_sb_one<16, STATS, 2, BNB, false, 1>(p, grid, st);
}

bool dl3p_sb_wide_config(int nt, int mi, int wm) {
  static const int ok[][3] = {{16, 2, 1}, {16, 1, 2}, {16, 2, 2}, {12, 2, 1}, {12, 2, 2}, {8, 2, 2}, {16, 1, 1}};
  for (auto& c : ok) if (c[0] == nt && c[1] == mi && c[2] == wm) return true;
  return false;
}

// what pwconv.hip calls once it has planned the launch (same (nt, mi, grid) conventions as its launch_gemm; wm > 1 or nt > 8
// selects the wide-tile family)
void dl3p_launch_gemm_sb(const GemmParams& p, bool stats, bool bnb, bool ga, int nt, int mi, int wm, dim3 grid, hipStream_t st) {
  if (wm > 1 || nt > 8) {
    if (bnb) launch_sb_wide<true, true>(p, nt, mi, wm, grid, st);
    else if (stats) launch_sb_wide<true, false>(p, nt, mi, wm, grid, st);
    else launch_sb_wide<false, false>(p, nt, mi, wm, grid, st);
    return;
  }
#define DL3P_SB(S, B, G) \
  { if (mi == 1) launch_sb_mi<S, 1, B, G>(p, nt, grid, st); else launch_sb_mi<S, 2, B, G>(p, nt, grid, st); return; }
  if (ga) {
    if (stats) DL3P_SB(true, false, true) else DL3P_SB(false, false, true)
  }
  if (bnb) DL3P_SB(true, true, false)
  if (stats) DL3P_SB(true, false, false)
  DL3P_SB(false, false, false)
#undef DL3P_SB
}

// ------------------------------------------------------------------------------ weight gradient
// GW[K][N] = act(X * scale + shift)^T @ DY over M rows, both operands fp32 ACTIVATIONS: each is split by the thread that stages it
// (the prologue first, in fp32) into three bf16 planes in LDS, [32 rows of M][tile columns]; the MFMA operands -- 16 output rows /
// columns x 32 rows of M -- come out of LDS transposed (ds_read_b64_tr_b16).  One workgroup = TK x TN entries of GW over a slice of
// M; the slices leave fp32 slabs for the batched slab reduction (same contract as pw_wgrad_kernel: dl3p_pwconv_bwd_weight_slabs).
// A 128 x 128 tile reads (128 + 128) x 4 B per row of M for 2 x 128 x 128 flop: 32 flop / B against 21 for the fp32 kernel's widest.
struct WgradSB {
  const float* X; int ldx; const float* scale; const float* shift; int act;
  const float* DY; int lddy;
  float* slabs;
  int M, K, N;
  int ktiles, ntiles, mrows;      // rows of M per slice (multiple of 32)
  int k_base;                     // first row of GW this launch covers (a K that is not a multiple of the tile: two launches)
  // GX instantiations (dense k x k convolutions, dl3p_conv2d_gemm_bwd_weight*): X is the input tensor [N][g_SH][g_SW][ldx] and the row
  // m = (n, y, x) over g_RH x g_RW / column k = tap * g_C + c of the patch matrix is gathered from it while the tile is staged
  // (the fp32 kernel's convention, pwconv.hip WgradParams); scale / shift are indexed by the channel c
  int g_RH, g_RW, g_SH, g_SW, g_C, g_kw, g_mul, g_ay, g_ax, g_d;
  float g_invRW, g_invRH;
};

// q = a / d, *r = a % d for 0 <= a < 2^24 and 0 < d < 2^14 (pwconv.hip divmod_small)
__device__ __forceinline__ int sb_divmod(int a, int d, float inv, int* r) {
  int q = (int)((float)a * inv);
  int rem = a - q * d;
  if (rem < 0) { --q; rem += d; }
  if (rem >= d) { ++q; rem -= d; }
  *r = rem;
  return q;
}

typedef short s16x4w __attribute__((ext_vector_type(4)));
__device__ __forceinline__ s16x4w tr_read16(const unsigned short* lds_ptr) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4w __attribute__((address_space(3)))*)lds_ptr);
}

// WN: the four waves as (4 / WN) x WN over (k, n): a wave owns KFW = KF WN sixteen-row blocks of k and NWW = NW / WN column blocks, and
// reads 6 (KFW + NWW) fragments from LDS per step for 6 KFW NWW MFMAs -- 2 x 2 halves the LDS reads of the 4 x 1 arrangement at 128 x 128
template <int KF, int NW, int WN, bool GX = false>
__global__ __launch_bounds__(256, 2) void pw_wgrad_sb_kernel(WgradSB p) {
  constexpr int TK = 64 * KF, TN = 16 * NW, MS = 32;
  constexpr int KFW = KF * WN, NWW = NW / WN;
  constexpr int XP = TK + 8, DP = TN + 8;             // bf16 pitches: 16-byte aligned rows
  constexpr int X_PLANE = MS * XP, D_PLANE = MS * DP;
  constexpr int XCPR = TK / 8, DCPR = TN / 8;         // 8-element chunks per tile row
  constexpr int XROWS = 256 / XCPR, DROWS = 256 / DCPR;   // rows staged per pass
  constexpr int NXP = MS / XROWS, NDP = MS / DROWS;
  // (dynamic: the 128 x 256 tile's operand planes are 76.8 KB)
  extern __shared__ __attribute__((aligned(16))) unsigned short wg_lds[];
  unsigned short* Xs = wg_lds;
  unsigned short* Ds = wg_lds + 3 * X_PLANE;
  const int t = threadIdx.x, l = t & 63, w = t >> 6;
  const int wk = w / WN, wn = w - wk * WN;
  const int tile = blockIdx.x, kt = tile / p.ntiles, nt = tile - kt * p.ntiles;
  const int k0 = p.k_base + kt * TK, n0 = nt * TN;
  const int mbeg = blockIdx.y * p.mrows, mend = min(p.M, mbeg + p.mrows);
  const int xm = t / XCPR, xc = (t - xm * XCPR) * 8;
  const int dm = t / DCPR, dc = (t - dm * DCPR) * 8;
  const int xk = k0 + xc, dn = n0 + dc;
  const bool xok0 = xk < p.K, xok1 = xk + 4 < p.K, dok0 = dn < p.N, dok1 = dn + 4 < p.N;
  const int xk0 = min(xk, p.K - 4), xk1 = min(xk + 4, p.K - 4), dn0 = min(dn, p.N - 4), dn1 = min(dn + 4, p.N - 4);
  const float* scp = p.scale ? p.scale : p.X;
  const float* shp = p.scale ? p.shift : p.X;
  // GX: this thread's two groups of four k are four channels of one tap each (g_C % 4 == 0): channel offset and tap displacement
  int gc[2] = {xk0, xk1}, gdy[2] = {0, 0}, gdx[2] = {0, 0};
  if (GX) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int k = h ? xk1 : xk0;
      const int tap = k / p.g_C, ky = tap / p.g_kw;
      gc[h] = k - tap * p.g_C;
      gdy[h] = p.g_ay + ky * p.g_d;
      gdx[h] = p.g_ax + (tap - ky * p.g_kw) * p.g_d;
    }
  }
  const float4 sc0 = ld4(scp + gc[0]), sc1 = ld4(scp + gc[1]), sh0 = ld4(shp + gc[0]), sh1 = ld4(shp + gc[1]);
  const float act_lo = p.act == DL3P_ACT_NONE ? -DL3P_INF : 0.f;
  const float act_hi = (p.act == DL3P_ACT_NONE || p.act == DL3P_ACT_RELU) ? DL3P_INF : 6.f;
  auto prologue4 = [&](float4 v, float4 sc4, float4 sh4) __attribute__((always_inline)) {
    if (p.scale) v = fma4(v, sc4, sh4);
    if (p.act >= DL3P_ACT_HSWISH) return act_apply4(v, p.act);
    return make_float4(__builtin_amdgcn_fmed3f(v.x, act_lo, act_hi), __builtin_amdgcn_fmed3f(v.y, act_lo, act_hi),
                       __builtin_amdgcn_fmed3f(v.z, act_lo, act_hi), __builtin_amdgcn_fmed3f(v.w, act_lo, act_hi));
  };
  const bool has_pro = p.scale != nullptr || p.act != DL3P_ACT_NONE;
  uint32_t gx_ok = 0;             // GX: bit (2 i + h) = group h of pass i hit the source tensor

  f32x4 acc[KFW][NWW];
#pragma unroll
  for (int kf = 0; kf < KFW; ++kf)
#pragma unroll
    for (int ni = 0; ni < NWW; ++ni) acc[kf][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float4 rx[NXP][2], rd[NDP][2];
  // every request unconditional on clamped indices (see the note at the prefetches of pw_gemm_sb_kernel)
  auto prefetch = [&](int m0) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NXP; ++i) {
      if (GX) {
        const int m = m0 + xm + XROWS * i;
        int x, y;
        const int row = sb_divmod(min(m, p.M - 1), p.g_RW, p.g_invRW, &x);
        const int n = sb_divmod(row, p.g_RH, p.g_invRH, &y);
        const uint32_t img = (uint32_t)n * (uint32_t)(p.g_SH * p.g_SW);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int sy = y * p.g_mul + gdy[h], sx = x * p.g_mul + gdx[h];
          const bool ok = m < mend && sy >= 0 && sx >= 0 && sy < p.g_SH && sx < p.g_SW;
          const uint32_t off = ok ? (img + (uint32_t)(sy * p.g_SW + sx)) * (uint32_t)p.ldx + (uint32_t)gc[h] : 0u;
          rx[i][h] = ld4(p.X + off);
          gx_ok = (gx_ok & ~(1u << (2 * i + h))) | (ok ? (1u << (2 * i + h)) : 0u);
        }
      } else {
        const float* xr = p.X + (size_t)min(m0 + xm + XROWS * i, p.M - 1) * p.ldx;
        rx[i][0] = ld4(xr + xk0); rx[i][1] = ld4(xr + xk1);
      }
    }
#pragma unroll
    for (int i = 0; i < NDP; ++i) {
      const float* dr = p.DY + (size_t)min(m0 + dm + DROWS * i, p.M - 1) * p.lddy;
      rd[i][0] = ld4(dr + dn0); rd[i][1] = ld4(dr + dn1);
    }
  };
  // transposed-read lanes: lane (g = l >> 4, q = (l & 15) >> 2, pp = l & 3) supplies rows 8 g + q and 8 g + 4 + q, columns 4 pp .. 4 pp + 3
  const int g = l >> 4, q = (l & 15) >> 2, pp = l & 3;
  const int r0 = (8 * g + q), r1 = (8 * g + 4 + q);
  const int nsteps = (mend - mbeg + MS - 1) / MS;
  if (nsteps > 0) prefetch(mbeg);
  for (int s = 0; s < nsteps; ++s) {
    const int m0 = mbeg + s * MS;
#pragma unroll
    for (int i = 0; i < NXP; ++i) {
      float4 v0 = rx[i][0], v1 = rx[i][1];
      if (has_pro) { v0 = prologue4(v0, sc0, sh0); v1 = prologue4(v1, sc1, sh1); }
      const bool rok = m0 + xm + XROWS * i < mend;
      const bool ok0 = GX ? ((gx_ok >> (2 * i)) & 1u) != 0 : rok, ok1 = GX ? ((gx_ok >> (2 * i + 1)) & 1u) != 0 : rok;
      if (!(ok0 && xok0)) v0 = zero4();
      if (!(ok1 && xok1)) v1 = zero4();
      uint4 hh, mm, ll;
      split2(v0.x, v0.y, hh.x, mm.x, ll.x);
      split2(v0.z, v0.w, hh.y, mm.y, ll.y);
      split2(v1.x, v1.y, hh.z, mm.z, ll.z);
      split2(v1.z, v1.w, hh.w, mm.w, ll.w);
      unsigned short* d = Xs + (xm + XROWS * i) * XP + xc;
      *reinterpret_cast<uint4*>(d) = hh;
      *reinterpret_cast<uint4*>(d + X_PLANE) = mm;
      *reinterpret_cast<uint4*>(d + 2 * X_PLANE) = ll;
    }
#pragma unroll
    for (int i = 0; i < NDP; ++i) {
      float4 v0 = rd[i][0], v1 = rd[i][1];
      const bool rok = m0 + dm + DROWS * i < mend;
      if (!(rok && dok0)) v0 = zero4();
      if (!(rok && dok1)) v1 = zero4();
      uint4 hh, mm, ll;
      split2(v0.x, v0.y, hh.x, mm.x, ll.x);
      split2(v0.z, v0.w, hh.y, mm.y, ll.y);
      split2(v1.x, v1.y, hh.z, mm.z, ll.z);
      split2(v1.z, v1.w, hh.w, mm.w, ll.w);
      unsigned short* d = Ds + (dm + DROWS * i) * DP + dc;
      *reinterpret_cast<uint4*>(d) = hh;
      *reinterpret_cast<uint4*>(d + D_PLANE) = mm;
      *reinterpret_cast<uint4*>(d + 2 * D_PLANE) = ll;
    }
    lds_barrier();
    prefetch(mbeg + min(s + 1, nsteps - 1) * MS);
    s16x8 af[KFW][3];
#pragma unroll
    for (int kf = 0; kf < KFW; ++kf)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
        const unsigned short* b = Xs + pl * X_PLANE + (wk * KFW + kf) * 16 + 4 * pp;
        const s16x4w a0 = tr_read16(b + r0 * XP), a1 = tr_read16(b + r1 * XP);
        af[kf][pl] = (s16x8){a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
      }
#pragma unroll
    for (int ni = 0; ni < NWW; ++ni) {
      s16x8 bf[3];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
        const unsigned short* b = Ds + pl * D_PLANE + (wn * NWW + ni) * 16 + 4 * pp;
        const s16x4w b0 = tr_read16(b + r0 * DP), b1 = tr_read16(b + r1 * DP);
        bf[pl] = (s16x8){b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
      }
#pragma unroll
      for (int kf = 0; kf < KFW; ++kf) {
        acc[kf][ni] = split_mac(acc[kf][ni], af[kf][0], af[kf][1], af[kf][2], bf[0], bf[1], bf[2]);
      }
    }
    lds_barrier();
  }
  // D[row = 4 (l >> 4) + j -> k][col = l & 15 -> n]
  float* slab = p.slabs + (size_t)blockIdx.y * p.K * p.N;
#pragma unroll
  for (int kf = 0; kf < KFW; ++kf)
#pragma unroll
    for (int ni = 0; ni < NWW; ++ni) {
      const int n = n0 + (wn * NWW + ni) * 16 + (l & 15);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = k0 + (wk * KFW + kf) * 16 + 4 * g + j;
        if (k < p.K && n < p.N) slab[(size_t)k * p.N + n] = acc[kf][ni][j];
      }
    }
}

// tile (64 KF x 16 NW; index into {128x128, 64x128, 128x64, 64x64}) and slices for one launch; -> number of slabs (0: shape not
// served).  max_slabs: what the caller's workspace holds; tile < 0 / per_cu <= 0: the heuristics (measured: scripts/micro/sb_wgrad.py)
int dl3p_wgrad_sb_plan(int M, int K, int N, int max_slabs, int tile, int per_cu, int* kf, int* nw, int* ktiles, int* ntiles, int* mrows) {
  if (M < 1024 || K < 32 || N < 32 || K % 4 || N % 4 || max_slabs < 1) return 0;
  static const int cand[5][2] = {{2, 8}, {1, 8}, {2, 4}, {1, 4}, {2, 16}};      // (128 x 256: by tile index 4 only, never by the cost rule)
  float best = 1e30f;
  for (int i = 0; i < (tile == 4 ? 5 : 4); ++i) {
    const int tk = 64 * cand[i][0], tn = 16 * cand[i][1];
    const float area = (float)((K + tk - 1) / tk * tk) * (float)((N + tn - 1) / tn * tn);
    const float cost = area * (1.f + 0.5f * (64.f / tk + 64.f / tn));
    if ((tile < 0 && cost < best) || tile == i) { best = cost; *kf = cand[i][0]; *nw = cand[i][1]; }
  }
  *ktiles = (K + 64 * *kf - 1) / (64 * *kf);
  *ntiles = (N + 16 * *nw - 1) / (16 * *nw);
  const int tiles = *ktiles * *ntiles;
  if (per_cu <= 0) per_cu = (*kf == 2 && *nw == 8) ? 2 : 4;      // 128 x 128: two resident workgroups per CU (registers)
  int s = (DL3P_NUM_CUS * per_cu) / tiles;
  if (s < 1) s = 1;
  const int max_s = (M + 255) / 256;
  if (s > max_s) s = max_s;
  if (s > max_slabs) s = max_slabs;
  const int chunk = ((M + s - 1) / s + 31) / 32 * 32;
  *mrows = chunk;
  return (M + chunk - 1) / chunk;
}

template <int KF, int NW, int WN, bool GX>
static void launch_wgrad_tile(const WgradSB& p, dim3 grid, hipStream_t st) {
  constexpr size_t lds = (size_t)3 * 32 * ((64 * KF + 8) + (16 * NW + 8)) * sizeof(unsigned short);
  static bool attr_set = false;
  if (!attr_set) {
    if (lds > 64 * 1024)
      (void)hipFuncSetAttribute((const void*)pw_wgrad_sb_kernel<KF, NW, WN, GX>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  dl3p_launch(pw_wgrad_sb_kernel<KF, NW, WN, GX>, grid, dim3(256), lds, st, p);
}

void dl3p_launch_wgrad_sb(const float* x, int ldx, const float* scale, const float* shift, int act, const float* dy, int lddy,
                          float* slabs, int M, int K, int N, int kf, int nw, int ktiles, int ntiles, int mrows, int splits, hipStream_t st) {
  WgradSB p = {x, ldx, scale, shift, act, dy, lddy, slabs, M, K, N, ktiles, ntiles, mrows, 0};
  const dim3 block(256);
  // the 128 x 128 tile with its waves 2 x 2 (247-258 against 270-278 us on 266256 x 256 x 256); 64 x 128 stays 4 x 1 (2 x 2 costs it
  // a resident workgroup: 327 against 284)
  if (kf == 2 && nw == 8) {
    // a last 128-row tile with at most 64 live rows (K = 304: 48 of 128) multiplies zeros for the rest.  Giving that strip to the
    // 64 x 128 tile in a second launch (same slices, same slabs, disjoint rows of GW) saves a sixth of the MFMAs and was measured
    // SLOWER in the step (12.83 against 12.76 ms, same box): the second launch streams dY from HBM once more, where the six tiles
    // of a slice in ONE launch share it through L2.  Opt-in (DL3P_WGRAD_SB_MIXED=1), parity-tested, not the default.
    const int rem = K % 128;
    static const int mixed = getenv("DL3P_WGRAD_SB_MIXED") ? atoi(getenv("DL3P_WGRAD_SB_MIXED")) : 0;
    if (mixed && rem > 0 && rem <= 64 && ktiles > 1) {
      p.ktiles = ktiles - 1;
      launch_wgrad_tile<2, 8, 2, false>(p, dim3(p.ktiles * ntiles, splits), st);
      p.k_base = 128 * (ktiles - 1);
      p.ktiles = 1;
      launch_wgrad_tile<1, 8, 1, false>(p, dim3(ntiles, splits), st);
      return;
    }
    launch_wgrad_tile<2, 8, 2, false>(p, dim3(ktiles * ntiles, splits), st);
  } else if (kf == 2 && nw == 16) launch_wgrad_tile<2, 16, 2, false>(p, dim3(ktiles * ntiles, splits), st);
  else if (kf == 1 && nw == 8) launch_wgrad_tile<1, 8, 1, false>(p, dim3(ktiles * ntiles, splits), st);
  else if (kf == 2 && nw == 4) launch_wgrad_tile<2, 4, 1, false>(p, dim3(ktiles * ntiles, splits), st);
  else launch_wgrad_tile<1, 4, 1, false>(p, dim3(ktiles * ntiles, splits), st);
}

// the same tiles with the X operand gathered from a convolution's input tensor (dl3p_conv2d_gemm_bwd_weight*); geo = {RH, RW, SH, SW,
// C, kw, mul, ay, ax, d} as in WgradSB
void dl3p_launch_wgrad_sb_gx(const float* x, int ldx, const float* scale, const float* shift, int act, const float* dy, int lddy,
                             float* slabs, int M, int K, int N, const int* geo, int kf, int nw, int ktiles, int ntiles, int mrows, int splits,
                             hipStream_t st) {
  WgradSB p = {x, ldx, scale, shift, act, dy, lddy, slabs, M, K, N, ktiles, ntiles, mrows, 0};
  p.g_RH = geo[0]; p.g_RW = geo[1]; p.g_SH = geo[2]; p.g_SW = geo[3]; p.g_C = geo[4]; p.g_kw = geo[5]; p.g_mul = geo[6];
  p.g_ay = geo[7]; p.g_ax = geo[8]; p.g_d = geo[9];
  p.g_invRW = 1.f / (float)p.g_RW; p.g_invRH = 1.f / (float)p.g_RH;
  const dim3 grid(ktiles * ntiles, splits);
  if (kf == 2 && nw == 8) launch_wgrad_tile<2, 8, 2, true>(p, grid, st);
  else if (kf == 1 && nw == 8) launch_wgrad_tile<1, 8, 1, true>(p, grid, st);
  else if (kf == 2 && nw == 4) launch_wgrad_tile<2, 4, 1, true>(p, grid, st);
  else launch_wgrad_tile<1, 4, 1, true>(p, grid, st);
}
