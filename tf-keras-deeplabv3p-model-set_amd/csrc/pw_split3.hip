// The split-bf16 forward GEMM with its staging PINNED under the multiply phase (round 6).
//
// y[M][256] = act(x * scale + shift)[M][K] . W^T with W pre-split into three bf16 planes [3][256][pitch] (the pointwise convs of
// deeplabv3p/models/layers.py:105,157,209-218 on the long decoder maps: 266256 rows at BASELINE configs[1]).  Same arithmetic as
// pw_split.hip (exact 3-way split of the staged operand, six bf16 products per tile and K-step, smallest first); what differs is who
// waits for whom:
//  * ONE workgroup of four waves per CU (one wave per SIMD: 512 registers), output tile 128 rows x 256 columns, each wave 64 x 128 =
//    2 x 4 accumulators of v_mfma_f32_32x32x16_bf16 (an MFMA holds the vector port 8 of its 32 cycles: 24 free for the staging).
//  * two LDS operand stages (2 x 72 KB), ONE barrier per K-step.  The staging of K-step i + 1 -- buffer loads two steps ahead, the
//    producer's BatchNorm + activation, the 3-way split, LDS stores -- is cut into units of 3-4 vector instructions and each unit is
//    assigned to ONE of the 96 MFMA-to-MFMA gaps of step i by the constexpr tables below; every gap ends in a sched_barrier(0), so
//    the compiler cannot move work across an MFMA and the instruction stream is the one written here (scripts/isa_gaps.py checks it).
//  * the barrier closes gap 70; behind it the first fragments of step i + 1 are read from the other stage while the last 25 MFMAs
//    of step i run.  Fragments live in a ring: a column tile's planes in one register set for both k halves.
//  * the output tile leaves at the end of its K loop (128 row stores per wave).  A form that moved the tile into spare registers and
//    stored it in four slices between the step pairs of the next tile was built and measured SLOWER (252-264 against 242-251 us on
//    262144 x 304 -> 256): the vector-memory counter completes in issue order, so every load issued behind a store waits for it.
// Global accesses are raw buffer loads / stores (row and k-step terms in SGPR offsets: no per-lane address arithmetic, and the
// hardware's range check -- which includes the SGPR offset on gfx950, scripts/micro/buf_range -- zero-fills the M and K tails).
// Measured (scripts/micro/sb3_loop.hip, MI355X): the loop alone 183 us against 142 us of bare MFMAs at the 1.73 GHz the chip holds
// under them; the tiled kernels of pw_split.hip 287-292 us on the same product.
#include "sb_common.h"
#include "pw_gemm.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));

namespace {
constexpr int S3_BM = 128, S3_BN = 256, S3_BKT = 32;
constexpr int S3_A_PLANE = S3_BM * 32, S3_B_PLANE = S3_BN * 32;     // bf16 elements; rows of 64 bytes, 16-byte chunks swizzled by (row >> 2) & 3
constexpr int S3_STAGE = 3 * (S3_A_PLANE + S3_B_PLANE);            // 73728 bytes
constexpr int S3_BARG = 70;                                        // the step's barrier closes this gap
constexpr int S3_LDS_STAGES = 2 * S3_STAGE * 2;                   // bytes of the two operand stages; the prologue coefficients (2 x pitch floats) follow
constexpr int S3_LDS_MAX = 160 * 1024;

// ---- the schedule: which gap (0 .. 95) of a K-step carries which piece of the staging of the NEXT step
// 40 arithmetic units: pair p (0..7: two consecutive k of one row; pairs 0-3 row group 0, 4-7 row group 1) x stage s (0..4).  Stage 0
// of all eight pairs first (units 0..7: the raw registers are free early, so the requests for the step after next leave early and every
// consumer finds its load a full step old: s_waitcnt vmcnt(15) throughout), then row group 0 stages 1..4 (units 8..23, stage-major: a
// plane of the group is complete after 4 / 8 / 16 of them), then row group 1 (24..39)
constexpr int unit_gap(int u) { return (u * 5) / 3; }                                             // 0 .. 65
constexpr int unit_pair(int u) { return u < 8 ? u : (u < 24 ? (u - 8) & 3 : 4 + ((u - 24) & 3)); }
constexpr int unit_stage(int u) { return u < 8 ? 0 : (u < 24 ? 1 + (u - 8) / 4 : 1 + (u - 24) / 4); }
constexpr int aw_gap(int i) { return unit_gap(8 + (i / 3) * 16 + (i % 3 == 0 ? 3 : i % 3 == 1 ? 7 : 15)) + 1; }   // LDS store of plane i % 3 of group i / 3
constexpr int al_gap(int j) { return unit_gap(2 * j + 1) + 1; }                                   // raw quad j is free after stage 0 of pairs 2j, 2j + 1: gaps 2, 6, 9, 12
constexpr int cl_gap(int j) { return 88 + j; }                                                    // prologue coefficients of the next step's staging (from LDS)
constexpr int bw_gap(int j) { return 5 * j + 4; }                                                 // kernel planes: LDS store 4 .. 59, the register's next request behind it
constexpr int bl_gap(int j) { return 5 * j + 8; }
// fragment reads.  The MFMAs of a step run k half (2) x column tile ni (4) x row tile mi (2) x 6 products; a column tile's three planes
// live in register set ni (both k halves), a row tile's in xa[mi][k half].  Reads 0..23 come from the stage being multiplied, in front
// of the barrier; 24..35 from the other stage behind it (the next step's first fragments):
//   0-2 B(0,2) | 3-5 A(0,1) | 6-8 A(1,1) | 9-11 B(0,3) | 12-14 B(1,0) | 15-17 B(1,1) | 18-20 B(1,2) | 21-23 B(1,3) | 24-26 A'(0,0) | 27-29 B'(0,0) |
//   30-32 A'(1,0) | 33-35 B'(0,1)       (B(k half, ni), A(mi, k half))
constexpr int fr_gap(int k) { return k < 21 ? 2 * k + 1 : k < 24 ? 2 * k + 7 : S3_BARG + 1 + (k - 24); }
// the index whose gap is g, or -1 (every table puts at most one of its entries into a gap)
template <class F>
constexpr int at(F f, int n, int g) {
  for (int i = 0; i < n; ++i)
    if (f(i) == g) return i;
  return -1;
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t s3_rsrc(const void* p, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}

// ACT: DL3P_ACT_NONE / RELU / RELU6 of the prologue; PRO: the operand passes through x * scale + shift first; STATS: per-column (sum, sum^2)
// of the outputs, one partial row per workgroup (the contract of pw_gemm_sb_kernel)
template <int ACT, bool PRO, bool STATS>
__global__ __launch_bounds__(256, 1) void pw_gemm_sb3_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char s3_lds[];
  unsigned short* S0 = reinterpret_cast<unsigned short*>(s3_lds);
  const int t = threadIdx.x, l = t & 63, w = t >> 6;
  const int wr = w >> 1, wc = w & 1;
  const int nk = p.bsp_pitch / S3_BKT;                     // even, >= 8 (the host's routing rule)
  const int my_tiles = (p.num_m_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  const int M = p.M, ldy = p.ldy;
  const uint32_t a_bytes = ((uint32_t)(M - 1) * (uint32_t)p.lda + (uint32_t)p.K) * 4u;
  const __amdgpu_buffer_rsrc_t rA = s3_rsrc(p.A, a_bytes),
                               rB = s3_rsrc(p.Bsp, (uint32_t)(3 * p.bsp_plane) * 2u),
                               rY = s3_rsrc(p.Y, ((uint32_t)(M - 1) * (uint32_t)ldy + (uint32_t)S3_BN) * 4u);
  const int ar = t >> 2, ac = t & 3;
  // LDS offsets (bf16 elements): the thread's staging chunks ...
  const int sw = (ar >> 2) & 3;                            // (the same for rows ar + 64 i)
  const int a_lds = ar * 32 + ((ac ^ sw) * 8);             // + i * 64 * 32 + plane * A_PLANE
  const int b_lds = 3 * S3_A_PLANE + a_lds;                // + (j & 3) * 64 * 32 + (j >> 2) * B_PLANE
  // ... and its fragments (32x32x16: lane = row (l & 31), 8 k at chunk 2 kh + (l >> 5))
  const int l31 = l & 31, hh5 = l >> 5, fsw = (l31 >> 2) & 3;
  int xa_off[2], wb_off[2];
#pragma unroll
  for (int kh = 0; kh < 2; ++kh) {
    xa_off[kh] = (wr * 64 + l31) * 32 + (((2 * kh + hh5) ^ fsw) * 8);
    wb_off[kh] = 3 * S3_A_PLANE + (wc * 128 + l31) * 32 + (((2 * kh + hh5) ^ fsw) * 8);
  }
  // global offsets (bytes)
  uint32_t arow[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) arow[i] = (uint32_t)((int)blockIdx.x * S3_BM + ar + 64 * i) * (uint32_t)p.lda * 4u + (uint32_t)ac * 32u;
  const uint32_t a_tile_stride = gridDim.x * (uint32_t)S3_BM * (uint32_t)p.lda * 4u;
  const uint32_t blane = (uint32_t)ar * (uint32_t)p.bsp_pitch * 2u + (uint32_t)ac * 16u;
  const uint32_t b_plane = (uint32_t)p.bsp_plane * 2u, b_rows = 64u * (uint32_t)p.bsp_pitch * 2u;
  uint4 ra[4];                  // raw quads: [row group * 2 + half]
  uint4 cs[2], ct[2];           // prologue coefficients of the thread's 8 k
  u32x4v rb[12];
  float v0[8], v1[8], r0[8], r1[8];
  uint32_t hp[8], mp[8], lp[8];
  // clamp bounds of the prologue; a row beyond M is clamped to [0, 0]: its operand row is exactly zero, whatever shift says
  constexpr float LO = ACT == DL3P_ACT_NONE ? -DL3P_INF : 0.f, HI = ACT == DL3P_ACT_RELU6 ? 6.f : DL3P_INF;
  float lo[2] = {LO, LO}, hi[2] = {HI, HI};

  int lkt = 0;                  // k-step of the NEXT global request; arow[] holds its tile
  int ckt = 0;                  // k-step whose prologue coefficients the next load_c fetches
  int skt = 1 % nk, srem = M - (int)blockIdx.x * S3_BM;     // k-step / rows left (from the tile's first row) of the data being STAGED
  const int tile_rows = (int)gridDim.x * S3_BM;
  auto advance = [&]() __attribute__((always_inline)) {
    ++lkt;
    if (lkt == nk) { lkt = 0; arow[0] += a_tile_stride; arow[1] += a_tile_stride; }
    ++skt;
    if (skt == nk) { skt = 0; srem -= tile_rows; }
    ++ckt;
    if (ckt == nk) ckt = 0;
  };
  auto row_bounds = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const bool ok = ar + 64 * i < srem;
      lo[i] = ok ? LO : 0.f;
      hi[i] = ok ? HI : 0.f;
    }
  };
  auto load_a = [&](int j) __attribute__((always_inline)) {
    ra[j] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rA, arow[j >> 1] + 16 * (j & 1), lkt * (S3_BKT * 4), 0));
  };
  // prologue coefficients: both vectors in LDS behind the operand stages (copied once, zero from K up: the K tail of the operand
  // becomes act(0 x + 0) = 0 whatever the buffer holds there), read per step for k-step ckt
  float* Cs = reinterpret_cast<float*>(s3_lds + S3_LDS_STAGES);
  const int kpad = p.bsp_pitch;
  if (PRO) {
    for (int i = t; i < kpad; i += 256) { Cs[i] = i < p.K ? p.scale[i] : 0.f; Cs[kpad + i] = i < p.K ? p.shift[i] : 0.f; }
  }
  auto load_c = [&](int j) __attribute__((always_inline)) {
    if (!PRO) return;
    const float* src = Cs + (j < 2 ? 0 : kpad) + ckt * S3_BKT + ac * 8 + 4 * (j & 1);
    if (j < 2) cs[j] = *reinterpret_cast<const uint4*>(src);
    else ct[j - 2] = *reinterpret_cast<const uint4*>(src);
  };
  auto load_b = [&](int j) __attribute__((always_inline)) {
    rb[j] = __builtin_amdgcn_raw_buffer_load_b128(rB, blane + (uint32_t)lkt * (S3_BKT * 2), (j >> 2) * b_plane + (j & 3) * b_rows, 0);
  };
  auto elem = [](const uint4& q, int e) __attribute__((always_inline)) {
    return __builtin_bit_cast(float, e == 0 ? q.x : e == 1 ? q.y : e == 2 ? q.z : q.w);
  };
  // one arithmetic unit of pair p, stage s
  auto unit = [&](auto pc, auto sc) __attribute__((always_inline)) {
    constexpr int pp = decltype(pc)::value, s = decltype(sc)::value;
    constexpr int q = (pp >> 2) * 2 + ((pp & 3) >> 1), e = (pp & 1) * 2, h = (pp & 3) >> 1, g = pp >> 2;
    if constexpr (s == 0) {
      float x = elem(ra[q], e), y = elem(ra[q], e + 1);
      if (PRO) {
        x = __builtin_fmaf(x, elem(cs[h], e), elem(ct[h], e));
        y = __builtin_fmaf(y, elem(cs[h], e + 1), elem(ct[h], e + 1));
      }
      v0[pp] = __builtin_amdgcn_fmed3f(x, lo[g], hi[g]);
      v1[pp] = __builtin_amdgcn_fmed3f(y, lo[g], hi[g]);
    } else if constexpr (s == 1) {
      const bf16x2v hv = {(__bf16)v0[pp], (__bf16)v1[pp]};
      hp[pp] = __builtin_bit_cast(uint32_t, hv);
      r0[pp] = __builtin_bit_cast(float, hp[pp] << 16);
      r1[pp] = __builtin_bit_cast(float, hp[pp] & 0xffff0000u);
    } else if constexpr (s == 2) {
      r0[pp] = v0[pp] - r0[pp];
      r1[pp] = v1[pp] - r1[pp];
      const bf16x2v mv = {(__bf16)r0[pp], (__bf16)r1[pp]};
      mp[pp] = __builtin_bit_cast(uint32_t, mv);
    } else if constexpr (s == 3) {
      v0[pp] = r0[pp] - __builtin_bit_cast(float, mp[pp] << 16);
      v1[pp] = r1[pp] - __builtin_bit_cast(float, mp[pp] & 0xffff0000u);
    } else {
      const bf16x2v lv = {(__bf16)v0[pp], (__bf16)v1[pp]};
      lp[pp] = __builtin_bit_cast(uint32_t, lv);
    }
  };
  auto write_a = [&](unsigned short* buf, int g, int pl) __attribute__((always_inline)) {
    const uint32_t* src = pl == 0 ? hp : pl == 1 ? mp : lp;
    const uint4 val = {src[4 * g], src[4 * g + 1], src[4 * g + 2], src[4 * g + 3]};
    *reinterpret_cast<uint4*>(buf + a_lds + g * (64 * 32) + pl * S3_A_PLANE) = val;
  };
  auto write_b = [&](unsigned short* buf, int j) __attribute__((always_inline)) {
    *reinterpret_cast<u32x4v*>(buf + b_lds + (j & 3) * (64 * 32) + (j >> 2) * S3_B_PLANE) = rb[j];
  };

  f32x16 acc[2][4];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;
  s16x8 xa[2][2][3], wb[4][3];          // A: [row tile][k half][plane]; B: [column tile = register set][plane]

  auto frag_read = [&](const unsigned short* cur, const unsigned short* nxt, auto kc) __attribute__((always_inline)) {
    constexpr int k = decltype(kc)::value, grp = k / 3, pl = k % 3;
    // grp: 0 B(0,2) 1 A(0,1) 2 A(1,1) 3 B(0,3) 4..7 B(1,0..3) 8 A'(0,0) 9 B'(0,0) 10 A'(1,0) 11 B'(0,1)
    if constexpr (grp == 0) wb[2][pl] = *reinterpret_cast<const s16x8*>(cur + pl * S3_B_PLANE + wb_off[0] + 2 * 32 * 32);
    else if constexpr (grp == 1) xa[0][1][pl] = *reinterpret_cast<const s16x8*>(cur + pl * S3_A_PLANE + xa_off[1]);
    else if constexpr (grp == 2) xa[1][1][pl] = *reinterpret_cast<const s16x8*>(cur + pl * S3_A_PLANE + xa_off[1] + 32 * 32);
    else if constexpr (grp == 3) wb[3][pl] = *reinterpret_cast<const s16x8*>(cur + pl * S3_B_PLANE + wb_off[0] + 3 * 32 * 32);
    else if constexpr (grp < 8) wb[grp - 4][pl] = *reinterpret_cast<const s16x8*>(cur + pl * S3_B_PLANE + wb_off[1] + (grp - 4) * 32 * 32);
    else if constexpr (grp == 8) xa[0][0][pl] = *reinterpret_cast<const s16x8*>(nxt + pl * S3_A_PLANE + xa_off[0]);
    else if constexpr (grp == 9) wb[0][pl] = *reinterpret_cast<const s16x8*>(nxt + pl * S3_B_PLANE + wb_off[0]);
    else if constexpr (grp == 10) xa[1][0][pl] = *reinterpret_cast<const s16x8*>(nxt + pl * S3_A_PLANE + xa_off[0] + 32 * 32);
    else wb[1][pl] = *reinterpret_cast<const s16x8*>(nxt + pl * S3_B_PLANE + wb_off[0] + 32 * 32);
  };

  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  // ---- pipeline head: step 0 staged into stage 0, step 1 requested
  {
    __syncthreads();             // (the coefficient vectors)
    row_bounds();
#pragma unroll
    for (int j = 0; j < 4; ++j) { load_a(j); load_c(j); }
    ckt = 1;
#pragma unroll
    for (int j = 0; j < 12; ++j) load_b(j);
    static_for<5>([&](auto sc) { static_for<8>([&](auto pc) { unit(pc, sc); }); });
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) write_a(S0, g, pl);
#pragma unroll
    for (int j = 0; j < 12; ++j) write_b(S0, j);
    ++lkt;                       // (nk >= 8: no wrap here; skt / srem already describe step 1)
#pragma unroll
    for (int j = 0; j < 4; ++j) { load_a(j); load_c(j); }
    ckt = 2;
#pragma unroll
    for (int j = 0; j < 12; ++j) load_b(j);
    ++lkt;
    lds_barrier();
    static_for<12>([&](auto kc) { frag_read(S0, S0, std::integral_constant<int, 24 + decltype(kc)::value>{}); });
  }

  auto step = [&](auto par) __attribute__((always_inline)) {
    constexpr int P = decltype(par)::value;
    const unsigned short* cur = S0 + P * S3_STAGE;
    unsigned short* nxt = S0 + (P ^ 1) * S3_STAGE;
    row_bounds();
    static_for<96>([&](auto gc) {
      constexpr int g = decltype(gc)::value;
      constexpr int kh = g / 48, rem = g % 48, ni = rem / 12, mi = (rem % 12) / 6, pr = rem % 6;
      constexpr int WB[6] = {2, 0, 1, 1, 0, 0}, XA[6] = {0, 2, 1, 0, 1, 0};          // smallest terms first
      acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[mi][kh][XA[pr]], wb[ni][WB[pr]], acc[mi][ni], 0, 0, 0);
      // ---- the fillers of this gap
      constexpr int fr = at(fr_gap, 36, g), u = at(unit_gap, 40, g), aw = at(aw_gap, 6, g), bw = at(bw_gap, 12, g), al = at(al_gap, 4, g),
                    cl = at(cl_gap, 4, g), bl = at(bl_gap, 12, g);
      if constexpr (g == S3_BARG) lds_barrier();
      if constexpr (fr >= 0) frag_read(cur, nxt, std::integral_constant<int, (fr >= 0 ? fr : 0)>{});
      if constexpr (u >= 0) {
        constexpr int uu = u >= 0 ? u : 0;
        unit(std::integral_constant<int, unit_pair(uu)>{}, std::integral_constant<int, unit_stage(uu)>{});
      }
      if constexpr (aw >= 0) write_a(nxt, (aw >= 0 ? aw : 0) / 3, (aw >= 0 ? aw : 0) % 3);
      if constexpr (bw >= 0) write_b(nxt, bw >= 0 ? bw : 0);
      if constexpr (al >= 0) load_a(al >= 0 ? al : 0);
      if constexpr (cl >= 0) load_c(cl >= 0 ? cl : 0);
      if constexpr (bl >= 0) load_b(bl >= 0 ? bl : 0);
      __builtin_amdgcn_sched_barrier(0);
    });
    advance();
  };

  // ---- the output tile: lane = output channel (l & 31), register e = pixel row (e & 3) + 8 (e >> 2) + 4 (l >> 5) of a 32-row tile
  const uint32_t yv = ((uint32_t)(wr * 64 + 4 * hh5) * (uint32_t)ldy + (uint32_t)(wc * 128 + l31)) * 4u;
  float bias_v[4] = {0.f, 0.f, 0.f, 0.f};
  if (p.bias) {
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) bias_v[ni] = p.bias[wc * 128 + ni * 32 + l31];
  }
  float st_s[4] = {0.f, 0.f, 0.f, 0.f}, st_q[4] = {0.f, 0.f, 0.f, 0.f};
  for (int tile = 0; tile < my_tiles; ++tile) {
    for (int kp = 0; kp < nk; kp += 2) { step(I0{}); step(I1{}); }
    // the finished tile: 128 row stores of 128 bytes per wave (rows beyond M fall to the range check; their values are exact zeros)
    const uint32_t out_m0 = (uint32_t)((int)blockIdx.x + tile * (int)gridDim.x) * S3_BM;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const float val = acc[mi][ni][e] + bias_v[ni];
          const uint32_t row = out_m0 + mi * 32 + (e & 3) + 8 * (e >> 2);
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(val), rY, yv + ni * 128, row * (uint32_t)ldy * 4u, 0);
          if (STATS) { st_s[ni] += val; st_q[ni] = __builtin_fmaf(val, val, st_q[ni]); }
          acc[mi][ni][e] = 0.f;
        }
      }
  }
  if (STATS) {
    // a column's two half-wave lanes, then the two wave rows through LDS (the operand stages are done with)
    float* red = reinterpret_cast<float*>(s3_lds);          // [which][wr][256]
    lds_barrier();
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      float s1 = st_s[ni], s2 = st_q[ni];
      s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
      if (hh5 == 0) {
        red[(0 * 2 + wr) * S3_BN + wc * 128 + ni * 32 + l31] = s1;
        red[(1 * 2 + wr) * S3_BN + wc * 128 + ni * 32 + l31] = s2;
      }
    }
    lds_barrier();
    if (p.partials) {
      for (int i = t; i < 2 * S3_BN; i += 256) {
        const int which = i / S3_BN, nn = i - which * S3_BN;
        p.partials[((size_t)blockIdx.x * 2 + which) * p.N + nn] = red[(which * 2 + 0) * S3_BN + nn] + red[(which * 2 + 1) * S3_BN + nn];
      }
    }
  }
}

// ------------------------------------------------------------------------------ data gradient with the folded BatchNorm-backward apply
// gx[M][256] = dz[M][256] . W (pre-split [3][256][256]) where dz = c0 (g act'(z s + t) - c1 - xhat c2) is FORMED while the operand is
// staged (dl3p_pwconv_bwd_data_sb_apply's contract, csrc/pw_split_rs.hip FOLD: the arithmetic is that kernel's, as A g m + nC z + D) and
// written once to f_dz (over g when they alias: every element is read and written by the same lane); BNB: the BatchNorm-backward sums
// (sum d, sum d xhat, d = gx act'(zf sf + tf)) of the layer in FRONT of the conv ride in the epilogue.  Same tile, stages, ring and
// barrier as the forward above; what changes in the schedule: a stage-0 unit is 12 vector instructions + 5 coefficient reads (one
// every second gap), two operand tensors are requested (8 + 12 loads per step) and the step stores its 4 quads of dz.
constexpr int unit_gap_d(int u) { return u < 8 ? 2 * u : 16 + ((u - 8) * 25) / 16; }              // 0 .. 14, 16 .. 64
constexpr int aw_gap_d(int i) { return unit_gap_d(8 + (i / 3) * 16 + (i % 3 == 0 ? 3 : i % 3 == 1 ? 7 : 15)) + 1; }
constexpr int gl_gap_d(int j) { return unit_gap_d(2 * j + 1) + 1; }                              // quad j of g / z / dz: behind stage 0 of pairs 2j, 2j + 1
constexpr int S3D_KP = 256;                                                                      // reduction length = pitch (the coefficient vectors' LDS offsets are immediates)

template <int FACT, bool BNB>
__global__ __launch_bounds__(256, 1) void pw_gemm_sb3d_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char s3_lds[];
  unsigned short* S0 = reinterpret_cast<unsigned short*>(s3_lds);
  const int t = threadIdx.x, l = t & 63, w = t >> 6;
  const int wr = w >> 1, wc = w & 1;
  constexpr int nk = S3D_KP / S3_BKT;
  const int my_tiles = (p.num_m_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  const int M = p.M, ldy = p.ldy;
  const __amdgpu_buffer_rsrc_t rG = s3_rsrc(p.A, ((uint32_t)(M - 1) * (uint32_t)p.lda + S3D_KP) * 4u),
                               rZ = s3_rsrc(p.f_z, ((uint32_t)(M - 1) * (uint32_t)p.f_ldz + S3D_KP) * 4u),
                               rD = s3_rsrc(p.f_dz, ((uint32_t)(M - 1) * (uint32_t)p.f_lddz + S3D_KP) * 4u),
                               rB = s3_rsrc(p.Bsp, (uint32_t)(3 * p.bsp_plane) * 2u),
                               rY = s3_rsrc(p.Y, ((uint32_t)(M - 1) * (uint32_t)ldy + (uint32_t)S3_BN) * 4u),
                               rF = s3_rsrc(BNB ? p.bb_z : p.Y, BNB ? ((uint32_t)(M - 1) * (uint32_t)p.bb_ldz + (uint32_t)S3_BN) * 4u : 0u);
  const int ar = t >> 2, ac = t & 3;
  const int sw = (ar >> 2) & 3;
  const int a_lds = ar * 32 + ((ac ^ sw) * 8);
  const int b_lds = 3 * S3_A_PLANE + a_lds;
  const int l31 = l & 31, hh5 = l >> 5, fsw = (l31 >> 2) & 3;
  int xa_off[2], wb_off[2];
#pragma unroll
  for (int kh = 0; kh < 2; ++kh) {
    xa_off[kh] = (wr * 64 + l31) * 32 + (((2 * kh + hh5) ^ fsw) * 8);
    wb_off[kh] = 3 * S3_A_PLANE + (wc * 128 + l31) * 32 + (((2 * kh + hh5) ^ fsw) * 8);
  }
  // per-lane byte offsets of the thread's two operand rows in g, z and dz (three pitches)
  uint32_t grow[2], zrow[2], drow[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const uint32_t r = (uint32_t)((int)blockIdx.x * S3_BM + ar + 64 * i);
    grow[i] = r * (uint32_t)p.lda * 4u + (uint32_t)ac * 32u;
    zrow[i] = r * (uint32_t)p.f_ldz * 4u + (uint32_t)ac * 32u;
    drow[i] = r * (uint32_t)p.f_lddz * 4u + (uint32_t)ac * 32u;
  }
  const uint32_t tile_rows_u = gridDim.x * (uint32_t)S3_BM;
  const uint32_t g_stride = tile_rows_u * (uint32_t)p.lda * 4u, z_stride = tile_rows_u * (uint32_t)p.f_ldz * 4u,
                 d_stride = tile_rows_u * (uint32_t)p.f_lddz * 4u;
  const uint32_t blane = (uint32_t)ar * (uint32_t)p.bsp_pitch * 2u + (uint32_t)ac * 16u;
  const uint32_t b_plane = (uint32_t)p.bsp_plane * 2u, b_rows = 64u * (uint32_t)p.bsp_pitch * 2u;
  uint4 ra[4], rz[4];            // raw quads of g and z: [row group * 2 + half]
  u32x4v rb[12];
  float v0[8], v1[8], r0[8], r1[8];
  uint32_t hp[8], mp[8], lp[8];
  // the five folded coefficient vectors in LDS behind the stages: s, t (of act'), fA = c0, fnC = -c0 c2 invstd, fD = -fnC mean - c0 c1
  float* Cs = reinterpret_cast<float*>(s3_lds + S3_LDS_STAGES);
  for (int i = t; i < S3D_KP; i += 256) {
    const float c0 = p.f_coef[i], c1 = p.f_coef[S3D_KP + i], c2 = p.f_coef[2 * S3D_KP + i];
    const float nC = -(c0 * p.f_invstd[i] * c2);
    Cs[i] = p.f_scale ? p.f_scale[i] : 1.f;
    Cs[S3D_KP + i] = p.f_scale ? p.f_shift[i] : 0.f;
    Cs[2 * S3D_KP + i] = c0;
    Cs[3 * S3D_KP + i] = nC;
    Cs[4 * S3D_KP + i] = -(nC * p.f_mean[i]) - c0 * c1;
  }
  int lkt = 0;                  // k-step of the NEXT operand request (grow / zrow hold its tile)
  int skt = 1, srem = M - (int)blockIdx.x * S3_BM;          // k-step / rows left of the data being staged (drow holds its tile)
  const int tile_rows = (int)tile_rows_u;
  auto advance = [&]() __attribute__((always_inline)) {
    ++lkt;
    if (lkt == nk) { lkt = 0; grow[0] += g_stride; grow[1] += g_stride; zrow[0] += z_stride; zrow[1] += z_stride; }
    ++skt;
    if (skt == nk) { skt = 0; srem -= tile_rows; drow[0] += d_stride; drow[1] += d_stride; }
  };
  bool rok[2] = {true, true};
  auto row_bounds = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i) rok[i] = ar + 64 * i < srem;
  };
  auto load_g = [&](int j) __attribute__((always_inline)) {
    ra[j] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rG, grow[j >> 1] + 16 * (j & 1), lkt * (S3_BKT * 4), 0));
    rz[j] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rZ, zrow[j >> 1] + 16 * (j & 1), lkt * (S3_BKT * 4), 0));
  };
  auto load_b = [&](int j) __attribute__((always_inline)) {
    rb[j] = __builtin_amdgcn_raw_buffer_load_b128(rB, blane + (uint32_t)lkt * (S3_BKT * 2), (j >> 2) * b_plane + (j & 3) * b_rows, 0);
  };
  auto elem = [](const uint4& q, int e) __attribute__((always_inline)) {
    return __builtin_bit_cast(float, e == 0 ? q.x : e == 1 ? q.y : e == 2 ? q.z : q.w);
  };
  constexpr float GLO = FACT == DL3P_ACT_NONE ? -DL3P_INF : 0.f, GHI = FACT == DL3P_ACT_RELU6 ? 6.f : DL3P_INF;
  auto unit = [&](auto pc, auto sc) __attribute__((always_inline)) {
    constexpr int pp = decltype(pc)::value, s = decltype(sc)::value;
    constexpr int q = (pp >> 2) * 2 + ((pp & 3) >> 1), e = (pp & 1) * 2, g = pp >> 2;
    if constexpr (s == 0) {
      // coefficients of the pair's two reduction channels k, k + 1 (k = 32 skt + 8 ac + 2 (pp & 3)): five 8-byte LDS reads
      const float* c5 = Cs + skt * S3_BKT + ac * 8 + 2 * (pp & 3);
      const float2 cS = *reinterpret_cast<const float2*>(c5), cT = *reinterpret_cast<const float2*>(c5 + S3D_KP),
                   cA = *reinterpret_cast<const float2*>(c5 + 2 * S3D_KP), cC = *reinterpret_cast<const float2*>(c5 + 3 * S3D_KP),
                   cD = *reinterpret_cast<const float2*>(c5 + 4 * S3D_KP);
      const float g0 = elem(ra[q], e), g1 = elem(ra[q], e + 1), z0 = elem(rz[q], e), z1 = elem(rz[q], e + 1);
      float m0 = g0, m1 = g1;
      if (FACT != DL3P_ACT_NONE) {
        const float u0 = __builtin_fmaf(z0, cS.x, cT.x), u1 = __builtin_fmaf(z1, cS.y, cT.y);
        m0 = (u0 > GLO && u0 < GHI) ? g0 : 0.f;
        m1 = (u1 > GLO && u1 < GHI) ? g1 : 0.f;
      }
      const float d0 = __builtin_fmaf(m0, cA.x, __builtin_fmaf(z0, cC.x, cD.x));
      const float d1 = __builtin_fmaf(m1, cA.y, __builtin_fmaf(z1, cC.y, cD.y));
      v0[pp] = rok[g] ? d0 : 0.f;                 // rows past M stay exactly zero (their loads returned zeros, the formula would give fD)
      v1[pp] = rok[g] ? d1 : 0.f;
    } else if constexpr (s == 1) {
      const bf16x2v hv = {(__bf16)v0[pp], (__bf16)v1[pp]};
      hp[pp] = __builtin_bit_cast(uint32_t, hv);
      r0[pp] = __builtin_bit_cast(float, hp[pp] << 16);
      r1[pp] = __builtin_bit_cast(float, hp[pp] & 0xffff0000u);
    } else if constexpr (s == 2) {
      r0[pp] = v0[pp] - r0[pp];
      r1[pp] = v1[pp] - r1[pp];
      const bf16x2v mv = {(__bf16)r0[pp], (__bf16)r1[pp]};
      mp[pp] = __builtin_bit_cast(uint32_t, mv);
    } else if constexpr (s == 3) {
      r0[pp] = r0[pp] - __builtin_bit_cast(float, mp[pp] << 16);
      r1[pp] = r1[pp] - __builtin_bit_cast(float, mp[pp] & 0xffff0000u);
    } else {
      const bf16x2v lv = {(__bf16)r0[pp], (__bf16)r1[pp]};
      lp[pp] = __builtin_bit_cast(uint32_t, lv);
    }
  };
  // dz of quad j (pairs 2j, 2j + 1: four consecutive reduction channels of one row) leaves for memory once, behind their stage 0
  auto store_dz = [&](int j) __attribute__((always_inline)) {
    const u32x4v val = {__float_as_uint(v0[2 * j]), __float_as_uint(v1[2 * j]), __float_as_uint(v0[2 * j + 1]), __float_as_uint(v1[2 * j + 1])};
    __builtin_amdgcn_raw_buffer_store_b128(val, rD, drow[j >> 1] + 16 * (j & 1), skt * (S3_BKT * 4), 0);
  };
  auto write_a = [&](unsigned short* buf, int g, int pl) __attribute__((always_inline)) {
    const uint32_t* src = pl == 0 ? hp : pl == 1 ? mp : lp;
    const uint4 val = {src[4 * g], src[4 * g + 1], src[4 * g + 2], src[4 * g + 3]};
    *reinterpret_cast<uint4*>(buf + a_lds + g * (64 * 32) + pl * S3_A_PLANE) = val;
  };
  auto write_b = [&](unsigned short* buf, int j) __attribute__((always_inline)) {
    *reinterpret_cast<u32x4v*>(buf + b_lds + (j & 3) * (64 * 32) + (j >> 2) * S3_B_PLANE) = rb[j];
  };

  f32x16 acc[2][4];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;
  s16x8 xa[2][2][3], wb[4][3];
  auto frag_read = [&](const unsigned short* cur, const unsigned short* nxt, auto kc) __attribute__((always_inline)) {
    constexpr int k = decltype(kc)::value, grp = k / 3, pl = k % 3;
    if constexpr (grp == 0) wb[2][pl] = *reinterpret_cast<const s16x8*>(cur + pl * S3_B_PLANE + wb_off[0] + 2 * 32 * 32);
    else if constexpr (grp == 1) xa[0][1][pl] = *reinterpret_cast<const s16x8*>(cur + pl * S3_A_PLANE + xa_off[1]);
    else if constexpr (grp == 2) xa[1][1][pl] = *reinterpret_cast<const s16x8*>(cur + pl * S3_A_PLANE + xa_off[1] + 32 * 32);
    else if constexpr (grp == 3) wb[3][pl] = *reinterpret_cast<const s16x8*>(cur + pl * S3_B_PLANE + wb_off[0] + 3 * 32 * 32);
    else if constexpr (grp < 8) wb[grp - 4][pl] = *reinterpret_cast<const s16x8*>(cur + pl * S3_B_PLANE + wb_off[1] + (grp - 4) * 32 * 32);
    else if constexpr (grp == 8) xa[0][0][pl] = *reinterpret_cast<const s16x8*>(nxt + pl * S3_A_PLANE + xa_off[0]);
    else if constexpr (grp == 9) wb[0][pl] = *reinterpret_cast<const s16x8*>(nxt + pl * S3_B_PLANE + wb_off[0]);
    else if constexpr (grp == 10) xa[1][0][pl] = *reinterpret_cast<const s16x8*>(nxt + pl * S3_A_PLANE + xa_off[0] + 32 * 32);
    else wb[1][pl] = *reinterpret_cast<const s16x8*>(nxt + pl * S3_B_PLANE + wb_off[0] + 32 * 32);
  };

  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  // ---- pipeline head: step 0 staged into stage 0 (its dz stored), step 1 requested
  {
    __syncthreads();
    row_bounds();
    skt = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) load_g(j);
#pragma unroll
    for (int j = 0; j < 12; ++j) load_b(j);
    static_for<8>([&](auto pc) { unit(pc, I0{}); });
#pragma unroll
    for (int j = 0; j < 4; ++j) store_dz(j);
    static_for<4>([&](auto sc) { static_for<8>([&](auto pc) { unit(pc, std::integral_constant<int, 1 + decltype(sc)::value>{}); }); });
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) write_a(S0, g, pl);
#pragma unroll
    for (int j = 0; j < 12; ++j) write_b(S0, j);
    skt = 1;
    ++lkt;
#pragma unroll
    for (int j = 0; j < 4; ++j) load_g(j);
#pragma unroll
    for (int j = 0; j < 12; ++j) load_b(j);
    ++lkt;
    lds_barrier();
    static_for<12>([&](auto kc) { frag_read(S0, S0, std::integral_constant<int, 24 + decltype(kc)::value>{}); });
  }

  auto step = [&](auto par) __attribute__((always_inline)) {
    constexpr int P = decltype(par)::value;
    const unsigned short* cur = S0 + P * S3_STAGE;
    unsigned short* nxt = S0 + (P ^ 1) * S3_STAGE;
    row_bounds();
    static_for<96>([&](auto gc) {
      constexpr int g = decltype(gc)::value;
      constexpr int kh = g / 48, rem = g % 48, ni = rem / 12, mi = (rem % 12) / 6, pr = rem % 6;
      constexpr int WB[6] = {2, 0, 1, 1, 0, 0}, XA[6] = {0, 2, 1, 0, 1, 0};
      acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[mi][kh][XA[pr]], wb[ni][WB[pr]], acc[mi][ni], 0, 0, 0);
      constexpr int fr = at(fr_gap, 36, g), u = at(unit_gap_d, 40, g), aw = at(aw_gap_d, 6, g), bw = at(bw_gap, 12, g), gl = at(gl_gap_d, 4, g),
                    bl = at(bl_gap, 12, g);
      if constexpr (g == S3_BARG) lds_barrier();
      if constexpr (fr >= 0) frag_read(cur, nxt, std::integral_constant<int, (fr >= 0 ? fr : 0)>{});
      if constexpr (u >= 0) {
        constexpr int uu = u >= 0 ? u : 0;
        unit(std::integral_constant<int, unit_pair(uu)>{}, std::integral_constant<int, unit_stage(uu)>{});
      }
      if constexpr (gl >= 0) { store_dz(gl >= 0 ? gl : 0); load_g(gl >= 0 ? gl : 0); }
      if constexpr (aw >= 0) write_a(nxt, (aw >= 0 ? aw : 0) / 3, (aw >= 0 ? aw : 0) % 3);
      if constexpr (bw >= 0) write_b(nxt, bw >= 0 ? bw : 0);
      if constexpr (bl >= 0) load_b(bl >= 0 ? bl : 0);
      __builtin_amdgcn_sched_barrier(0);
    });
    advance();
  };

  const uint32_t yv = ((uint32_t)(wr * 64 + 4 * hh5) * (uint32_t)ldy + (uint32_t)(wc * 128 + l31)) * 4u;
  const uint32_t fv = BNB ? ((uint32_t)(wr * 64 + 4 * hh5) * (uint32_t)p.bb_ldz + (uint32_t)(wc * 128 + l31)) * 4u : 0u;
  float st_s[4] = {0.f, 0.f, 0.f, 0.f}, st_q[4] = {0.f, 0.f, 0.f, 0.f};
  float fsc[4], fsh[4], fmu[4], fis[4];
  if (BNB) {
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const int c = wc * 128 + ni * 32 + l31;
      fsc[ni] = p.bb_scale[c]; fsh[ni] = p.bb_shift[c]; fmu[ni] = p.bb_mean[c]; fis[ni] = p.bb_invstd[c];
    }
  }
  const float blo = p.bb_act == DL3P_ACT_NONE ? -DL3P_INF : 0.f;
  const float bhi = (p.bb_act == DL3P_ACT_NONE || p.bb_act == DL3P_ACT_RELU) ? DL3P_INF : 6.f;
  for (int tile = 0; tile < my_tiles; ++tile) {
    for (int kp = 0; kp < nk; kp += 2) { step(I0{}); step(I1{}); }
    const uint32_t out_m0 = (uint32_t)((int)blockIdx.x + tile * (int)gridDim.x) * S3_BM;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        float zf[16];
        if (BNB) {
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const uint32_t row = out_m0 + mi * 32 + (e & 3) + 8 * (e >> 2);
            zf[e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rF, fv + ni * 128, row * (uint32_t)p.bb_ldz * 4u, 0));
          }
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const float val = acc[mi][ni][e];
          const uint32_t row = out_m0 + mi * 32 + (e & 3) + 8 * (e >> 2);
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(val), rY, yv + ni * 128, row * (uint32_t)ldy * 4u, 0);
          if (BNB) {
            // (rows past M: val is an exact zero and zf a zero from the range check: no contribution)
            const float u = __builtin_fmaf(zf[e], fsc[ni], fsh[ni]);
            const float d = (u > blo && u < bhi) ? val : 0.f;
            const float xh = (zf[e] - fmu[ni]) * fis[ni];
            st_s[ni] += d;
            st_q[ni] = __builtin_fmaf(d, xh, st_q[ni]);
          }
          acc[mi][ni][e] = 0.f;
        }
      }
  }
  if (BNB) {
    float* red = reinterpret_cast<float*>(s3_lds);
    lds_barrier();
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      float s1 = st_s[ni], s2 = st_q[ni];
      s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
      if (hh5 == 0) {
        red[(0 * 2 + wr) * S3_BN + wc * 128 + ni * 32 + l31] = s1;
        red[(1 * 2 + wr) * S3_BN + wc * 128 + ni * 32 + l31] = s2;
      }
    }
    lds_barrier();
    if (p.partials) {
      for (int i = t; i < 2 * S3_BN; i += 256) {
        const int which = i / S3_BN, nn = i - which * S3_BN;
        p.partials[((size_t)blockIdx.x * 2 + which) * p.N + nn] = red[(which * 2 + 0) * S3_BN + nn] + red[(which * 2 + 1) * S3_BN + nn];
      }
    }
  }
}

template <int FACT>
void launch_sb3d_act(const GemmParams& p, bool bnb, int grid, hipStream_t st) {
  const size_t lds = (size_t)S3_LDS_STAGES + 5 * S3D_KP * 4;
  if (bnb) {
    static bool once = (hipFuncSetAttribute((const void*)pw_gemm_sb3d_kernel<FACT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, S3_LDS_MAX), true);
    (void)once;
    dl3p_launch(pw_gemm_sb3d_kernel<FACT, true>, dim3(grid), dim3(256), lds, st, p);
  } else {
    static bool once = (hipFuncSetAttribute((const void*)pw_gemm_sb3d_kernel<FACT, false>, hipFuncAttributeMaxDynamicSharedMemorySize, S3_LDS_MAX), true);
    (void)once;
    dl3p_launch(pw_gemm_sb3d_kernel<FACT, false>, dim3(grid), dim3(256), lds, st, p);
  }
}

template <int ACT, bool PRO>
void launch_sb3_act(const GemmParams& p, bool stats, int grid, hipStream_t st) {
  if (stats) {
    static bool once = (hipFuncSetAttribute((const void*)pw_gemm_sb3_kernel<ACT, PRO, true>, hipFuncAttributeMaxDynamicSharedMemorySize, S3_LDS_MAX), true);
    (void)once;
    dl3p_launch(pw_gemm_sb3_kernel<ACT, PRO, true>, dim3(grid), dim3(256), (size_t)(S3_LDS_STAGES + 2 * p.bsp_pitch * 4), st, p);
  } else {
    static bool once = (hipFuncSetAttribute((const void*)pw_gemm_sb3_kernel<ACT, PRO, false>, hipFuncAttributeMaxDynamicSharedMemorySize, S3_LDS_MAX), true);
    (void)once;
    dl3p_launch(pw_gemm_sb3_kernel<ACT, PRO, false>, dim3(grid), dim3(256), (size_t)(S3_LDS_STAGES + 2 * p.bsp_pitch * 4), st, p);
  }
}
}  // namespace

// role 0 / 1 (forward without / with statistics), (M, K, N) as launched, pitch = row length of the pre-split kernel planes
bool dl3p_sb3_supported(int role, int M, int K, int N, int pitch, int act, bool has_scale, bool accumulate, bool bias) {
  if (role < 0 || role > 1 || N != S3_BN || K % 4 || pitch % 64 || pitch < 256 || pitch < K || pitch > 1024 || M < 2 * S3_BM) return false;      // (pitch <= 1024: the coefficient vectors share the LDS)
  if (!(act == DL3P_ACT_NONE || act == DL3P_ACT_RELU || act == DL3P_ACT_RELU6) || accumulate) return false;
  if (!has_scale && K % 32) return false;                 // (the K tail is zeroed by the out-of-range prologue coefficients)
  if (role == 1 && bias) return false;                    // (padding rows of the last tile are exact zeros only without a bias)
  return true;
}
// the data gradient with the folded BatchNorm-backward apply (dl3p_pwconv_bwd_data_sb_apply): 256 output columns, reduction 256 = pitch
bool dl3p_sb3d_supported(int M, int kout, int nred, int pitch, int f_act, int bb_act, bool bnb, bool accumulate) {
  if (kout != S3_BN || nred != S3D_KP || pitch != S3D_KP || M < 2 * S3_BM || accumulate) return false;
  if (!(f_act == DL3P_ACT_NONE || f_act == DL3P_ACT_RELU || f_act == DL3P_ACT_RELU6)) return false;
  if (bnb && !(bb_act == DL3P_ACT_NONE || bb_act == DL3P_ACT_RELU || bb_act == DL3P_ACT_RELU6)) return false;
  return true;
}
bool dl3p_launch_gemm_sb3d(GemmParams p, bool bnb, int grid, hipStream_t st) {
  p.num_m_tiles = ceil_div(p.M, S3_BM);
#ifdef S3_ONLY_ONE
  return false;
#else
  if (p.f_act == DL3P_ACT_NONE) launch_sb3d_act<DL3P_ACT_NONE>(p, bnb, grid, st);
  else if (p.f_act == DL3P_ACT_RELU) launch_sb3d_act<DL3P_ACT_RELU>(p, bnb, grid, st);
  else if (p.f_act == DL3P_ACT_RELU6) launch_sb3d_act<DL3P_ACT_RELU6>(p, bnb, grid, st);
  else return false;
  return true;
#endif
}
int dl3p_sb3_grid(int M) {
  const int mt = ceil_div(M, S3_BM);
  return mt < DL3P_NUM_CUS ? mt : DL3P_NUM_CUS;
}

bool dl3p_launch_gemm_sb3(GemmParams p, bool stats, int grid, hipStream_t st) {
  p.num_m_tiles = ceil_div(p.M, S3_BM);
  const bool pro = p.scale != nullptr;
#ifdef S3_ONLY_ONE
  // (tests/test_isa_gaps.py compiles ONE instantiation to assembly and counts what sits between its MFMAs)
  if (p.act == DL3P_ACT_RELU && pro) { launch_sb3_act<DL3P_ACT_RELU, true>(p, stats, grid, st); return true; }
  return false;
#else
#define S3_CASE(A) \
  if (p.act == A) { if (pro) launch_sb3_act<A, true>(p, stats, grid, st); else launch_sb3_act<A, false>(p, stats, grid, st); return true; }
  S3_CASE(DL3P_ACT_NONE) S3_CASE(DL3P_ACT_RELU) S3_CASE(DL3P_ACT_RELU6)
#undef S3_CASE
  return false;
#endif
}
