// ROW-STATIONARY form of the fp32-accurate split-bf16 pointwise GEMM (gfx950) for the long, short-reduction layers of the decoder
// (reference call sites: deeplabv3p/models/layers.py:105,209-218 -- `decoder_conv{0,1}_pointwise`, 266256 x 304 -> 256 at
// BASELINE configs[1]; their data gradients 256 -> 304 / 256).  Same arithmetic as pw_split.hip (exact 3-way bf16 split of both
// operands, six products per K-step on v_mfma_f32_16x16x32_bf16, smallest terms first, fp32 accumulation); what differs is the loop
// order and who waits for whom:
//
//   * a wave owns 16 rows of A for the WHOLE reduction (K <= 32 NK): it loads them (coalesced: 4 lanes x 32 B per row), applies the
//     producer's BatchNorm + activation, splits, passes the pieces through a wave-PRIVATE LDS slice into MFMA fragment order and
//     keeps all NK x 3 fragments in registers (120 VGPRs at K = 320).  No workgroup barrier is involved: LDS executes one wave's
//     accesses in order.  Every A element is staged exactly once, whatever the output width.
//   * the pre-split conv kernel streams through LDS in COLUMN GROUPS of CG output channels x the whole K (two buffers, one LDS-only
//     barrier per group = per 6 NK CG/16 MFMAs of every wave, against two barriers per 96 MFMAs in the K-outer kernels); a wave's
//     accumulators are the CG/16 tiles of the current group only and leave for HBM when the group is done.
//   * time is cut into SLOTS (one column group of the B stream each).  A wave spends 2 slots staging a row tile (the VALU-heavy
//     part: prologue + 5.5 instructions per element of split) and the next ceil(N / CG) slots multiplying it against every column
//     group -- any ceil(N / CG) consecutive slots of the cyclic stream hold them all.  The eight waves are two groups of four (one wave
//     of each group per SIMD) whose schedules are SKEWED by half a period, so that on every SIMD the staging of one wave runs under
//     the MFMAs of the other (an MFMA holds the vector issue port for 8 of its 16 cycles; MI355X_MICROARCH.md): the stage ->
//     barrier -> multiply lock-step that bounded pw_gemm_sb_kernel (0.27-0.48 matrix-pipe busy) is gone by construction.
//   * BatchNorm statistics (forward) / BatchNorm-backward sums (data gradient) of a finished 16 x CG tile go through the wave's LDS
//     slice once more: a lane then owns one output column and adds its rows in a fixed order -- deterministic partial rows in the
//     format of the other GEMM kernels ([workgroups][2][N]).
#include "common.h"
#include "pw_gemm.h"
#include "sb_common.h"

template <int NK, int CG>
struct RsCfg {
  static_assert(NK % 2 == 0, "the two staging slots take NK / 2 chunks each");
  static constexpr int NI = CG / 16;
  static constexpr int NMAX = NK <= 8 ? 320 : 256;   // output columns served (LDS: the epilogue coefficients; one statistics register per column group)
  // bytes per B-tile row: + 32 -> 16 consecutive rows x four 16-byte chunks cover the 64 banks once (+ 16 at K = 320, two-way, to fit)
  static constexpr int PITCHB = NK * 64 + ((NK <= 8 || CG == 16) ? 32 : 16);
  static constexpr int BPLANE = CG * PITCHB;
  static constexpr int BTILE = 3 * BPLANE;
  // producer scale / shift (zero-padded to 32 NK) + the epilogue's per-column vectors: bias, or the four of the fused BatchNorm-backward sums
  static constexpr int COEF = (NK <= 8 ? 5 : 2) * NK * 32 * 4 + 4 * NMAX * 4;      // (five per reduction channel for the folded BatchNorm-backward apply, K <= 256)
  static constexpr int APITCH = (2 * BTILE + 8 * 3 * 16 * 96 + COEF <= 160 * 1024) ? 96 : 80;   // bytes per row of the wave-private A slice
  static constexpr int AREG = 3 * 16 * APITCH;       // (also the 16 x 36-float statistics patch: 2304 B)
  static constexpr int COEF_OFF = 2 * BTILE + 8 * AREG;
  static constexpr int LDS_BYTES = COEF_OFF + COEF;
  static constexpr int TPR = 512 / (3 * CG) < NK * 4 ? 512 / (3 * CG) : NK * 4;     // threads per B-tile row
  static constexpr int NBC = (NK * 4 + TPR - 1) / TPR;          // 16-byte chunks of one B tile per thread
  static_assert(LDS_BYTES <= 160 * 1024, "LDS");
  static_assert(AREG >= 16 * 36 * 4, "statistics patch");      // (two of them with the fused BatchNorm-backward sums: checked at the launch)
};

// MODE 0: plain, 1: + BatchNorm statistics of the output, 2: + fused BatchNorm-backward sums (data gradient; GemmParams::bb_*)
// CPS: K chunks staged per staging slot (NK / CPS staging slots per row tile, CPS chunk loads of 32 bytes in flight per lane)
template <int NK, int CG, int MODE, int CPS, bool FOLD = false>
__global__ __launch_bounds__(512, 1) void pw_gemm_sbr_kernel(GemmParams p) {
  using C = RsCfg<NK, CG>;
  static_assert(MODE != 2 || C::AREG >= 2 * 16 * 36 * 4, "two statistics patches");
  constexpr int NI = C::NI, PITCHB = C::PITCHB, BPLANE = C::BPLANE, BTILE = C::BTILE, APITCH = C::APITCH, NBC = C::NBC;
  constexpr bool STATS = MODE >= 1, BNB = MODE == 2;
  constexpr int HR = 64 / CG, RPL = 16 / HR;         // statistics read-back: HR row groups of RPL rows, one column per lane
  extern __shared__ __attribute__((aligned(16))) unsigned char rs_lds[];
  const int t = threadIdx.x, l = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);     // wave-uniform: the slot schedule branches on it
  const int grp = w >> 2, wr = w & 3;                // a workgroup's waves go round the SIMDs: waves w and w + 4 share one
  const int l15 = l & 15, q = l >> 4;
  const int ar = l >> 2, ac = l & 3;                 // staging: row of the wave's 16, 32-byte piece of the 128-byte K chunk
  unsigned char* Areg = rs_lds + 2 * BTILE + w * C::AREG;
  float* coef = reinterpret_cast<float*>(rs_lds + C::COEF_OFF);
  const bool has_pro = p.scale != nullptr || p.act != DL3P_ACT_NONE;
#ifdef DL3P_SB_ABLATE
  const int abl = p.stagger;     // ablation build (scripts/micro/sb_ablate.sh): one ingredient dropped per run, results wrong, time only
#else
  constexpr int abl = 0;
#endif
#ifdef DL3P_SB_ABLATE
  // abl == 100: s_memtime stamps of workgroup 0 (every wave, every slot, four marks) into the buffer p.B points at
  unsigned long long* stamps = (abl == 100 && blockIdx.x == 0) ? reinterpret_cast<unsigned long long*>(const_cast<float*>(p.B)) : nullptr;
#define RS_STAMP(slot, k) do { if (stamps && l == 0 && (slot) < 128) stamps[((size_t)w * 128 + (slot)) * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define RS_STAMP(slot, k) do { } while (0)
#endif
  static_assert(!FOLD || NK <= 8, "LDS for the five folded coefficient vectors");
  for (int i = t; i < NK * 32; i += 512) {
    if (FOLD) {
      // dz = c0 (g m - c1 - (z - mean) invstd c2) = fA g m + fnC z + fD with fA = c0, fnC = -c0 c2 invstd, fD = -fnC mean - c0 c1
      const bool in = i < p.K;
      const int k = min(i, p.K - 1);
      const float c0 = p.f_coef[k], c1 = p.f_coef[p.K + k], c2 = p.f_coef[2 * p.K + k];
      const float nC = -(c0 * p.f_invstd[k] * c2);
      coef[i] = in ? p.f_scale[k] : 1.f;
      coef[NK * 32 + i] = in ? p.f_shift[k] : 0.f;
      coef[2 * NK * 32 + i] = in ? c0 : 0.f;
      coef[3 * NK * 32 + i] = in ? nC : 0.f;
      coef[4 * NK * 32 + i] = in ? -(nC * p.f_mean[k]) - c0 * c1 : 0.f;
    } else {
      const bool in = p.scale && i < p.K;
      coef[i] = in ? p.scale[i] : 1.f;
      coef[NK * 32 + i] = in ? p.shift[i] : 0.f;
    }
  }
  // per-output-column vectors of the epilogue: a global load there is a full memory latency in front of every slot's barrier
  float* ecoef = coef + (NK <= 8 ? 5 : 2) * NK * 32;                 // [4][NMAX]
  for (int i = t; i < C::NMAX; i += 512) {
    const int n = min(i, p.N - 1);
    if (BNB) {
      ecoef[i] = p.bb_scale[n]; ecoef[C::NMAX + i] = p.bb_shift[n]; ecoef[2 * C::NMAX + i] = p.bb_mean[n]; ecoef[3 * C::NMAX + i] = p.bb_invstd[n];
    } else {
      ecoef[i] = p.bias ? p.bias[n] : 0.f;
    }
  }

  const int ncg = (p.N + CG - 1) / CG;
  static_assert(NK % CPS == 0, "whole staging slots");
  constexpr int ASL = NK / CPS;                      // staging slots per row tile
  const int P = ncg + ASL, SKEW = P / 2;
  const int nht = p.num_m_tiles;                     // half tiles of 64 rows (one per wave group and period)
  const int stride = 2 * (int)gridDim.x;
  const int first = 2 * (int)blockIdx.x + grp;
  const int my_nt = first < nht ? (nht - first + stride - 1) / stride : 0;
  int S = 0;                                         // slots of this workgroup (both groups' schedules)
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const int f = 2 * (int)blockIdx.x + g;
    const int n = f < nht ? (nht - f + stride - 1) / stride : 0;
    if (n > 0) S = max(S, g * SKEW + n * P);
  }

  // ---------------------------------------------------------------- the B stream
  // a tile row = (plane, output channel) holds 4 NK chunks of 16 bytes; TPR threads share it, thread j taking chunks j, j + TPR, ...:
  // one base address per thread on either side, the rest are immediate offsets (the pre-split kernel's pitch is exactly 32 NK,
  // its zero padding included -- dl3p_sb_rs_supported)
  constexpr int TPR = C::TPR;
  const char* Bb = reinterpret_cast<const char*>(p.Bsp);
  uint4 rb[NBC];
  // (threads past the last (plane, row) pair repeat pair 3 CG - 1, and a round that runs past the row repeats its last chunk: the
  // same bytes to the same address -- no branch inside the MFMA steps the pieces are interleaved with)
  const int b_pair = min(t / TPR, 3 * CG - 1), b_j = t - (t / TPR) * TPR;        // (plane, row) pair, thread within it
  const int b_plane = b_pair / CG, b_row = b_pair - (b_pair / CG) * CG;
  const uint32_t b_goff = (uint32_t)(((long long)b_plane * p.bsp_plane + b_j * 8) * 2);
  unsigned char* b_dst = rs_lds + b_plane * BPLANE + b_row * PITCHB + b_j * 16;
  const int b_last = (min(b_j + (NBC - 1) * TPR, NK * 4 - 1) - b_j) * 16;      // the last round may run past the row: clamped (not stored)
  // The duty is cut into NBC pieces (store chunk i of tile s + 1, request chunk i of tile s + 2 into the registers just freed) that
  // the slot's own work interleaves with its MFMA steps / staging chunks: done in one go at the head of a slot, all eight waves
  // write 52-63 KB into LDS at once behind the barrier and no MFMA is in flight on any SIMD meanwhile (-59 us of 333 in the ablation)
  const char* b_src = Bb;            // row of the tile being requested (slot s + 2)
  unsigned char* b_put = b_dst;      // this thread's chunk 0 in the buffer being filled (slot s + 1)
  bool b_live = true;                // output channels past N: zero rows
  auto b_slot = [&](int buf, int cg_put, int cg_get) __attribute__((always_inline)) {
    b_put = b_dst + buf * BTILE;
    b_live = cg_put * CG + b_row < p.N;
    b_src = Bb + (b_goff + (uint32_t)min(cg_get * CG + b_row, p.N - 1) * (uint32_t)p.bsp_pitch * 2u);
  };
  auto b_get = [&](auto ic) __attribute__((always_inline)) {
    constexpr int I = decltype(ic)::value;
    rb[I] = *reinterpret_cast<const uint4*>(b_src + (I + 1 < NBC ? I * TPR * 16 : b_last));
  };
  auto b_piece = [&](auto ic) __attribute__((always_inline)) {
    constexpr int I = decltype(ic)::value;
    if constexpr (I < NBC) {
      uint4 v = rb[I];
      if (!b_live) v = make_uint4(0u, 0u, 0u, 0u);
      *reinterpret_cast<uint4*>(b_put + (I + 1 < NBC ? I * TPR * 16 : b_last)) = v;
      b_get(ic);
    }
  };
  auto b_all = [&]() __attribute__((always_inline)) { static_for<NBC>([&](auto ic) { b_piece(ic); }); };

  // ---------------------------------------------------------------- staging a row tile (wave-private)
  s16x8 afr[NK][3];
  constexpr int NH = CPS;           // chunks per staging slot = row-tile loads in flight per lane (NH x 32 bytes)
  float4 ra[NH][2];
  const float act_lo = p.act == DL3P_ACT_NONE ? -DL3P_INF : 0.f;
  const float act_hi = (p.act == DL3P_ACT_NONE || p.act == DL3P_ACT_RELU) ? DL3P_INF : 6.f;
  const char* Ab = reinterpret_cast<const char*>(p.A);
  // one base address per row tile (set when its chunk 0 is requested), chunks at immediate offsets; only the LAST chunk can reach
  // past K and clamps per lane (per-chunk clamps and masks are loop invariants the compiler keeps in registers: 20 per chunk)
  const char* arow = Ab;
  const char* zrow = reinterpret_cast<const char*>(p.f_z);       // FOLD: the second tensor of the staged operand, and where dz goes
  char* dzrow = reinterpret_cast<char*>(p.f_dz);
  float4 rz[NH][2];      // (FOLD only: dead otherwise)
  const float* coef_l = coef + ac * 8;
  auto issue_a = [&](int m0w, auto ktc) __attribute__((always_inline)) {
    constexpr int KT = decltype(ktc)::value;
    if constexpr (KT == 0) {
      const size_t r = (size_t)min(m0w + ar, p.M - 1);
      arow = Ab + (r * (size_t)p.lda * 4u + (size_t)ac * 32u);
      if (FOLD) {
        zrow = reinterpret_cast<const char*>(p.f_z) + (r * (size_t)p.f_ldz * 4u + (size_t)ac * 32u);
        dzrow = reinterpret_cast<char*>(p.f_dz) + (r * (size_t)p.f_lddz * 4u + (size_t)ac * 32u);
      }
    }
    if (FOLD) {
      if constexpr (KT == NK - 1) {
        const int kk = KT * 32 + ac * 8;
        rz[KT % NH][0] = *reinterpret_cast<const float4*>(zrow + (min(kk, p.K - 4) - ac * 8) * 4);
        rz[KT % NH][1] = *reinterpret_cast<const float4*>(zrow + (min(kk + 4, p.K - 4) - ac * 8) * 4);
      } else {
        rz[KT % NH][0] = *reinterpret_cast<const float4*>(zrow + KT * 128);
        rz[KT % NH][1] = *reinterpret_cast<const float4*>(zrow + KT * 128 + 16);
      }
    }
    if constexpr (KT == NK - 1) {
      const int kk = KT * 32 + ac * 8;
      ra[KT % NH][0] = *reinterpret_cast<const float4*>(arow + (min(kk, p.K - 4) - ac * 8) * 4);
      ra[KT % NH][1] = *reinterpret_cast<const float4*>(arow + (min(kk + 4, p.K - 4) - ac * 8) * 4);
    } else {
      ra[KT % NH][0] = *reinterpret_cast<const float4*>(arow + KT * 128);
      ra[KT % NH][1] = *reinterpret_cast<const float4*>(arow + KT * 128 + 16);
    }
  };
  // a row tile's loads: its first NK / 2 chunks are requested one slot ahead (the last multiply slot of the tile before), each of
  // the others when the chunk NK / 2 in front of it has left its registers -- half a tile's K in flight per lane
  auto issue_half0 = [&](int m0w) __attribute__((always_inline)) {
    static_for<NH>([&](auto ktc) { issue_a(m0w, ktc); });
  };
  // A chunk in three parts, so that the slice's write -> read round trip of chunk k is covered by the arithmetic of chunk k + 1
  // (LDS executes one wave's accesses in order: the wave never has to wait between committing a chunk and fetching the one before):
  //   stage(k): prologue + split into registers;  commit(k): three ds_write_b128 into the wave's slice;  fetch(k): three
  //   ds_read_b128 in MFMA fragment order -> afr[k].     stage(0) commit(0) | stage(1) fetch(0) commit(1) | ... | fetch(last)
  uint4 st_h, st_m, st_l;
  auto stage_chunk = [&](int m0w, auto ktc) __attribute__((always_inline)) {
    constexpr int KT = decltype(ktc)::value;
    float4 v[2] = {ra[KT % NH][0], ra[KT % NH][1]};
    const bool rok = m0w + ar < p.M;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (FOLD) {
        // the BatchNorm-backward apply of (g, z), channel k of the reduction: none / ReLU / ReLU6 derivative as a 0 / 1 mask
        const float* c5 = coef_l + KT * 32 + 4 * h;
        const float4 z = rz[KT % NH][h];
        const float4 u = fma4(z, *reinterpret_cast<const float4*>(c5), *reinterpret_cast<const float4*>(c5 + NK * 32));
        const float glo = p.f_act == DL3P_ACT_NONE ? -DL3P_INF : 0.f;
        const float ghi = (p.f_act == DL3P_ACT_NONE || p.f_act == DL3P_ACT_RELU) ? DL3P_INF : 6.f;
        const float4 gm = make_float4((u.x > glo && u.x < ghi) ? v[h].x : 0.f, (u.y > glo && u.y < ghi) ? v[h].y : 0.f,
                                      (u.z > glo && u.z < ghi) ? v[h].z : 0.f, (u.w > glo && u.w < ghi) ? v[h].w : 0.f);
        v[h] = fma4(gm, *reinterpret_cast<const float4*>(c5 + 2 * NK * 32),
                    fma4(z, *reinterpret_cast<const float4*>(c5 + 3 * NK * 32), *reinterpret_cast<const float4*>(c5 + 4 * NK * 32)));
        bool okf = rok;
        if constexpr (KT == NK - 1) okf = okf && KT * 32 + ac * 8 + 4 * h < p.K;
        if (okf) *reinterpret_cast<float4*>(dzrow + KT * 128 + 16 * h) = v[h];      // each element exactly once: in place of g is fine
      } else if (has_pro) {
        if (p.scale) v[h] = fma4(v[h], *reinterpret_cast<const float4*>(coef_l + KT * 32 + 4 * h), *reinterpret_cast<const float4*>(coef_l + NK * 32 + KT * 32 + 4 * h));
        if (p.act >= DL3P_ACT_HSWISH) v[h] = act_apply4(v[h], p.act);
        else v[h] = make_float4(__builtin_amdgcn_fmed3f(v[h].x, act_lo, act_hi), __builtin_amdgcn_fmed3f(v[h].y, act_lo, act_hi),
                                __builtin_amdgcn_fmed3f(v[h].z, act_lo, act_hi), __builtin_amdgcn_fmed3f(v[h].w, act_lo, act_hi));
      }
      // rows past M and the K tail stay exactly zero
      bool ok = rok;
      if constexpr (KT == NK - 1) ok = ok && KT * 32 + ac * 8 + 4 * h < p.K;
      if (!ok) v[h] = zero4();
    }
    split2(v[0].x, v[0].y, st_h.x, st_m.x, st_l.x);
    split2(v[0].z, v[0].w, st_h.y, st_m.y, st_l.y);
    split2(v[1].x, v[1].y, st_h.z, st_m.z, st_l.z);
    split2(v[1].z, v[1].w, st_h.w, st_m.w, st_l.w);
  };
  auto commit_chunk = [&]() __attribute__((always_inline)) {
    unsigned char* d = Areg + ar * APITCH + ac * 16;
    __builtin_amdgcn_wave_barrier();
    *reinterpret_cast<uint4*>(d) = st_h;
    *reinterpret_cast<uint4*>(d + 16 * APITCH) = st_m;
    *reinterpret_cast<uint4*>(d + 32 * APITCH) = st_l;
    __builtin_amdgcn_wave_barrier();
  };
  auto fetch_chunk = [&](auto ktc) __attribute__((always_inline)) {
    constexpr int KT = decltype(ktc)::value;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) afr[KT][pl] = *reinterpret_cast<const s16x8*>(Areg + pl * 16 * APITCH + l15 * APITCH + q * 16);
    __builtin_amdgcn_wave_barrier();
  };

  // ---------------------------------------------------------------- statistics
  constexpr int NST = C::NMAX / CG;      // column groups = statistics registers per lane
  float stat[NST];
#pragma unroll
  for (int i = 0; i < NST; ++i) stat[i] = 0.f;
  float* patch = reinterpret_cast<float*>(Areg);         // [16 rows][36]
  const int sc_ = l & (CG - 1), sh_ = l / CG;           // read-back lane: column, row group
  // sum over the 16 rows of the patch column this lane owns (fixed order), both row-group halves combined
  auto patch_colsum = [&](const float* pt, bool squares, float& s1, float& s2) __attribute__((always_inline)) {
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int r = 0; r < RPL; ++r) {
      const float v = pt[(sh_ * RPL + r) * 36 + sc_];
      a += v;
      if (squares) b = fmaf(v, v, b);
    }
#pragma unroll
    for (int o = CG; o < 64; o <<= 1) { a += __shfl_xor(a, o); if (squares) b += __shfl_xor(b, o); }
    s1 = a; s2 = b;
  };
  auto stat_add = [&](int cg, float v) __attribute__((always_inline)) {
    static_for<NST>([&](auto jc) { if (cg == decltype(jc)::value) stat[decltype(jc)::value] += v; });
  };
  int stat_cg = -1;                 // column group whose patch is waiting (wave-uniform)
  auto lazy_stats = [&]() __attribute__((always_inline)) {
    if (stat_cg >= 0) {
      float s1, s2;
      __builtin_amdgcn_wave_barrier();
      patch_colsum(patch, !BNB, s1, s2);
      if (BNB) { float t2; patch_colsum(patch + 16 * 36, false, s2, t2); }
      __builtin_amdgcn_wave_barrier();
      stat_add(stat_cg, sh_ == 0 ? s1 : s2);        // row-group-0 lanes keep the first sum, row-group-1 lanes the second
      stat_cg = -1;
    }
  };

  int s = 0;                                         // slot counter (same in every wave)
  // ---------------------------------------------------------------- one column group of a staged row tile
  auto mfma_slot = [&](int buf, int cg, int m0w, int m0w_next) __attribute__((always_inline)) {      // m0w_next >= 0: request that row tile's first half
    f32x4 acc[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int m = m0w + l15;
    const bool rok = m < p.M;
    const int mc = min(m, p.M - 1);
    // (the z tile of the fused BatchNorm-backward sums is requested at the head of the slot: behind the pieces its latency stands in
    // front of the epilogue -- 374 against 362 us on 266256 x 256 -> 304)
    float4 zv[BNB ? NI : 1];
    if (BNB) {
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
        zv[ni] = ld4(p.bb_z + (size_t)mc * p.bb_ldz + min(cg * CG + ni * 16 + q * 4, p.N - 4));
    }
    const unsigned char* Bt = rs_lds + buf * BTILE + l15 * PITCHB + q * 16;
    // NK x NI steps of six MFMAs; the B fragments of step + 1 are requested in front of the MFMAs of step (two register sets) and a
    // scheduling barrier closes every step: left alone, the scheduler hoists the fragment reads of the whole slot to its top
    // (24 registers per K chunk -- 150 spilled at K = 320)
    constexpr int WBD = (FOLD && BNB) ? 1 : 2;        // fragment sets (one where the registers are needed for the second staged tensor)
    s16x8 wb[WBD][3];
    auto read_b = [&](auto stc) __attribute__((always_inline)) {
      constexpr int ST = decltype(stc)::value, KT = ST / NI, ni = ST % NI;
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) wb[ST % WBD][pl] = *reinterpret_cast<const s16x8*>(Bt + pl * BPLANE + ni * 16 * PITCHB + KT * 64);
    };
    RS_STAMP(s, 1);
    if (abl != 6 && WBD == 2) read_b(std::integral_constant<int, 0>{});
    if (abl != 6) static_for<NK * NI>([&](auto stc) {
      constexpr int ST = decltype(stc)::value, KT = ST / NI, ni = ST % NI, S2 = ST % WBD;
      if constexpr (WBD == 1) read_b(stc);
      else if constexpr (ST + 1 < NK * NI) read_b(std::integral_constant<int, ST + 1>{});
      if (abl != 4) b_piece(stc);          // (pieces past NBC are empty)
      // smallest terms first; MFMA "A" = weights (row = output channel), "B" = activations (column = pixel): a lane ends with 4
      // consecutive output channels of one pixel
      f32x4 c = acc[ni];
      if (abl == 2) { c[0] += __builtin_bit_cast(float, (int)wb[S2][0][0] + (int)wb[S2][1][1] + (int)wb[S2][2][2]); acc[ni] = c; __builtin_amdgcn_sched_barrier(0); return; }
      acc[ni] = split_mac(c, wb[S2][0], wb[S2][1], wb[S2][2], afr[KT][0], afr[KT][1], afr[KT][2]);
      // inside the step: the next step's fragment reads FIRST (left alone they sink behind the fourth MFMA and the step after
      // waits a whole LDS round trip for them), then this step's piece of the B stream, then the six MFMAs
      if constexpr (WBD == 1 || ST + 1 < NK * NI) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
      if constexpr (ST < NBC) { __builtin_amdgcn_sched_group_barrier(0x200, 1, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); }
      __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (ST == 1 && STATS) { lazy_stats(); __builtin_amdgcn_sched_barrier(0); }
      if constexpr (ST == (NBC < NK * NI ? NBC : NK * NI - 1)) {
        // the next row tile's first half is requested BEHIND the slot's last B piece (see "order of the memory requests" below)
        if (m0w_next >= 0) issue_half0(m0w_next);
        __builtin_amdgcn_sched_barrier(0);
      }
    });
    RS_STAMP(s, 2);
    // epilogue: lane (l15 = pixel, q) holds output channels cg CG + 16 ni + 4 q .. + 3
    float4 o[NI], e2[BNB ? NI : 1];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int n = cg * CG + ni * 16 + q * 4;
      const bool ok = rok && n < p.N;
      const int nc = min(n, p.N - 4);
      o[ni] = make_float4(acc[ni][0], acc[ni][1], acc[ni][2], acc[ni][3]);
      if (!BNB && p.bias) o[ni] = add4(o[ni], *reinterpret_cast<const float4*>(ecoef + nc));
      float* yp = p.Y + (size_t)mc * p.ldy + nc;
      if (p.accumulate) o[ni] = add4(o[ni], ld4(yp));
      if (ok && abl != 5) st4(yp, o[ni]);
      if (BNB) {
        const float4 bsc = *reinterpret_cast<const float4*>(ecoef + nc), bsh = *reinterpret_cast<const float4*>(ecoef + C::NMAX + nc);
        const float4 bmu = *reinterpret_cast<const float4*>(ecoef + 2 * C::NMAX + nc), bis = *reinterpret_cast<const float4*>(ecoef + 3 * C::NMAX + nc);
        const float4 z = zv[ni];
        const float4 u = fma4(z, bsc, bsh);
        float4 d;
        if (p.bb_act >= DL3P_ACT_HSWISH) {
          d = make_float4(o[ni].x * act_grad(u.x, p.bb_act), o[ni].y * act_grad(u.y, p.bb_act),
                          o[ni].z * act_grad(u.z, p.bb_act), o[ni].w * act_grad(u.w, p.bb_act));
        } else {      // none / ReLU / ReLU6: the derivative is a 0 / 1 mask (two compares and a select instead of act_grad's ten instructions)
          const float glo = p.bb_act == DL3P_ACT_NONE ? -DL3P_INF : 0.f;
          const float ghi = (p.bb_act == DL3P_ACT_NONE || p.bb_act == DL3P_ACT_RELU) ? DL3P_INF : 6.f;
          d = make_float4((u.x > glo && u.x < ghi) ? o[ni].x : 0.f, (u.y > glo && u.y < ghi) ? o[ni].y : 0.f,
                          (u.z > glo && u.z < ghi) ? o[ni].z : 0.f, (u.w > glo && u.w < ghi) ? o[ni].w : 0.f);
        }
        const float4 xh = make_float4((z.x - bmu.x) * bis.x, (z.y - bmu.y) * bis.y, (z.z - bmu.z) * bis.z, (z.w - bmu.w) * bis.w);
        o[ni] = d;
        e2[ni] = mul4(d, xh);
        if (!ok) e2[ni] = zero4();
      }
      if (STATS && !ok) o[ni] = zero4();
    }
    if (STATS && abl != 1 && abl != 5) {
      // the tile's values (and, BNB, its second factor) go to the wave's patch; the column sums are taken LAZILY -- by lazy_stats()
      // behind the first MFMA steps of the wave's next slot (or at the head of its next staging slot), where the write -> read
      // round trip is long over and the reads hide behind queued MFMAs instead of standing in front of the slot's barrier
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) *reinterpret_cast<float4*>(patch + l15 * 36 + ni * 16 + q * 4) = o[ni];
      if (BNB) {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) *reinterpret_cast<float4*>(patch + 16 * 36 + l15 * 36 + ni * 16 + q * 4) = e2[ni];
      }
      __builtin_amdgcn_wave_barrier();
      stat_cg = cg;
    }
  };

  // ---------------------------------------------------------------- the slot loop
  __syncthreads();          // coef
  if (S > 0) {
    b_slot(0, 0, 0);
    static_for<NBC>([&](auto ic) { b_get(ic); });
    b_slot(0, 0, 1 % ncg);
    b_all();              // tile 0 into buffer 0, tile 1 requested
  }
  // Structured per wave group (not a state machine over slots: with one, every fragment register travels round the slot loop through
  // the merge points of all three phases and the allocator doubles them): idle slots in front (the skew), then per row tile two
  // staging slots and ncg multiply slots, idle slots behind; every slot = B duty at its head, one barrier at its tail, S in all.
  int cg0 = 0, cg1 = 1 % ncg, cg2 = 2 % ncg;         // column group of slots s, s + 1, s + 2
  auto slot_head = [&]() __attribute__((always_inline)) { RS_STAMP(s, 0); b_slot((s + 1) & 1, cg1, cg2); };
  auto slot_tail = [&]() __attribute__((always_inline)) {
    RS_STAMP(s, 3);
    ++s;
    cg0 = cg1; cg1 = cg2; cg2 = cg2 + 1 == ncg ? 0 : cg2 + 1;
    lds_barrier();
  };
  const int lead = my_nt > 0 ? SKEW * grp : S;      // idle slots in front
  if (my_nt > 0 && lead == 0) issue_half0(first * 64 + wr * 16);
  lds_barrier();
  for (int i = 0; i < lead; ++i) {
    slot_head();
    if (abl != 4) b_all();
    if (i == lead - 1 && my_nt > 0) issue_half0(first * 64 + wr * 16);
    slot_tail();
  }
  for (int j = 0; j < my_nt; ++j) {
    const int m0w = (first + j * stride) * 64 + wr * 16;
    const bool do_a = !(abl == 3 && j > 0);
    // ORDER OF THE MEMORY REQUESTS.  Loads return in issue order and the compiler's s_waitcnt for a B register counts conservatively
    // (it cannot know what the slot before issued: vmcnt(6) at every piece), so a B piece behind a row-tile request waits out that
    // request's whole HBM latency (in-kernel stamps: 1.2-4.6 k cycles per staged chunk, the multiply loop of the slot that prefetches
    // the next tile twice as long).  Hence, in every slot: ALL B pieces first, row-tile requests after them.
    static_for<ASL>([&](auto slc) {
      constexpr int SL = decltype(slc)::value;
      slot_head();
      if constexpr (SL == 0) { if (STATS) lazy_stats(); }      // (the patch lives in the slice the chunks are about to overwrite)
      if (abl != 4) b_all();
      RS_STAMP(s, 1);
      static_for<CPS>([&](auto ic) {
        constexpr int I = decltype(ic)::value, KT = SL * CPS + I;
        if (do_a) stage_chunk(m0w, std::integral_constant<int, KT>{});
        if constexpr (KT + CPS < NK) { if (!(abl == 7 && j > 0)) issue_a(m0w, std::integral_constant<int, KT + CPS>{}); }
        if constexpr (I > 0) { if (do_a) fetch_chunk(std::integral_constant<int, KT - 1>{}); }
        if (do_a) commit_chunk();
        RS_STAMP(s, 4 + (I & 3));
        __builtin_amdgcn_sched_barrier(0);      // one chunk at a time: the scheduler would otherwise start every chunk's loads first (33 registers per chunk)
      });
      if (do_a) fetch_chunk(std::integral_constant<int, SL * CPS + CPS - 1>{});
      slot_tail();
    });
    for (int c = 0; c < ncg; ++c) {
      slot_head();
      mfma_slot((s & 1), cg0, m0w, (c == ncg - 1 && j + 1 < my_nt && abl != 7) ? m0w + stride * 64 : -1);
      slot_tail();
    }
  }
  while (s < S) {
    slot_head();
    if (abl != 4) b_all();
    slot_tail();
  }

  if (STATS) {
    lazy_stats();
    // lane (column sc_, row group sh_ in {0, 1}) holds, per column group j, sum (sh_ = 0) / second sum (sh_ = 1) of column j CG + sc_
    float* red = reinterpret_cast<float*>(rs_lds);        // [2][8 waves][N], over the B buffers (every wave is past its last read)
    if (sh_ < 2) {
      static_for<NST>([&](auto jc) {
        constexpr int J = decltype(jc)::value;
        const int col = J * CG + sc_;
        if (J < ncg && col < p.N) red[(sh_ * 8 + w) * p.N + col] = stat[J];
      });
    }
    lds_barrier();
    if (p.partials) {
      for (int i = t; i < 2 * p.N; i += 512) {
        const int which = i / p.N, nn = i - which * p.N;
        float s = 0.f;
#pragma unroll
        for (int ww = 0; ww < 8; ++ww) s += red[(which * 8 + ww) * p.N + nn];
        p.partials[((size_t)blockIdx.x * 2 + which) * p.N + nn] = s;
      }
    }
  }
}

// ------------------------------------------------------------------------------ host side
template <int NK, int CG, int MODE, int CPS = 2, bool FOLD = false>
static void launch_sbr_one(const GemmParams& p, int grid, hipStream_t st) {
  using C = RsCfg<NK, CG>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)pw_gemm_sbr_kernel<NK, CG, MODE, CPS, FOLD>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
    attr_set = true;
  }
  dl3p_launch(pw_gemm_sbr_kernel<NK, CG, MODE, CPS, FOLD>, dim3(grid), dim3(512), (size_t)C::LDS_BYTES, st, p);
}

// -> false when no instantiation serves (reduction length, mode, fold): the caller then fails with DL3P_EINVAL instead of returning
// success with the output untouched (the *_supported predicates keep these combinations unreachable today)
template <int MODE>
static bool launch_sbr_nk(const GemmParams& p, int grid, hipStream_t st) {
  const int nk = (p.K + 31) / 32;
  if (nk == 4) launch_sbr_one<4, 32, MODE>(p, grid, st);
  else if (nk == 6) launch_sbr_one<6, 32, MODE>(p, grid, st);
  else if (nk == 8 && p.f_z) {      // the folded BatchNorm-backward apply (data gradients only): two chunks per staging slot (registers)
    if constexpr (MODE != 1) launch_sbr_one<8, 32, MODE, 2, true>(p, grid, st);
    else return false;
  } else if (nk == 8) {
    // K = 256: four chunks per staging slot (244 / 288 / 218 us forward / data gradient + sums / plain on 266256 rows against 255 / 311 /
    // 233 with two); K = 320: two -- five spill (335 against 372 us)
    static const int cps = getenv("DL3P_SB_RS_CPS") ? atoi(getenv("DL3P_SB_RS_CPS")) : 4;
    if (cps == 4) launch_sbr_one<8, 32, MODE, 4>(p, grid, st);
    else launch_sbr_one<8, 32, MODE, 2>(p, grid, st);
  }
  else if (nk == 10) {
    if constexpr (MODE != 2) {      // (K = 320 leaves no room for the second statistics patch)
      static const int cg16 = getenv("DL3P_SB_RS_CG16") ? atoi(getenv("DL3P_SB_RS_CG16")) : 0;
      if (cg16) launch_sbr_one<10, 16, MODE>(p, grid, st);
      else launch_sbr_one<10, 32, MODE>(p, grid, st);
    } else return false;
  } else return false;
  return true;
}

// shapes served: reduction up to 320 (the A fragments of a row tile live in registers), up to 320 output columns (10 statistics
// registers per lane), 2 N floats of statistics scratch per wave inside the B buffers
bool dl3p_sb_rs_supported(int role, int M, int K, int N) {
  const int nk = (K + 31) / 32;
  if (role == 3 && nk > 8) return false;        // instantiated reduction lengths: 32 nk (the pre-split kernel's pitch, its zero padding included)
  return M >= 2048 && (nk == 4 || nk == 6 || nk == 8 || nk == 10) && K % 4 == 0 && N >= 16 && N <= (nk <= 8 ? 320 : 256) && N % 4 == 0;
}

// workgroups for M rows: one per CU, each wave group at least one half tile
int dl3p_sb_rs_grid(int M) {
  const int nht = (M + 63) / 64;
  int g = (nht + 1) / 2;
  if (g > DL3P_NUM_CUS) g = DL3P_NUM_CUS;
  return g;
}

// p.num_m_tiles = half tiles of 64 rows; mode 0 plain, 1 statistics, 2 fused BatchNorm-backward sums
bool dl3p_launch_gemm_sbr(const GemmParams& p, int mode, int grid, hipStream_t st) {
  if (mode == 2) return launch_sbr_nk<2>(p, grid, st);
  if (mode == 1) return launch_sbr_nk<1>(p, grid, st);
  return launch_sbr_nk<0>(p, grid, st);
}

// the BatchNorm-backward apply folded into the staged operand of the data gradient: reductions of 225 .. 256 (the five per-channel
// coefficient vectors in LDS), none / ReLU / ReLU6
bool dl3p_sb_rs_fold_supported(int M, int K, int N, int act) {
  return dl3p_sb_rs_supported(2, M, K, N) && (K + 31) / 32 == 8 && (act == DL3P_ACT_NONE || act == DL3P_ACT_RELU || act == DL3P_ACT_RELU6);
}
