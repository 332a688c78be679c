// The RGB stem convolution (3x3, stride 2, 3 -> 16 / 32 channels; reference: deeplabv3p/models/deeplabv3p_mobilenetv2.py
// `Conv2D(first_block_filters, kernel_size=3, strides=(2, 2), padding='same', use_bias=False, name='Conv')`, the same
// layer in deeplabv3p_mobilenetv3.py and `entry_flow_conv1_1` of deeplabv3p_xception.py) as an implicit GEMM on the fp32
// matrix cores.
//
// The im2col route writes the [N*Ho*Wo][28] patch matrix to HBM (118 MB at batch 16, 513 x 513) and reads it back twice
// (forward GEMM, weight gradient).  Here a wave stages the three input rows its 64 output pixels touch ((2*64+1) x 3
// floats each) in LDS -- every input pixel is fetched from HBM/L2 once per output row instead of 2.25 times per patch --
// and both GEMM roles read their patch operand from that tile:
//   forward  D[co][pixel] = sum_k W[k][co] * patch[k][pixel]      7 v_mfma_f32_16x16x4_f32 per 16 pixels x 16 channels
//   wgrad    D[k][co]     = sum_pixel patch[k][pixel] * dy[pixel][co]
// k = (ky*3 + kx)*3 + ci, so inside one kernel row ky the 9 values (kx, ci) of a pixel are CONTIGUOUS in the staged
// input row: patch[k][pixel] = row[ky][6*pixel + (k - 9*ky)] -- no index arithmetic in the inner loop.
// The LDS tiles are wave-private (LDS executes one wave's instructions in order), so the tile loop has no workgroup barrier.
#include "common.h"

#define STEM_TP 64                       // output pixels of one wave tile (a segment of one output row)
#define STEM_ROW ((2 * STEM_TP + 1) * 3) // staged floats per input row
#define STEM_PP 392                      // LDS pitch of a staged row

struct StemParams {
  const float* x; int ldx;
  const float* w;            // [28][Cout] (row 27 is padding)
  float* y; int ldy;
  float* partials;           // forward: [grid][2][Cout] BatchNorm statistic rows
  const float* dy; int lddy;
  float* slabs;              // weight gradient: [grid][28][Cout]
  // weight gradient with the BatchNorm-backward apply of the conv's own BatchNorm folded in (FOLD instantiation): dy is the gradient g
  // of act(BN(z)); dz = c0 (g act'(z scale + shift) - c1 - xhat c2) is formed while the gradient rows are staged (as A g act' - C z + D)
  const float* f_z; int f_ldz; const float* f_scale; const float* f_shift; const float* f_mean; const float* f_invstd;
  const float* f_coef; int f_act;
  int N, H, W, Ho, Wo, Cout, pad_t, pad_l, segs, tiles;
};

// stage the 3 input rows of tile (n, oy, ox0 ..) into this wave's LDS tile; out-of-image taps are zero
#define STEM_IT ((STEM_ROW + 63) / 64)
// The staging of a tile in two halves, so that the loads of the NEXT tile are in flight while this one is multiplied (round 5: a tile
// was load -> wait -> LDS -> MFMA; with two waves per SIMD -- the LDS tiles allow no more -- the launch sat at 3.3 TB/s).
// stem_fetch: the three input rows of tile (n, oy, ox0 ..) into registers.  Branch-free: every load is issued from a clamped address and
// out-of-image taps are selected to zero in stem_put.  (A predicated load per tap -- which is also what the compiler makes of
// `cond ? load : 0` by sinking the load into the branch -- waits for each load before the next branch: 21 serialized memory latencies
// per tile, 87 us per launch.)
// (fp32 MFMAs do not overlap vector instructions on this chip -- DESIGN 8, valu_issue.hip -- so the index arithmetic of the staging
// competes with the multiply: tiles whose three rows and 129 columns all lie inside the image -- three of five per row at 513 x 513 --
// take a path without clamps and selects: one add per load.)
__device__ __forceinline__ bool stem_interior(const StemParams& p, int oy, int ox0) {
  const int ix0 = 2 * ox0 - p.pad_l, iy0 = 2 * oy - p.pad_t;
  return ix0 >= 0 && ix0 + 2 * STEM_TP < p.W && iy0 >= 0 && iy0 + 2 < p.H;
}
__device__ __forceinline__ void stem_fetch(const StemParams& p, int n, int oy, int ox0, int l, float (&v)[3][STEM_IT]) {
  const int ix0 = 2 * ox0 - p.pad_l;
  if (stem_interior(p, oy, ox0) && p.ldx == 3) {          // (wave-uniform)
    const float* src = p.x + (((size_t)n * p.H + (2 * oy - p.pad_t)) * p.W + ix0) * 3 + l;
    const size_t rowf = (size_t)p.W * 3;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int it = 0; it < STEM_IT; ++it) v[ky][it] = src[ky * rowf + (64 * it < STEM_ROW - 63 ? 64 * it : STEM_ROW - 64)];
    return;
  }
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int iy = 2 * oy - p.pad_t + ky;
    const bool rowok = iy >= 0 && iy < p.H;
    const float* src = p.x + ((size_t)n * p.H + (rowok ? iy : 0)) * p.W * p.ldx;
#pragma unroll
    for (int it = 0; it < STEM_IT; ++it) {
      const int i = l + 64 * it;
      const int ic = i < STEM_ROW ? i : STEM_ROW - 1;
      const int px = ic / 3, ci = ic - 3 * px;
      const int ix = ix0 + px;
      const int ixc = ix < 0 ? 0 : (ix < p.W ? ix : p.W - 1);
      v[ky][it] = src[(size_t)ixc * p.ldx + ci];
    }
  }
}
// stem_put: what stem_fetch requested for tile (n, oy, ox0 ..) into this wave's LDS tile; out-of-image taps are zero
__device__ __forceinline__ void stem_put(const StemParams& p, float* tile, int oy, int ox0, int l, float (&v)[3][STEM_IT]) {
  const int ix0 = 2 * ox0 - p.pad_l;
  if (stem_interior(p, oy, ox0) && p.ldx == 3) {
    // the last pass of 64 lanes was fetched from element STEM_ROW - 64 + l (inside the row): it lands there
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int it = 0; it < STEM_IT; ++it) tile[ky * STEM_PP + (64 * it < STEM_ROW - 63 ? 64 * it : STEM_ROW - 64) + l] = v[ky][it];
    return;
  }
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int iy = 2 * oy - p.pad_t + ky;
    const bool rowok = iy >= 0 && iy < p.H;
#pragma unroll
    for (int it = 0; it < STEM_IT; ++it) {
      const int i = l + 64 * it;
      const int ic = i < STEM_ROW ? i : STEM_ROW - 1;
      const int ix = ix0 + ic / 3;
      asm volatile("" : "+v"(v[ky][it]));  // keeps the load in stem_fetch unconditional
      const float x = (rowok && ix >= 0 && ix < p.W) ? v[ky][it] : 0.f;
      if (i < STEM_ROW) tile[ky * STEM_PP + i] = x;
    }
  }
}

__device__ __forceinline__ void stem_tile_coords(const StemParams& p, int tile, int* n, int* oy, int* ox0) {
  const int row = tile / p.segs;
  *ox0 = (tile - row * p.segs) * STEM_TP;
  *n = row / p.Ho;
  *oy = row - *n * p.Ho;
}

template <int NI>
__global__ __launch_bounds__(256) void stem_fwd_kernel(StemParams p) {
  __shared__ float patch[4][3 * STEM_PP];
  __shared__ float red[2][4][NI * 16];
  const int t = threadIdx.x, wv = t >> 6, l = t & 63, r = l & 15, kq = l >> 4;
  float wf[NI][7];
  int koff[7];
#pragma unroll
  for (int j = 0; j < 7; ++j) {
    const int k = 4 * j + kq;
    const int kk = k < 27 ? k : 26;
    const int ky = kk / 9;
    koff[j] = ky * STEM_PP + (kk - 9 * ky) + 6 * r;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) wf[ni][j] = k < 27 ? p.w[(size_t)k * p.Cout + ni * 16 + r] : 0.f;
  }
  float4 s1[NI], s2[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) { s1[ni] = zero4(); s2[ni] = zero4(); }
  float* tile = patch[wv];
  const XcdRange rg = xcd_range(p.tiles, blockIdx.x, gridDim.x, 4, wv);
  float xv3[3][STEM_IT];
  if (rg.begin < rg.end) {
    int n, oy, ox0;
    stem_tile_coords(p, rg.begin, &n, &oy, &ox0);
    stem_fetch(p, n, oy, ox0, l, xv3);
  }
  for (int ti = rg.begin; ti < rg.end; ti += rg.step) {
    int n, oy, ox0;
    stem_tile_coords(p, ti, &n, &oy, &ox0);
    __builtin_amdgcn_wave_barrier();
    stem_put(p, tile, oy, ox0, l, xv3);
    {
      // the next tile's rows: in flight while this one is multiplied (the last tile requests itself again: no branch around loads)
      int n2, oy2, ox2;
      stem_tile_coords(p, ti + rg.step < rg.end ? ti + rg.step : ti, &n2, &oy2, &ox2);
      stem_fetch(p, n2, oy2, ox2, l, xv3);
    }
    __builtin_amdgcn_wave_barrier();
    typedef float f4v __attribute__((ext_vector_type(4)));
    f4v acc[4][NI];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) acc[g][ni] = (f4v){0.f, 0.f, 0.f, 0.f};
    const int ng = (p.Wo - ox0 + 15) >> 4;  // 16-pixel groups of this tile that lie in the row (the last segment is short)
#pragma unroll
    for (int g = 0; g < 4; ++g)
      if (g < ng)
#pragma unroll
      for (int j = 0; j < 7; ++j) {
        const float xv = tile[koff[j] + g * 96];
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[g][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[ni][j], xv, acc[g][ni], 0, 0, 0);
      }
    const size_t m0 = ((size_t)n * p.Ho + oy) * p.Wo;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int ox = ox0 + g * 16 + r;
      if (ox < p.Wo) {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          const float4 o = make_float4(acc[g][ni][0], acc[g][ni][1], acc[g][ni][2], acc[g][ni][3]);
#ifdef DL3P_ABLATE_STORES
          if (o.x == 1234.5678f)
#endif
          st4(p.y + (m0 + ox) * p.ldy + ni * 16 + 4 * kq, o);
          s1[ni] = add4(s1[ni], o);
          s2[ni] = fma4(o, o, s2[ni]);
        }
      }
    }
  }
  if (!p.partials) return;
  // per-channel (sum, sum^2): over the 16 pixel lanes, then over the 4 waves
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    float a[8] = {s1[ni].x, s1[ni].y, s1[ni].z, s1[ni].w, s2[ni].x, s2[ni].y, s2[ni].z, s2[ni].w};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float v = a[e];
      v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
      if (r == 0) red[e >> 2][wv][ni * 16 + 4 * kq + (e & 3)] = v;
    }
  }
  __syncthreads();
  if (t < 2 * NI * 16) {
    const int which = t / (NI * 16), c = t - which * NI * 16;
    p.partials[((size_t)blockIdx.x * 2 + which) * p.Cout + c] = red[which][0][c] + red[which][1][c] + red[which][2][c] + red[which][3][c];
  }
}

template <int NI, bool FOLD = false>
__global__ __launch_bounds__(256) void stem_wgrad_kernel(StemParams p) {
  constexpr int CP = NI * 16 + 8;  // pitch of the staged gradient rows: the four pixel quarters land on disjoint bank octets
  __shared__ float patch[4][3 * STEM_PP];
  __shared__ float dys[4][STEM_TP * CP];
  const int t = threadIdx.x, wv = t >> 6, l = t & 63, r = l & 15, kq = l >> 4;
  int koff[2];
  float kmask[2];
#pragma unroll
  for (int kt = 0; kt < 2; ++kt) {
    const int k = kt * 16 + r;
    const int kk = k < 27 ? k : 26;
    const int ky = kk / 9;
    koff[kt] = ky * STEM_PP + (kk - 9 * ky) + 6 * kq;
    kmask[kt] = k < 27 ? 1.f : 0.f;
  }
  typedef float f4v __attribute__((ext_vector_type(4)));
  f4v acc[2][NI];
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[kt][ni] = (f4v){0.f, 0.f, 0.f, 0.f};
  float* tile = patch[wv];
  float* dt = dys[wv];
  // FOLD: the lane's four channels are the same in every staging pass (64 % C4 == 0): their coefficients once
  float4 fA = zero4(), fC = zero4(), fD = zero4(), fsc = make_float4(1.f, 1.f, 1.f, 1.f), fsh = zero4();
  if (FOLD) {
    const int c = 4 * (l % (NI * 4));
    if (p.f_scale) { fsc = ld4(p.f_scale + c); fsh = ld4(p.f_shift + c); }
    const float4 mu = ld4(p.f_mean + c), is = ld4(p.f_invstd + c);
    const float4 c0 = ld4(p.f_coef + c), c1 = ld4(p.f_coef + p.Cout + c), c2 = ld4(p.f_coef + 2 * p.Cout + c);
    fA = c0;
    fC = mul4(mul4(c0, is), c2);
    fD = make_float4(fC.x * mu.x - c0.x * c1.x, fC.y * mu.y - c0.y * c1.y, fC.z * mu.z - c0.z * c1.z, fC.w * mu.w - c0.w * c1.w);
  }
  const XcdRange rg = xcd_range(p.tiles, blockIdx.x, gridDim.x, 4, wv);
  constexpr int C4 = NI * 4;  // 16-byte vectors per row
  constexpr int IT = STEM_TP * C4 / 64;
  float xv3[3][STEM_IT];
  float4 g[IT];
  float4 zz[FOLD ? IT : 1];
  // the input rows and the 64 gradient rows of a tile: requested one tile ahead, in flight while the current tile is multiplied
  auto fetch = [&](int tile_index) __attribute__((always_inline)) {
    int n, oy, ox0;
    stem_tile_coords(p, tile_index, &n, &oy, &ox0);
    stem_fetch(p, n, oy, ox0, l, xv3);
    const size_t m0 = ((size_t)n * p.Ho + oy) * p.Wo + ox0;
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int i = l + 64 * it;
      const int px = i / C4, c4 = i - px * C4;
      const int last = p.Wo - 1 - ox0;  // >= 0: a tile starts inside the row
      g[it] = ld4(p.dy + (m0 + (px < last ? px : last)) * p.lddy + 4 * c4);
      if (FOLD) zz[it] = ld4(p.f_z + (m0 + (px < last ? px : last)) * p.f_ldz + 4 * c4);
    }
  };
  if (rg.begin < rg.end) fetch(rg.begin);
  for (int ti = rg.begin; ti < rg.end; ti += rg.step) {
    int n, oy, ox0;
    stem_tile_coords(p, ti, &n, &oy, &ox0);
    __builtin_amdgcn_wave_barrier();
    stem_put(p, tile, oy, ox0, l, xv3);
    if (FOLD) {
      const int fact = p.f_act;
#pragma unroll
      for (int it = 0; it < IT; ++it) {
        const float4 z = zz[it], gg = g[it];
        const float4 u = fma4(z, fsc, fsh);
        g[it] = make_float4(fmaf(fA.x, gg.x * act_grad(u.x, fact), fmaf(-fC.x, z.x, fD.x)),
                            fmaf(fA.y, gg.y * act_grad(u.y, fact), fmaf(-fC.y, z.y, fD.y)),
                            fmaf(fA.z, gg.z * act_grad(u.z, fact), fmaf(-fC.z, z.z, fD.z)),
                            fmaf(fA.w, gg.w * act_grad(u.w, fact), fmaf(-fC.w, z.w, fD.w)));
      }
    }
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int px = (l + 64 * it) / C4;
      asm volatile("" : "+v"(g[it].x), "+v"(g[it].y), "+v"(g[it].z), "+v"(g[it].w));
      if (px > p.Wo - 1 - ox0) g[it] = zero4();
    }
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int i = l + 64 * it;
      const int px = i / C4, c4 = i - px * C4;
      st4(dt + px * CP + 4 * c4, g[it]);
    }
    fetch(ti + rg.step < rg.end ? ti + rg.step : ti);      // (the last tile requests itself again: no branch around loads)
    __builtin_amdgcn_wave_barrier();
    const int ns = (p.Wo - ox0 + 3) >> 2;  // 4-pixel steps of this tile that lie in the row
#pragma unroll
    for (int s = 0; s < STEM_TP / 4; ++s) {
      if (s >= ns) break;
      float a[2], b[NI];
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) a[kt] = tile[koff[kt] + 24 * s] * kmask[kt];
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) b[ni] = dt[(4 * s + kq) * CP + ni * 16 + r];
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[kt][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kt], b[ni], acc[kt][ni], 0, 0, 0);
    }
  }
  // sum the four waves (the staging tiles are free now), one [28][Cout] slab per workgroup
  __syncthreads();
  float* red = &patch[0][0];  // [4][32][NI*16] floats = 4 * 1024 <= 4 * 3 * 392
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int e = 0; e < 4; ++e) red[(wv * 32 + kt * 16 + 4 * kq + e) * (NI * 16) + ni * 16 + r] = acc[kt][ni][e];
  __syncthreads();
  const int nw = 28 * NI * 16;
  for (int i = t; i < nw; i += 256)
    p.slabs[(size_t)blockIdx.x * nw + i] = red[i] + red[32 * NI * 16 + i] + red[2 * 32 * NI * 16 + i] + red[3 * 32 * NI * 16 + i];
}

static int stem_grid(int tiles, int per_cu) {
  static int env = -1;
  if (env < 0) {
    const char* e = getenv("DL3P_STEM_PER_CU");
    env = e ? atoi(e) : 0;
  }
  if (env > 0) per_cu = env;
  long long want = (long long)DL3P_NUM_CUS * per_cu;
  long long need = ceil_div_ll(ceil_div_ll(tiles, DL3P_NUM_XCDS), 4) * DL3P_NUM_XCDS;
  long long g = need < want ? need : want;
  if (g > DL3P_MAX_STAT_ROWS) g = DL3P_MAX_STAT_ROWS / DL3P_NUM_XCDS * DL3P_NUM_XCDS;
  if (g < DL3P_NUM_XCDS) g = DL3P_NUM_XCDS;
  return (int)g;
}

extern "C" int dl3p_stem_conv_supported(int Cin, int Cout, int k, int stride, int rate) {
  return Cin == 3 && (Cout == 16 || Cout == 32) && k == 3 && stride == 2 && rate == 1;
}

static int stem_fill(const char* who, StemParams* p, int N, int H, int W, int Cout, int pad_t, int pad_l, int Ho, int Wo) {
  DL3P_CHECK_ARG(Cout == 16 || Cout == 32, "%s: Cout must be 16 or 32 (got %d)", who, Cout);
  DL3P_CHECK_ARG(N > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0 && pad_t >= 0 && pad_l >= 0 && pad_t <= 2 && pad_l <= 2,
                 "%s: bad geometry", who);
  DL3P_CHECK_ARG(2 * (Ho - 1) - pad_t < H && 2 * (Wo - 1) - pad_l < W, "%s: output larger than the strided input", who);
  p->N = N; p->H = H; p->W = W; p->Ho = Ho; p->Wo = Wo; p->Cout = Cout; p->pad_t = pad_t; p->pad_l = pad_l;
  p->segs = ceil_div(Wo, STEM_TP);
  const long long tiles = (long long)N * Ho * p->segs;
  DL3P_CHECK_ARG(tiles < (1ll << 31) && (long long)N * H * W * 3 < (1ll << 40), "%s: tensor too large", who);
  p->tiles = (int)tiles;
  return DL3P_OK;
}

extern "C" int dl3p_stem_conv_fwd(const float* x, int ldx, const float* w, float* y, int ldy, float* stat_partials,
                                  int* rows_out, int N, int H, int W, int Cout, int pad_t, int pad_l, int Ho, int Wo,
                                  void* stream) {
  StemParams p = {};
  int rc = stem_fill("dl3p_stem_conv_fwd", &p, N, H, W, Cout, pad_t, pad_l, Ho, Wo);
  if (rc) return rc;
  DL3P_CHECK_ARG(x && w && y && ldx >= 3 && ldy >= Cout && ldy % 4 == 0 && aligned16(y), "dl3p_stem_conv_fwd: bad layout");
  p.x = x; p.ldx = ldx; p.w = w; p.y = y; p.ldy = ldy; p.partials = stat_partials;
  const int grid = stem_grid(p.tiles, 4);
  if (rows_out) *rows_out = grid;
  if (Cout == 32) dl3p_launch(stem_fwd_kernel<2>, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
  else dl3p_launch(stem_fwd_kernel<1>, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
  DL3P_CHECK_LAUNCH("dl3p_stem_conv_fwd");
  return DL3P_OK;
}

static int stem_wgrad_grid(int N, int Ho, int Wo) {
  const long long tiles = (long long)N * Ho * ceil_div(Wo, STEM_TP);
  return stem_grid((int)(tiles < (1ll << 31) ? tiles : (1ll << 31) - 1), 2);
}

extern "C" size_t dl3p_stem_conv_bwd_weight_workspace(int N, int Ho, int Wo, int Cout) {
  if (Cout != 16 && Cout != 32) return 0;
  return (size_t)stem_wgrad_grid(N, Ho, Wo) * 28 * Cout * sizeof(float);
}

struct StemFold { const float* z; int ldz; const float* scale; const float* shift; int act; const float* mean; const float* invstd;
                  const float* coef; };

static int stem_conv_bwd_weight_impl(const float* x, int ldx, const float* dy, int lddy, float* gw, float* workspace,
                                     size_t workspace_bytes, int N, int H, int W, int Cout, int pad_t, int pad_l,
                                     int Ho, int Wo, int* rows_out, void* stream, const StemFold* fold = nullptr) {
  StemParams p = {};
  int rc = stem_fill("dl3p_stem_conv_bwd_weight", &p, N, H, W, Cout, pad_t, pad_l, Ho, Wo);
  if (rc) return rc;
  DL3P_CHECK_ARG(x && dy && (rows_out || (gw && aligned16(gw))) && workspace && ldx >= 3 && lddy >= Cout && lddy % 4 == 0 &&
                     aligned16(dy) && aligned16(workspace),
                 "dl3p_stem_conv_bwd_weight: bad layout");
  const size_t need = dl3p_stem_conv_bwd_weight_workspace(N, Ho, Wo, Cout);
  DL3P_CHECK_ARG(workspace_bytes >= need, "dl3p_stem_conv_bwd_weight: workspace %zu < %zu bytes", workspace_bytes, need);
  p.x = x; p.ldx = ldx; p.dy = dy; p.lddy = lddy; p.slabs = workspace;
  const int grid = stem_wgrad_grid(N, Ho, Wo);
  if (fold) {
    DL3P_CHECK_ARG(fold->z && aligned16(fold->z) && fold->ldz >= Cout && fold->ldz % 4 == 0 && fold->mean && fold->invstd && fold->coef &&
                       (!fold->scale || fold->shift),
                   "dl3p_stem_conv_bwd_weight_slabs_bn: bad BatchNorm operands");
    p.f_z = fold->z; p.f_ldz = fold->ldz; p.f_scale = fold->scale; p.f_shift = fold->shift; p.f_act = fold->act;
    p.f_mean = fold->mean; p.f_invstd = fold->invstd; p.f_coef = fold->coef;
    if (Cout == 32) dl3p_launch(stem_wgrad_kernel<2, true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
    else dl3p_launch(stem_wgrad_kernel<1, true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
  } else if (Cout == 32) dl3p_launch(stem_wgrad_kernel<2>, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
  else dl3p_launch(stem_wgrad_kernel<1>, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
  DL3P_CHECK_LAUNCH("dl3p_stem_conv_bwd_weight");
  if (rows_out) { *rows_out = grid; return DL3P_OK; }
  return dl3p_reduce_rows_impl(workspace, grid, (size_t)28 * Cout, gw, 0, (hipStream_t)stream);
}

extern "C" int dl3p_stem_conv_bwd_weight(const float* x, int ldx, const float* dy, int lddy, float* gw, float* workspace,
                                         size_t workspace_bytes, int N, int H, int W, int Cout, int pad_t, int pad_l,
                                         int Ho, int Wo, void* stream) {
  return stem_conv_bwd_weight_impl(x, ldx, dy, lddy, gw, workspace, workspace_bytes, N, H, W, Cout, pad_t, pad_l, Ho, Wo, nullptr,
                                   stream);
}

extern "C" int dl3p_stem_conv_bwd_weight_slabs(const float* x, int ldx, const float* dy, int lddy, float* workspace,
                                               size_t workspace_bytes, int* rows_out, int N, int H, int W, int Cout, int pad_t,
                                               int pad_l, int Ho, int Wo, void* stream) {
  DL3P_CHECK_ARG(rows_out != nullptr, "dl3p_stem_conv_bwd_weight_slabs: rows_out is required");
  return stem_conv_bwd_weight_impl(x, ldx, dy, lddy, nullptr, workspace, workspace_bytes, N, H, W, Cout, pad_t, pad_l, Ho, Wo,
                                   rows_out, stream);
}

// dl3p_stem_conv_bwd_weight_slabs with the BatchNorm-backward apply of the stem's own BatchNorm (Conv_BN) folded in: the stem has no data
// gradient (its input is the image), so dz = apply(g, z) has this one reader and never needs to exist in HBM -- the separate pass read g
// and z and wrote dz (405 MB at 16 x 257 x 257 x 32), this kernel reads g and z where it read dz.
extern "C" int dl3p_stem_conv_bwd_weight_slabs_bn(const float* x, int ldx, const float* g, int ldg, const float* z, int ldz,
                                                  const float* bn_scale, const float* bn_shift, int bn_act, const float* save_mean,
                                                  const float* save_invstd, const float* coef, float* workspace,
                                                  size_t workspace_bytes, int* rows_out, int N, int H, int W, int Cout, int pad_t,
                                                  int pad_l, int Ho, int Wo, void* stream) {
  DL3P_CHECK_ARG(rows_out != nullptr, "dl3p_stem_conv_bwd_weight_slabs_bn: rows_out is required");
  const StemFold f = {z, ldz, bn_scale, bn_shift, bn_act, save_mean, save_invstd, coef};
  return stem_conv_bwd_weight_impl(x, ldx, g, ldg, nullptr, workspace, workspace_bytes, N, H, W, Cout, pad_t, pad_l, Ho, Wo, rows_out,
                                   stream, &f);
}
