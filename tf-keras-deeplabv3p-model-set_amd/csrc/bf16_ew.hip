// bf16-storage twins of the streaming kernels (BatchNorm backward, materialise / residual / dropout, pooling,
// squeeze-excite multiply, bilinear resize, im2col, boundary conversions) for the mixed-precision path of
// BASELINE.json configs[4] (reference switch: train.py:37-46).  Same call sites as their fp32 twins in
// bn_elementwise.hip / resize_head.hip / conv.hip; tensors in HBM are bf16, all arithmetic is fp32 (bf16.h states the
// rounding points).  HBM-bound: 16-byte lanes (8 channels) when C % 8 == 0, 8-byte lanes otherwise; channel lanes fastest.
#include "bf16.h"

namespace {

struct EwB {
  const bf16* a; int lda;
  const bf16* z; int ldz;
  const float* scale; const float* shift; int act;
  const float* mean; const float* invstd; const float* coef;
  const bf16* r; int ldr; const float* rscale; const float* rshift; int ract;
  bf16* out; int ldo;
  float* partials;
  long long M; int C;
  int cs, px, nslab, nbx;
  float rate; uint64_t seed; const int64_t* step;
  int accumulate;
};

inline bool vec8(int C, int lda, int ldb = 8, int ldc = 8) { return C % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && ldc % 8 == 0; }

inline int ew_setup_b(EwB& p, long long M, int C, int V, int max_rows, int per_cu = 4) {
  const LaneSplit s = lane_split(C, V);
  p.cs = s.cs; p.px = s.px; p.nslab = s.nslab;
  // at least 4 rows per thread: the per-thread coefficient loads and the launch are amortised over them
  long long need = ceil_div_ll(M, (long long)p.px * 4);
  long long target = DL3P_NUM_CUS * per_cu / p.nslab;
  if (target < 1) target = 1;
  long long nbx = need < target ? need : target;
  if (nbx < 1) nbx = 1;
  if (nbx > max_rows) nbx = max_rows;
  p.nbx = (int)nbx; p.M = M; p.C = C;
  return p.nbx;
}

// per-workgroup reduction over the pixel lanes of NV vectors held per thread -> out_row[v][ldc] for this slab's channels
template <int V, int NV>
__device__ __forceinline__ void block_reduce_rows(const fvec<V> (&vals)[NV], bool active, int pl, int cl, int cs, int px,
                                                  int cbase, int ldc, float* out_row) {
  __shared__ float sm[256 * V];
  for (int v = 0; v < NV; ++v) {
    __syncthreads();
    if (active) {
#pragma unroll
      for (int i = 0; i < V; ++i) sm[(pl * cs + cl) * V + i] = vals[v].v[i];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < cs * V; e += 256) {
      float a = sm[e];
      for (int q = 1; q < px; ++q) a += sm[q * cs * V + e];
      out_row[(size_t)v * ldc + cbase + e] = a;
    }
  }
}

template <int V>
__global__ __launch_bounds__(256) void bn_bwd_reduce_b(EwB p) {
  const int slab = blockIdx.x / p.nbx, bx = blockIdx.x - slab * p.nbx;
  const int pl = threadIdx.x / p.cs, cl = threadIdx.x - pl * p.cs;
  const bool active = pl < p.px;
  const int cbase = slab * p.cs * V, c = cbase + cl * V;
  fvec<V> acc[2] = {fzero<V>(), fzero<V>()};
  if (active) {
    const fvec<V> sc = ldv_f32_or<V>(p.scale, c, 1.f), sh = ldv_f32_or<V>(p.shift, c, 0.f);
    const fvec<V> mu = ldv_f32_or<V>(p.mean, c, 0.f), is = ldv_f32_or<V>(p.invstd, c, 1.f);
    for (long long m = (long long)bx * p.px + pl; m < p.M; m += (long long)p.nbx * p.px) {
      const fvec<V> g = ldv<V>(p.a + (size_t)m * p.lda + c), z = ldv<V>(p.z + (size_t)m * p.ldz + c);
#pragma unroll
      for (int i = 0; i < V; ++i) {
        const float d = g.v[i] * act_grad(fmaf(z.v[i], sc.v[i], sh.v[i]), p.act);
        acc[0].v[i] += d;
        acc[1].v[i] = fmaf(d, (z.v[i] - mu.v[i]) * is.v[i], acc[1].v[i]);
      }
    }
  }
  block_reduce_rows<V, 2>(acc, active, pl, cl, p.cs, p.px, cbase, p.C, p.partials + (size_t)bx * 2 * p.C);
}

template <int V>
__global__ __launch_bounds__(256) void bn_bwd_apply_b(EwB p) {
  const int slab = blockIdx.x / p.nbx, bx = blockIdx.x - slab * p.nbx;
  const int pl = threadIdx.x / p.cs, cl = threadIdx.x - pl * p.cs;
  if (pl >= p.px) return;
  const int c = (slab * p.cs + cl) * V;
  const fvec<V> sc = ldv_f32_or<V>(p.scale, c, 1.f), sh = ldv_f32_or<V>(p.shift, c, 0.f);
  const fvec<V> mu = ldv_f32_or<V>(p.mean, c, 0.f), is = ldv_f32_or<V>(p.invstd, c, 1.f);
  const fvec<V> c0 = ldv_f32_or<V>(p.coef, c, 1.f);
  const fvec<V> c1 = ldv_f32_or<V>(p.coef ? p.coef + p.C : nullptr, c, 0.f);
  const fvec<V> c2 = ldv_f32_or<V>(p.coef ? p.coef + 2 * p.C : nullptr, c, 0.f);
  for (long long m = (long long)bx * p.px + pl; m < p.M; m += (long long)p.nbx * p.px) {
    const fvec<V> g = ldv<V>(p.a + (size_t)m * p.lda + c), z = ldv<V>(p.z + (size_t)m * p.ldz + c);
    bf16* op = p.out + (size_t)m * p.ldo + c;
    fvec<V> o;
#pragma unroll
    for (int i = 0; i < V; ++i) {
      const float d = g.v[i] * act_grad(fmaf(z.v[i], sc.v[i], sh.v[i]), p.act);
      o.v[i] = c0.v[i] * (d - c1.v[i] - (z.v[i] - mu.v[i]) * is.v[i] * c2.v[i]);
    }
    if (p.accumulate) {
      const fvec<V> old = ldv<V>(op);
#pragma unroll
      for (int i = 0; i < V; ++i) o.v[i] += old.v[i];
    }
    stv<V>(op, o);
  }
}

template <int V>
__global__ __launch_bounds__(256) void affine_act_b(EwB p) {
  const int slab = blockIdx.x / p.nbx, bx = blockIdx.x - slab * p.nbx;
  const int pl = threadIdx.x / p.cs, cl = threadIdx.x - pl * p.cs;
  if (pl >= p.px) return;
  const int c = (slab * p.cs + cl) * V;
  const fvec<V> sc = ldv_f32_or<V>(p.scale, c, 1.f), sh = ldv_f32_or<V>(p.shift, c, 0.f);
  const fvec<V> rsc = ldv_f32_or<V>(p.rscale, c, 1.f), rsh = ldv_f32_or<V>(p.rshift, c, 0.f);
  const int64_t step = (p.rate > 0.f && p.step) ? *p.step : 0;
  const float keep_scale = p.rate > 0.f ? 1.f / (1.f - p.rate) : 1.f;
  for (long long m = (long long)bx * p.px + pl; m < p.M; m += (long long)p.nbx * p.px) {
    fvec<V> v = prologue_bf16<V>(ldv<V>(p.a + (size_t)m * p.lda + c), sc, sh, p.act);
    if (p.rate > 0.f) {
      const uint64_t e = (uint64_t)m * p.C + c;
#pragma unroll
      for (int i = 0; i < V; ++i) v.v[i] = dropout_keep(p.seed, step, e + i, p.rate) ? v.v[i] * keep_scale : 0.f;
    }
    if (p.r) {
      const fvec<V> rv = prologue_bf16<V>(ldv<V>(p.r + (size_t)m * p.ldr + c), rsc, rsh, p.ract);
#pragma unroll
      for (int i = 0; i < V; ++i) v.v[i] += rv.v[i];
    }
    stv<V>(p.out + (size_t)m * p.ldo + c, v);
  }
}

template <int V>
__global__ __launch_bounds__(256) void scale_mask_bwd_b(EwB p) {
  const int slab = blockIdx.x / p.nbx, bx = blockIdx.x - slab * p.nbx;
  const int pl = threadIdx.x / p.cs, cl = threadIdx.x - pl * p.cs;
  if (pl >= p.px) return;
  const int c = (slab * p.cs + cl) * V;
  const int64_t step = (p.rate > 0.f && p.step) ? *p.step : 0;
  const float keep_scale = p.rate > 0.f ? 1.f / (1.f - p.rate) : 1.f;
  for (long long m = (long long)bx * p.px + pl; m < p.M; m += (long long)p.nbx * p.px) {
    fvec<V> v = ldv<V>(p.a + (size_t)m * p.lda + c);
    if (p.rate > 0.f) {
      const uint64_t e = (uint64_t)m * p.C + c;
#pragma unroll
      for (int i = 0; i < V; ++i) v.v[i] = dropout_keep(p.seed, step, e + i, p.rate) ? v.v[i] * keep_scale : 0.f;
    }
    bf16* o = p.out + (size_t)m * p.ldo + c;
    if (p.accumulate) {
      const fvec<V> old = ldv<V>(o);
#pragma unroll
      for (int i = 0; i < V; ++i) v.v[i] += old.v[i];
    }
    stv<V>(o, v);
  }
}

int check_b(const char* fn, const void* a, int lda, int C) {
  DL3P_CHECK_ARG(a != nullptr, "%s: null pointer", fn);
  DL3P_CHECK_ARG(C > 0 && C % 4 == 0, "%s: C=%d must be a positive multiple of 4", fn, C);
  DL3P_CHECK_ARG(lda % 4 == 0 && lda >= C && (reinterpret_cast<uintptr_t>(a) & 7u) == 0, "%s: bad layout (ld=%d)", fn, lda);
  return DL3P_OK;
}
inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

#define EW_LAUNCH(kernel, use8, p, st)                                                                         \
  do {                                                                                                         \
    if (use8) hipLaunchKernelGGL((kernel<8>), dim3(p.nbx * p.nslab), dim3(256), 0, st, p);                     \
    else hipLaunchKernelGGL((kernel<4>), dim3(p.nbx * p.nslab), dim3(256), 0, st, p);                          \
  } while (0)

}  // namespace

extern "C" int dl3p_bn_bwd_reduce_bf16(const void* g, int ldg, const void* z, int ldz, const float* scale,
                                       const float* shift, int act, const float* save_mean, const float* save_invstd,
                                       float* partials, int* rows_out, int M, int C, void* stream) {
  int rc = check_b("dl3p_bn_bwd_reduce_bf16", g, ldg, C);
  if (rc) return rc;
  rc = check_b("dl3p_bn_bwd_reduce_bf16", z, ldz, C);
  if (rc) return rc;
  DL3P_CHECK_ARG(partials && M > 0, "dl3p_bn_bwd_reduce_bf16: bad arguments");
  EwB p = {};
  p.a = (const bf16*)g; p.lda = ldg; p.z = (const bf16*)z; p.ldz = ldz; p.scale = scale; p.shift = shift; p.act = act;
  p.mean = save_mean; p.invstd = save_invstd; p.partials = partials;
  const bool v8 = vec8(C, ldg, ldz) && al16(g) && al16(z);
  const int rows = ew_setup_b(p, M, C, v8 ? 8 : 4, DL3P_MAX_STAT_ROWS);
  if (rows_out) *rows_out = rows;
  EW_LAUNCH(bn_bwd_reduce_b, v8, p, (hipStream_t)stream);
  DL3P_CHECK_LAUNCH("dl3p_bn_bwd_reduce_bf16");
  return DL3P_OK;
}

extern "C" int dl3p_bn_bwd_apply_bf16(const void* g, int ldg, const void* z, int ldz, const float* scale,
                                      const float* shift, int act, const float* save_mean, const float* save_invstd,
                                      const float* coef, void* dz, int lddz, int accumulate, int M, int C, void* stream) {
  int rc = check_b("dl3p_bn_bwd_apply_bf16", g, ldg, C);
  if (rc) return rc;
  rc = check_b("dl3p_bn_bwd_apply_bf16", z, ldz, C);
  if (rc) return rc;
  rc = check_b("dl3p_bn_bwd_apply_bf16", dz, lddz, C);
  if (rc) return rc;
  EwB p = {};
  p.a = (const bf16*)g; p.lda = ldg; p.z = (const bf16*)z; p.ldz = ldz; p.scale = scale; p.shift = shift; p.act = act;
  p.mean = save_mean; p.invstd = save_invstd; p.coef = coef; p.out = (bf16*)dz; p.ldo = lddz; p.accumulate = accumulate;
  const bool v8 = vec8(C, ldg, ldz, lddz) && al16(g) && al16(z) && al16(dz);
  ew_setup_b(p, M, C, v8 ? 8 : 4, 1 << 20, 8);
  EW_LAUNCH(bn_bwd_apply_b, v8, p, (hipStream_t)stream);
  DL3P_CHECK_LAUNCH("dl3p_bn_bwd_apply_bf16");
  return DL3P_OK;
}

extern "C" int dl3p_affine_act_bf16(const void* x, int ldx, const float* scale, const float* shift, int act,
                                    const void* r, int ldr, const float* rscale, const float* rshift, int ract,
                                    float dropout_rate, uint64_t seed, const int64_t* step_counter, void* y, int ldy,
                                    int M, int C, void* stream) {
  int rc = check_b("dl3p_affine_act_bf16", x, ldx, C);
  if (rc) return rc;
  rc = check_b("dl3p_affine_act_bf16", y, ldy, C);
  if (rc) return rc;
  if (r) { rc = check_b("dl3p_affine_act_bf16", r, ldr, C); if (rc) return rc; }
  DL3P_CHECK_ARG(dropout_rate >= 0.f && dropout_rate < 1.f && M > 0, "dl3p_affine_act_bf16: bad arguments");
  EwB p = {};
  p.a = (const bf16*)x; p.lda = ldx; p.scale = scale; p.shift = shift; p.act = act;
  p.r = (const bf16*)r; p.ldr = ldr; p.rscale = rscale; p.rshift = rshift; p.ract = ract;
  p.rate = dropout_rate; p.seed = seed; p.step = step_counter; p.out = (bf16*)y; p.ldo = ldy;
  const bool v8 = vec8(C, ldx, ldy, r ? ldr : 8) && al16(x) && al16(y) && (!r || al16(r));
  ew_setup_b(p, M, C, v8 ? 8 : 4, 1 << 20, 8);
  EW_LAUNCH(affine_act_b, v8, p, (hipStream_t)stream);
  DL3P_CHECK_LAUNCH("dl3p_affine_act_bf16");
  return DL3P_OK;
}

extern "C" int dl3p_scale_mask_bwd_bf16(const void* gy, int ldgy, float dropout_rate, uint64_t seed,
                                        const int64_t* step_counter, void* gx, int ldgx, int accumulate, int M, int C,
                                        void* stream) {
  int rc = check_b("dl3p_scale_mask_bwd_bf16", gy, ldgy, C);
  if (rc) return rc;
  rc = check_b("dl3p_scale_mask_bwd_bf16", gx, ldgx, C);
  if (rc) return rc;
  EwB p = {};
  p.a = (const bf16*)gy; p.lda = ldgy; p.rate = dropout_rate; p.seed = seed; p.step = step_counter;
  p.out = (bf16*)gx; p.ldo = ldgx; p.accumulate = accumulate;
  const bool v8 = vec8(C, ldgy, ldgx) && al16(gy) && al16(gx);
  ew_setup_b(p, M, C, v8 ? 8 : 4, 1 << 20, 8);
  EW_LAUNCH(scale_mask_bwd_b, v8, p, (hipStream_t)stream);
  DL3P_CHECK_LAUNCH("dl3p_scale_mask_bwd_bf16");
  return DL3P_OK;
}

// ------------------------------------------------------------------------------ per-image reductions (pooling, SE)
// Two launches, no tickets: (image, pixel chunk, channel slab) workgroups leave fp32 partial rows in the workspace,
// a second small kernel adds the chunks of an image in chunk order (deterministic) and writes the bf16 row.
namespace {

struct PoolB {
  const bf16* x; int ldx; const float* scale; const float* shift; int act;
  const bf16* s; int lds; int s_act;
  const bf16* gy; int ldgy;
  bf16* out; int ldo; int accumulate;
  float* ws;            // [N][nchunk][C]
  int N, HW, C, nchunk, per, cs, px, nslab;
};

inline void pool_plan_b(PoolB& p, int N, int HW, int C) {
  // reductions want pixel lanes: the NARROWEST channel-lane group that still reads >= 64 contiguous bytes per pixel
  const int cv = C / 4;
  int cs = cv;
  for (int d = 8; d < cv; ++d)
    if (cv % d == 0) { cs = d; break; }
  if (cs > 256) cs = 256;                      // (C > 1024 with no divisor in [8, 256]: not a shape of these models)
  p.cs = cs; p.px = 256 / cs; p.nslab = cv / cs;
  int nchunk = (2 * DL3P_NUM_CUS + N * p.nslab - 1) / (N * p.nslab);
  const int most = (HW + 4 * p.px - 1) / (4 * p.px);
  if (nchunk > most) nchunk = most;
  if (nchunk > 32) nchunk = 32;
  if (nchunk < 1) nchunk = 1;
  p.per = (HW + nchunk - 1) / nchunk;
  p.nchunk = (HW + p.per - 1) / p.per;
  p.N = N; p.HW = HW; p.C = C;
}

// partial[n][chunk][c] = sum over the chunk's pixels of (gy *) act(x*scale+shift); with `out` set also
// out[n,i,c] (+)= gy * act_s(s[n,c])  (the squeeze-excite multiply's gradient w.r.t. its tensor input)
template <bool SE_BWD>
__global__ __launch_bounds__(256) void pool_partial_b(PoolB p) {
  const int chunk = blockIdx.x % p.nchunk;
  const int rest = blockIdx.x / p.nchunk;
  const int n = rest / p.nslab, slab = rest - n * p.nslab;
  const int pl = threadIdx.x / p.cs, cl = threadIdx.x - pl * p.cs;
  const bool active = pl < p.px;
  const int cbase = slab * p.cs * 4, c = cbase + cl * 4;
  fvec<4> acc[1] = {fzero<4>()};
  if (active) {
    const fvec<4> sc = ldv_f32_or<4>(p.scale, c, 1.f), sh = ldv_f32_or<4>(p.shift, c, 0.f);
    fvec<4> sv = fzero<4>();
    if (SE_BWD) {
      sv = ldv<4>(p.s + (size_t)n * p.lds + c);
#pragma unroll
      for (int i = 0; i < 4; ++i) sv.v[i] = bf16_round(act_apply(sv.v[i], p.s_act));
    }
    const int i1 = min(p.HW, (chunk + 1) * p.per);
    for (int i = chunk * p.per + pl; i < i1; i += p.px) {
      const size_t m = (size_t)n * p.HW + i;
      const fvec<4> a = prologue_bf16<4>(ldv<4>(p.x + m * p.ldx + c), sc, sh, p.act);
      if (SE_BWD) {
        const fvec<4> g = ldv<4>(p.gy + m * p.ldgy + c);
        fvec<4> o;
#pragma unroll
        for (int k = 0; k < 4; ++k) { acc[0].v[k] = fmaf(g.v[k], a.v[k], acc[0].v[k]); o.v[k] = g.v[k] * sv.v[k]; }
        bf16* op = p.out + m * p.ldo + c;
        if (p.accumulate) {
          const fvec<4> old = ldv<4>(op);
#pragma unroll
          for (int k = 0; k < 4; ++k) o.v[k] += old.v[k];
        }
        stv<4>(op, o);
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[0].v[k] += a.v[k];
      }
    }
  }
  block_reduce_rows<4, 1>(acc, active, pl, cl, p.cs, p.px, cbase, p.C, p.ws + ((size_t)n * p.nchunk + chunk) * p.C);
}

// out[n][c] = scale * sum over chunks (chunk order: deterministic); 64 channels x 4 chunk lanes per workgroup
__global__ __launch_bounds__(256) void pool_finish_b(const float* ws, int nchunk, int C, float scale, bf16* out, int ldo) {
  __shared__ float sm[4][64];
  const int n = blockIdx.y;
  const int cl = threadIdx.x & 63, ql = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  float a = 0.f;
  if (c < C)
    for (int q = ql; q < nchunk; q += 4) a += ws[((size_t)n * nchunk + q) * C + c];
  sm[ql][cl] = a;
  __syncthreads();
  if (ql == 0 && c < C) out[(size_t)n * ldo + c] = (bf16)(((sm[0][cl] + sm[1][cl]) + (sm[2][cl] + sm[3][cl])) * scale);
}

template <int V>
__global__ __launch_bounds__(256) void gap_bwd_b(EwB p, int HW) {
  const int slab = blockIdx.x / p.nbx, bx = blockIdx.x - slab * p.nbx;
  const int pl = threadIdx.x / p.cs, cl = threadIdx.x - pl * p.cs;
  if (pl >= p.px) return;
  const int c = (slab * p.cs + cl) * V;
  const float inv = 1.f / (float)HW;
  for (long long m = (long long)bx * p.px + pl; m < p.M; m += (long long)p.nbx * p.px) {
    const long long n = m / HW;
    fvec<V> g = ldv<V>(p.a + (size_t)n * p.lda + c);
    bf16* o = p.out + (size_t)m * p.ldo + c;
#pragma unroll
    for (int i = 0; i < V; ++i) g.v[i] *= inv;
    if (p.accumulate) {
      const fvec<V> old = ldv<V>(o);
#pragma unroll
      for (int i = 0; i < V; ++i) g.v[i] += old.v[i];
    }
    stv<V>(o, g);
  }
}

template <int V>
__global__ __launch_bounds__(256) void scale_bcast_fwd_b(EwB p, int HW, const bf16* s, int lds, int s_act) {
  const int slab = blockIdx.x / p.nbx, bx = blockIdx.x - slab * p.nbx;
  const int pl = threadIdx.x / p.cs, cl = threadIdx.x - pl * p.cs;
  if (pl >= p.px) return;
  const int c = (slab * p.cs + cl) * V;
  const fvec<V> sc = ldv_f32_or<V>(p.scale, c, 1.f), sh = ldv_f32_or<V>(p.shift, c, 0.f);
  for (long long m = (long long)bx * p.px + pl; m < p.M; m += (long long)p.nbx * p.px) {
    const long long n = m / HW;
    const fvec<V> a = prologue_bf16<V>(ldv<V>(p.a + (size_t)m * p.lda + c), sc, sh, p.act);
    const fvec<V> sv = ldv<V>(s + (size_t)n * lds + c);
    fvec<V> o;
#pragma unroll
    for (int i = 0; i < V; ++i) o.v[i] = a.v[i] * bf16_round(act_apply(sv.v[i], s_act));
    stv<V>(p.out + (size_t)m * p.ldo + c, o);
  }
}

}  // namespace

extern "C" size_t dl3p_pool_workspace_bf16(int N, int HW, int C) {
  if (N <= 0 || HW <= 0 || C <= 0 || C % 4) return 0;
  PoolB p = {};
  pool_plan_b(p, N, HW, C);
  return (size_t)N * p.nchunk * C * sizeof(float);
}

extern "C" int dl3p_global_avgpool_fwd_bf16(const void* x, int ldx, const float* in_scale, const float* in_shift,
                                            int in_act, void* y, int ldy, float out_scale, int N, int HW, int C,
                                            float* workspace, size_t workspace_bytes, void* stream) {
  int rc = check_b("dl3p_global_avgpool_fwd_bf16", x, ldx, C);
  if (rc) return rc;
  DL3P_CHECK_ARG(y && ldy >= C && N > 0 && HW > 0, "dl3p_global_avgpool_fwd_bf16: bad arguments");
  DL3P_CHECK_ARG(workspace && workspace_bytes >= dl3p_pool_workspace_bf16(N, HW, C),
                 "dl3p_global_avgpool_fwd_bf16: workspace of dl3p_pool_workspace_bf16() bytes required");
  PoolB p = {};
  pool_plan_b(p, N, HW, C);
  p.x = (const bf16*)x; p.ldx = ldx; p.scale = in_scale; p.shift = in_shift; p.act = in_act; p.ws = workspace;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL((pool_partial_b<false>), dim3(N * p.nslab * p.nchunk), dim3(256), 0, st, p);
  hipLaunchKernelGGL(pool_finish_b, dim3(ceil_div(C, 64), N), dim3(256), 0, st, workspace, p.nchunk, C,
                     out_scale / (float)HW, (bf16*)y, ldy);
  DL3P_CHECK_LAUNCH("dl3p_global_avgpool_fwd_bf16");
  return DL3P_OK;
}

extern "C" int dl3p_global_avgpool_bwd_bf16(const void* gy, int ldgy, void* gx, int ldgx, int accumulate, int N, int HW,
                                            int C, void* stream) {
  int rc = check_b("dl3p_global_avgpool_bwd_bf16", gy, ldgy, C);
  if (rc) return rc;
  rc = check_b("dl3p_global_avgpool_bwd_bf16", gx, ldgx, C);
  if (rc) return rc;
  EwB p = {};
  p.a = (const bf16*)gy; p.lda = ldgy; p.out = (bf16*)gx; p.ldo = ldgx; p.accumulate = accumulate;
  const bool v8 = vec8(C, ldgy, ldgx) && al16(gy) && al16(gx);
  ew_setup_b(p, (long long)N * HW, C, v8 ? 8 : 4, 1 << 20, 8);
  if (v8) hipLaunchKernelGGL((gap_bwd_b<8>), dim3(p.nbx * p.nslab), dim3(256), 0, (hipStream_t)stream, p, HW);
  else hipLaunchKernelGGL((gap_bwd_b<4>), dim3(p.nbx * p.nslab), dim3(256), 0, (hipStream_t)stream, p, HW);
  DL3P_CHECK_LAUNCH("dl3p_global_avgpool_bwd_bf16");
  return DL3P_OK;
}

extern "C" int dl3p_scale_bcast_fwd_bf16(const void* x, int ldx, const float* scale, const float* shift, int act,
                                         const void* s, int lds, int s_act, void* y, int ldy, int N, int HW, int C,
                                         void* stream) {
  int rc = check_b("dl3p_scale_bcast_fwd_bf16", x, ldx, C);
  if (rc) return rc;
  rc = check_b("dl3p_scale_bcast_fwd_bf16", y, ldy, C);
  if (rc) return rc;
  rc = check_b("dl3p_scale_bcast_fwd_bf16", s, lds, C);
  if (rc) return rc;
  EwB p = {};
  p.a = (const bf16*)x; p.lda = ldx; p.scale = scale; p.shift = shift; p.act = act; p.out = (bf16*)y; p.ldo = ldy;
  const bool v8 = vec8(C, ldx, ldy, lds) && al16(x) && al16(y) && al16(s);
  ew_setup_b(p, (long long)N * HW, C, v8 ? 8 : 4, 1 << 20, 8);
  if (v8) hipLaunchKernelGGL((scale_bcast_fwd_b<8>), dim3(p.nbx * p.nslab), dim3(256), 0, (hipStream_t)stream, p, HW, (const bf16*)s, lds, s_act);
  else hipLaunchKernelGGL((scale_bcast_fwd_b<4>), dim3(p.nbx * p.nslab), dim3(256), 0, (hipStream_t)stream, p, HW, (const bf16*)s, lds, s_act);
  DL3P_CHECK_LAUNCH("dl3p_scale_bcast_fwd_bf16");
  return DL3P_OK;
}

extern "C" int dl3p_scale_bcast_bwd_bf16(const void* gy, int ldgy, const void* x, int ldx, const float* scale,
                                         const float* shift, int act, const void* s, int lds, int s_act, void* gx,
                                         int ldgx, int accumulate_gx, void* gs, int ldgs, int N, int HW, int C,
                                         float* workspace, size_t workspace_bytes, void* stream) {
  int rc = check_b("dl3p_scale_bcast_bwd_bf16", gy, ldgy, C);
  if (rc) return rc;
  rc = check_b("dl3p_scale_bcast_bwd_bf16", x, ldx, C);
  if (rc) return rc;
  rc = check_b("dl3p_scale_bcast_bwd_bf16", gx, ldgx, C);
  if (rc) return rc;
  DL3P_CHECK_ARG(s && gs && N > 0 && HW > 0 && workspace && workspace_bytes >= dl3p_pool_workspace_bf16(N, HW, C),
                 "dl3p_scale_bcast_bwd_bf16: bad arguments / workspace of dl3p_pool_workspace_bf16() bytes required");
  PoolB p = {};
  pool_plan_b(p, N, HW, C);
  p.x = (const bf16*)x; p.ldx = ldx; p.scale = scale; p.shift = shift; p.act = act; p.s = (const bf16*)s; p.lds = lds;
  p.s_act = s_act; p.gy = (const bf16*)gy; p.ldgy = ldgy; p.out = (bf16*)gx; p.ldo = ldgx; p.accumulate = accumulate_gx;
  p.ws = workspace;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL((pool_partial_b<true>), dim3(N * p.nslab * p.nchunk), dim3(256), 0, st, p);
  hipLaunchKernelGGL(pool_finish_b, dim3(ceil_div(C, 64), N), dim3(256), 0, st, workspace, p.nchunk, C, 1.f, (bf16*)gs, ldgs);
  DL3P_CHECK_LAUNCH("dl3p_scale_bcast_bwd_bf16");
  return DL3P_OK;
}

// ------------------------------------------------------------------------------ bilinear resize
namespace {
struct LerpB { int lo, hi; float t; };
__device__ __forceinline__ LerpB lerp_b(int o, float scale, int in_size) {
  const float src = ((float)o + 0.5f) * scale - 0.5f;
  const float fl = floorf(src);
  LerpB r;
  r.lo = max((int)fl, 0);
  r.hi = min((int)ceilf(src), in_size - 1);
  r.t = src - fl;
  return r;
}
struct ResizeB {
  const bf16* x; int ldx; bf16* y; int ldy;
  int N, h, w, C, H, W, cs, px, nslab, nbx;
  long long total;
  int accumulate;
  int v;               // channels per thread (8: 16-byte accesses; 4)
};

// V channels per thread: 8 (16-byte accesses) wherever the channel count and the row pitches allow, else 4.  32-bit index arithmetic
// (hosts check the pixel counts).
template <int V>
__global__ __launch_bounds__(256) void resize_fwd_b(ResizeB p) {
  const int slab = blockIdx.x / p.nbx, bx = blockIdx.x - slab * p.nbx;
  const int pl = threadIdx.x / p.cs, cl = threadIdx.x - pl * p.cs;
  if (pl >= p.px) return;
  const int c = (slab * p.cs + cl) * V;
  const float sy = (float)p.h / (float)p.H, sx = (float)p.w / (float)p.W;
  const int total = (int)p.total, step = p.nbx * p.px;
  for (int s = bx * p.px + pl; s < total; s += step) {
    const int row = s / p.W, ox = s - row * p.W;
    const int n = row / p.H, oy = row - n * p.H;
    const LerpB ly = lerp_b(oy, sy, p.h), lx = lerp_b(ox, sx, p.w);
    const bf16* img = p.x + (size_t)n * p.h * p.w * p.ldx + c;
    const fvec<V> tl = ldv<V>(img + ((size_t)ly.lo * p.w + lx.lo) * p.ldx), tr = ldv<V>(img + ((size_t)ly.lo * p.w + lx.hi) * p.ldx);
    const fvec<V> bl = ldv<V>(img + ((size_t)ly.hi * p.w + lx.lo) * p.ldx), br = ldv<V>(img + ((size_t)ly.hi * p.w + lx.hi) * p.ldx);
    fvec<V> o;
#pragma unroll
    for (int i = 0; i < V; ++i) {
      const float top = tl.v[i] + (tr.v[i] - tl.v[i]) * lx.t;
      const float bot = bl.v[i] + (br.v[i] - bl.v[i]) * lx.t;
      o.v[i] = top + (bot - top) * ly.t;
    }
    stv<V>(p.y + (size_t)s * p.ldy + c, o);
  }
}

__device__ __forceinline__ void touch_b(int i, float inv_scale, int out_size, int& o0, int& o1) {
  const float a = ((float)i - 0.5f) * inv_scale - 0.5f;
  const float b = ((float)i + 1.5f) * inv_scale - 0.5f;
  o0 = max((int)floorf(a) - 1, 0);
  o1 = min((int)ceilf(b) + 1, out_size - 1);
}

// gather form of the transpose (deterministic): each input pixel sums the output pixels that read it
template <int V>
__global__ __launch_bounds__(256) void resize_bwd_b(ResizeB p) {
  const int slab = blockIdx.x / p.nbx, bx = blockIdx.x - slab * p.nbx;
  const int pl = threadIdx.x / p.cs, cl = threadIdx.x - pl * p.cs;
  if (pl >= p.px) return;
  const int c = (slab * p.cs + cl) * V;
  const float sy = (float)p.h / (float)p.H, sx = (float)p.w / (float)p.W;
  const float isy = (float)p.H / (float)p.h, isx = (float)p.W / (float)p.w;
  const int total = (int)p.total, step = p.nbx * p.px;
  for (int s = bx * p.px + pl; s < total; s += step) {
    const int row = s / p.w, ix = s - row * p.w;
    const int n = row / p.h, iy = row - n * p.h;
    int y0, y1, x0, x1;
    touch_b(iy, isy, p.H, y0, y1);
    touch_b(ix, isx, p.W, x0, x1);
    if (iy == 0) y0 = 0;
    if (iy == p.h - 1) y1 = p.H - 1;
    if (ix == 0) x0 = 0;
    if (ix == p.w - 1) x1 = p.W - 1;
    fvec<V> acc = fzero<V>();
    const bf16* gimg = p.x + (size_t)n * p.H * p.W * p.ldx + c;
    for (int oy = y0; oy <= y1; ++oy) {
      const LerpB ly = lerp_b(oy, sy, p.h);
      const float wy = (ly.lo == iy ? 1.f - ly.t : 0.f) + (ly.hi == iy ? ly.t : 0.f);
      if (wy == 0.f) continue;
      for (int ox = x0; ox <= x1; ++ox) {
        const LerpB lx = lerp_b(ox, sx, p.w);
        const float wx = (lx.lo == ix ? 1.f - lx.t : 0.f) + (lx.hi == ix ? lx.t : 0.f);
        if (wx == 0.f) continue;
        const float wgt = wy * wx;
        const fvec<V> g = ldv<V>(gimg + ((size_t)oy * p.W + ox) * p.ldx);
#pragma unroll
        for (int i = 0; i < V; ++i) acc.v[i] = fmaf(g.v[i], wgt, acc.v[i]);
      }
    }
    bf16* o = p.y + (size_t)s * p.ldy + c;
    if (p.accumulate) {
      const fvec<V> old = ldv<V>(o);
#pragma unroll
      for (int i = 0; i < V; ++i) acc.v[i] += old.v[i];
    }
    stv<V>(o, acc);
  }
}

int resize_setup_b(const char* fn, ResizeB& p, const void* x, int ldx, void* y, int ldy, int N, int h, int w, int C, int H,
                   int W, long long total) {
  int rc = check_b(fn, x, ldx, C);
  if (rc) return rc;
  rc = check_b(fn, y, ldy, C);
  if (rc) return rc;
  p.x = (const bf16*)x; p.ldx = ldx; p.y = (bf16*)y; p.ldy = ldy; p.N = N; p.h = h; p.w = w; p.C = C; p.H = H; p.W = W;
  DL3P_CHECK_ARG(total < (1ll << 31), "%s: tensor too large", fn);
  p.v = (C % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0) ? 8 : 4;
  const LaneSplit s = lane_split(C, p.v);
  p.cs = s.cs; p.px = s.px; p.nslab = s.nslab;
  p.total = total;
  long long nbx = ceil_div_ll(total, p.px);
  const long long target = DL3P_NUM_CUS * 8 / p.nslab > 0 ? DL3P_NUM_CUS * 8 / p.nslab : 1;
  if (nbx > target) nbx = target;
  p.nbx = (int)nbx;
  return DL3P_OK;
}
}  // namespace

extern "C" int dl3p_resize_bilinear_fwd_bf16(const void* x, int ldx, void* y, int ldy, int N, int h, int w, int C, int H,
                                             int W, void* stream) {
  ResizeB p = {};
  int rc = resize_setup_b("dl3p_resize_bilinear_fwd_bf16", p, x, ldx, y, ldy, N, h, w, C, H, W, (long long)N * H * W);
  if (rc) return rc;
  if (p.v == 8) hipLaunchKernelGGL(resize_fwd_b<8>, dim3(p.nbx * p.nslab), dim3(256), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(resize_fwd_b<4>, dim3(p.nbx * p.nslab), dim3(256), 0, (hipStream_t)stream, p);
  DL3P_CHECK_LAUNCH("dl3p_resize_bilinear_fwd_bf16");
  return DL3P_OK;
}

extern "C" int dl3p_resize_bilinear_bwd_bf16(const void* gy, int ldgy, void* gx, int ldgx, int accumulate, int N, int h,
                                             int w, int C, int H, int W, void* stream) {
  ResizeB p = {};
  int rc = resize_setup_b("dl3p_resize_bilinear_bwd_bf16", p, gy, ldgy, gx, ldgx, N, h, w, C, H, W, (long long)N * h * w);
  if (rc) return rc;
  p.accumulate = accumulate;
  if (p.v == 8) hipLaunchKernelGGL(resize_bwd_b<8>, dim3(p.nbx * p.nslab), dim3(256), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(resize_bwd_b<4>, dim3(p.nbx * p.nslab), dim3(256), 0, (hipStream_t)stream, p);
  DL3P_CHECK_LAUNCH("dl3p_resize_bilinear_bwd_bf16");
  return DL3P_OK;
}

// ------------------------------------------------------------------------------ im2col (RGB stem) and conversions
namespace {
struct Im2colB {
  const bf16* x; int ldx; const float* scale; const float* shift; int act; bf16* col; int ld_col;
  int N, H, W, Cin, Ho, Wo, k, stride, rate, pad_t, pad_l;
};
__global__ __launch_bounds__(256) void im2col_b(Im2colB p) {
  const int k4n = p.ld_col / 4;
  const long long total = (long long)p.N * p.Ho * p.Wo * k4n;
  const int kk = p.k * p.k * p.Cin;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int k4 = (int)(i % k4n);
    const long long m = i / k4n;
    const int ox = (int)(m % p.Wo);
    const long long row = m / p.Wo;
    const int oy = (int)(row % p.Ho), n = (int)(row / p.Ho);
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int e = k4 * 4 + j;
      float a = 0.f;
      if (e < kk) {
        const int tap = e / p.Cin, ci = e - tap * p.Cin;
        const int ky = tap / p.k, kx = tap - ky * p.k;
        const int iy = oy * p.stride - p.pad_t + ky * p.rate, ix = ox * p.stride - p.pad_l + kx * p.rate;
        if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) {
          a = (float)p.x[(((size_t)n * p.H + iy) * p.W + ix) * p.ldx + ci];
          if (p.scale) a = fmaf(a, p.scale[ci], p.shift[ci]);
          a = act_apply(a, p.act);
        }
      }
      v[j] = a;
    }
    st4(p.col + (size_t)m * p.ld_col + k4 * 4, make_float4(v[0], v[1], v[2], v[3]));
  }
}

__global__ __launch_bounds__(256) void f32_to_bf16_k(const float* __restrict__ src, bf16* __restrict__ dst, size_t n) {
  size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
  const size_t stride = (size_t)gridDim.x * 1024;
  for (; i + 3 < n; i += stride) st4(dst + i, ld4(src + i));
  if (blockIdx.x == 0 && threadIdx.x == 0)
    for (size_t j = n & ~(size_t)3; j < n; ++j) dst[j] = (bf16)src[j];
}
__global__ __launch_bounds__(256) void u8_to_bf16_k(const unsigned char* __restrict__ src, bf16* __restrict__ dst, size_t n,
                                                    float div, float sub) {
  const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i + 3 < n) {
    const uchar4 v = *reinterpret_cast<const uchar4*>(src + i);
    st4(dst + i, make_float4((float)v.x / div - sub, (float)v.y / div - sub, (float)v.z / div - sub, (float)v.w / div - sub));
  } else {
    for (size_t j = i; j < n; ++j) dst[j] = (bf16)((float)src[j] / div - sub);
  }
}
// dst[off + n*K + k] = bf16(src[off + k*N + n]) for every (off, K, N) row of `table`: the [N][K] bf16 copies of the
// pointwise / im2col'd kernels that the forward GEMM reads (the data-gradient GEMM reads the plain bf16 mirror)
__global__ __launch_bounds__(256) void transpose_batch_b(const float* src, bf16* dst, const int* table) {
  __shared__ float tile[32][33];
  const int off = table[blockIdx.x * 4], K = table[blockIdx.x * 4 + 1], N = table[blockIdx.x * 4 + 2];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int tk = (K + 31) / 32, tn = (N + 31) / 32;
  for (int tl = blockIdx.y; tl < tk * tn; tl += gridDim.y) {
    const int k0 = (tl / tn) * 32, n0 = (tl % tn) * 32;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = k0 + ty + 8 * i, n = n0 + tx;
      tile[ty + 8 * i][tx] = (k < K && n < N) ? src[off + (size_t)k * N + n] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int n = n0 + ty + 8 * i, k = k0 + tx;
      if (k < K && n < N) dst[off + (size_t)n * K + k] = (bf16)tile[tx][ty + 8 * i];
    }
  }
}

// ---- dense k x k convs and max pooling under the mixed policy (Xception's entry_flow_conv1_2, ResNet50's 3x3 convs and pool1:
// train.py:37-46 applies the policy to every model type).  Same gather forms as conv.hip / pool.hip, bf16 in HBM, fp32 sums.
struct Col2imB {
  const bf16* gcol; int ld_col; bf16* gx; int ldgx; int accumulate;
  int N, H, W, Cin, Ho, Wo, k, stride, rate, pad_t, pad_l;
};
// gx[n, iy, ix, ci] (+)= sum over the taps (ky, kx) and output pixels that read this input pixel of gcol[m][(ky k + kx) Cin + ci]
__global__ __launch_bounds__(256) void col2im_b(Col2imB p) {
  const int c4n = p.Cin / 4;
  const long long total = (long long)p.N * p.H * p.W * c4n;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c4 = (int)(i % c4n);
    long long pix = i / c4n;
    const int ix = (int)(pix % p.W);
    pix /= p.W;
    const int iy = (int)(pix % p.H);
    const int n = (int)(pix / p.H);
    float4 acc = zero4();
    for (int ky = 0; ky < p.k; ++ky) {
      const int ty = iy + p.pad_t - ky * p.rate;
      if (ty < 0 || ty % p.stride) continue;
      const int oy = ty / p.stride;
      if (oy >= p.Ho) continue;
      for (int kx = 0; kx < p.k; ++kx) {
        const int tx = ix + p.pad_l - kx * p.rate;
        if (tx < 0 || tx % p.stride) continue;
        const int ox = tx / p.stride;
        if (ox >= p.Wo) continue;
        const size_t m = ((size_t)n * p.Ho + oy) * p.Wo + ox;
        acc = add4(acc, ld4(p.gcol + m * p.ld_col + (size_t)(ky * p.k + kx) * p.Cin + c4 * 4));
      }
    }
    bf16* o = p.gx + (((size_t)n * p.H + iy) * p.W + ix) * p.ldgx + c4 * 4;
    if (p.accumulate) acc = add4(acc, ld4(o));
    st4(o, acc);
  }
}

struct MaxPoolB {
  const bf16* x; int ldx; const float* scale; const float* shift; int act;
  bf16* y; int ldy;
  const bf16* dy; int lddy;
  unsigned char* arg;
  int accumulate;
  int N, H, W, C, k, stride, pad_t, pad_l, Ho, Wo;
  long long total;
};
// ZeroPadding2D + MaxPooling2D (deeplabv3p_resnet50.py:266-267), one thread per (output pixel, 4 channels); the input carries its
// producer's BatchNorm + activation as a prologue, rounded to bf16 like every consumer-side prologue of this path
__global__ __launch_bounds__(256) void maxpool_fwd_b(MaxPoolB p) {
  const int c4s = p.C / 4;
  for (long long s = (long long)blockIdx.x * 256 + threadIdx.x; s < p.total; s += (long long)gridDim.x * 256) {
    const int c = (int)(s % c4s) * 4;
    long long r = s / c4s;
    const int ox = (int)(r % p.Wo); r /= p.Wo;
    const int oy = (int)(r % p.Ho);
    const int n = (int)(r / p.Ho);
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = zero4();
    if (p.scale) { sc = ld4(p.scale + c); sh = ld4(p.shift + c); }
    const bf16* img = p.x + (size_t)n * p.H * p.W * p.ldx;
    float4 m = make_float4(-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f);
    uchar4 a = make_uchar4(0, 0, 0, 0);
    for (int ky = 0; ky < p.k; ++ky)
      for (int kx = 0; kx < p.k; ++kx) {
        const int iy = oy * p.stride - p.pad_t + ky, ix = ox * p.stride - p.pad_l + kx;
        float4 v = zero4();                                    // ZeroPadding2D: real zeros that take part in the maximum
        if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) {
          v = ld4(img + ((size_t)iy * p.W + ix) * p.ldx + c);
          if (p.scale || p.act != DL3P_ACT_NONE) v = bf16_round4(act_apply4(p.scale ? fma4(v, sc, sh) : v, p.act));
        }
        const unsigned char t = (unsigned char)(ky * p.k + kx);
        if (v.x > m.x) { m.x = v.x; a.x = t; }             // strict: the first maximum in (ky, kx) order wins
        if (v.y > m.y) { m.y = v.y; a.y = t; }
        if (v.z > m.z) { m.z = v.z; a.z = t; }
        if (v.w > m.w) { m.w = v.w; a.w = t; }
      }
    st4(p.y + (((size_t)n * p.Ho + oy) * p.Wo + ox) * p.ldy + c, m);
    if (p.arg) *reinterpret_cast<uchar4*>(p.arg + (((size_t)n * p.Ho + oy) * p.Wo + ox) * p.C + c) = a;
  }
}
// gather form: an input pixel collects dy of every window whose recorded winner it is
__global__ __launch_bounds__(256) void maxpool_bwd_b(MaxPoolB p) {
  const int c4s = p.C / 4;
  for (long long s = (long long)blockIdx.x * 256 + threadIdx.x; s < p.total; s += (long long)gridDim.x * 256) {
    const int c = (int)(s % c4s) * 4;
    long long r = s / c4s;
    const int ix = (int)(r % p.W); r /= p.W;
    const int iy = (int)(r % p.H);
    const int n = (int)(r / p.H);
    float4 g = zero4();
    const int oy_hi = (iy + p.pad_t) / p.stride, ox_hi = (ix + p.pad_l) / p.stride;
    for (int oy = oy_hi; oy >= 0 && oy * p.stride - p.pad_t + p.k > iy; --oy) {
      if (oy >= p.Ho) continue;
      for (int ox = ox_hi; ox >= 0 && ox * p.stride - p.pad_l + p.k > ix; --ox) {
        if (ox >= p.Wo) continue;
        const unsigned char t = (unsigned char)((iy - (oy * p.stride - p.pad_t)) * p.k + ix - (ox * p.stride - p.pad_l));
        const size_t o = ((size_t)n * p.Ho + oy) * p.Wo + ox;
        const uchar4 a = *reinterpret_cast<const uchar4*>(p.arg + o * p.C + c);
        const float4 d = ld4(p.dy + o * p.lddy + c);
        g = make_float4(g.x + (a.x == t ? d.x : 0.f), g.y + (a.y == t ? d.y : 0.f), g.z + (a.z == t ? d.z : 0.f),
                        g.w + (a.w == t ? d.w : 0.f));
      }
    }
    bf16* o = p.y + (((size_t)n * p.H + iy) * p.W + ix) * p.ldy + c;
    if (p.accumulate) g = add4(g, ld4(o));
    st4(o, g);
  }
}
inline unsigned pool_grid_b(long long total) {
  long long b = (total + 255) / 256;
  const long long cap = (long long)DL3P_NUM_CUS * 16;
  return (unsigned)(b < 1 ? 1 : (b > cap ? cap : b));
}
}  // namespace

extern "C" int dl3p_im2col_bf16(const void* x, int ldx, const float* in_scale, const float* in_shift, int in_act,
                                void* col, int ld_col, int N, int H, int W, int Cin, int k, int stride, int rate,
                                int pad_t, int pad_l, int Ho, int Wo, void* stream) {
  DL3P_CHECK_ARG(x && col && ((uintptr_t)col & 7u) == 0 && ld_col % 4 == 0 && ld_col >= k * k * Cin && ldx >= Cin,
                 "dl3p_im2col_bf16: bad layout (ld_col=%d)", ld_col);
  Im2colB p = {};
  p.x = (const bf16*)x; p.ldx = ldx; p.scale = in_scale; p.shift = in_shift; p.act = in_act; p.col = (bf16*)col;
  p.ld_col = ld_col; p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Ho = Ho; p.Wo = Wo; p.k = k; p.stride = stride;
  p.rate = rate; p.pad_t = pad_t; p.pad_l = pad_l;
  long long blocks = ceil_div_ll((long long)N * Ho * Wo * (ld_col / 4), 256);
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(im2col_b, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p);
  DL3P_CHECK_LAUNCH("dl3p_im2col_bf16");
  return DL3P_OK;
}

extern "C" int dl3p_col2im_bf16(const void* gcol, int ld_col, void* gx, int ldgx, int accumulate, int N, int H, int W, int Cin,
                                int k, int stride, int rate, int pad_t, int pad_l, int Ho, int Wo, void* stream) {
  DL3P_CHECK_ARG(gcol && gx && ((uintptr_t)gcol & 7u) == 0 && ((uintptr_t)gx & 7u) == 0 && Cin > 0 && Cin % 4 == 0 && ld_col % 4 == 0 &&
                 ld_col >= k * k * Cin && ldgx % 4 == 0 && ldgx >= Cin && N > 0 && k >= 1 && stride >= 1 && rate >= 1,
                 "dl3p_col2im_bf16: bad layout");
  Col2imB p = {};
  p.gcol = (const bf16*)gcol; p.ld_col = ld_col; p.gx = (bf16*)gx; p.ldgx = ldgx; p.accumulate = accumulate;
  p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Ho = Ho; p.Wo = Wo; p.k = k; p.stride = stride; p.rate = rate;
  p.pad_t = pad_t; p.pad_l = pad_l;
  long long blocks = ceil_div_ll((long long)N * H * W * (Cin / 4), 256);
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(col2im_b, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p);
  DL3P_CHECK_LAUNCH("dl3p_col2im_bf16");
  return DL3P_OK;
}

extern "C" int dl3p_maxpool2d_fwd_bf16(const void* x, int ldx, const float* in_scale, const float* in_shift, int in_act, void* y,
                                       int ldy, uint8_t* argmax, int N, int H, int W, int C, int k, int stride, int pad_t,
                                       int pad_l, int Ho, int Wo, void* stream) {
  DL3P_CHECK_ARG(x && y && ((uintptr_t)x & 7u) == 0 && ((uintptr_t)y & 7u) == 0 && C > 0 && C % 4 == 0 && ldx % 4 == 0 && ldx >= C &&
                 ldy % 4 == 0 && ldy >= C && N > 0 && k >= 1 && k <= 15 && stride >= 1 && Ho > 0 && Wo > 0,
                 "dl3p_maxpool2d_fwd_bf16: bad arguments");
  DL3P_CHECK_ARG(!argmax || (uintptr_t)argmax % 4 == 0, "dl3p_maxpool2d_fwd_bf16: argmax must be 4-byte aligned");
  MaxPoolB p = {};
  p.x = (const bf16*)x; p.ldx = ldx; p.scale = in_scale; p.shift = in_shift; p.act = in_act; p.y = (bf16*)y; p.ldy = ldy;
  p.arg = argmax; p.N = N; p.H = H; p.W = W; p.C = C; p.k = k; p.stride = stride; p.pad_t = pad_t; p.pad_l = pad_l;
  p.Ho = Ho; p.Wo = Wo; p.total = (long long)N * Ho * Wo * (C / 4);
  hipLaunchKernelGGL(maxpool_fwd_b, dim3(pool_grid_b(p.total)), dim3(256), 0, (hipStream_t)stream, p);
  DL3P_CHECK_LAUNCH("dl3p_maxpool2d_fwd_bf16");
  return DL3P_OK;
}

extern "C" int dl3p_maxpool2d_bwd_bf16(const void* dy, int lddy, const uint8_t* argmax, void* gx, int ldgx, int accumulate, int N,
                                       int H, int W, int C, int k, int stride, int pad_t, int pad_l, int Ho, int Wo,
                                       void* stream) {
  DL3P_CHECK_ARG(dy && gx && argmax && ((uintptr_t)dy & 7u) == 0 && ((uintptr_t)gx & 7u) == 0 && (uintptr_t)argmax % 4 == 0 && C > 0 &&
                 C % 4 == 0 && lddy % 4 == 0 && lddy >= C && ldgx % 4 == 0 && ldgx >= C && N > 0 && k >= 1 && stride >= 1 &&
                 Ho > 0 && Wo > 0, "dl3p_maxpool2d_bwd_bf16: bad arguments (the recorded winners of the forward pass are required)");
  MaxPoolB p = {};
  p.dy = (const bf16*)dy; p.lddy = lddy; p.arg = const_cast<uint8_t*>(argmax); p.y = (bf16*)gx; p.ldy = ldgx;
  p.accumulate = accumulate; p.N = N; p.H = H; p.W = W; p.C = C; p.k = k; p.stride = stride; p.pad_t = pad_t;
  p.pad_l = pad_l; p.Ho = Ho; p.Wo = Wo; p.total = (long long)N * H * W * (C / 4);
  hipLaunchKernelGGL(maxpool_bwd_b, dim3(pool_grid_b(p.total)), dim3(256), 0, (hipStream_t)stream, p);
  DL3P_CHECK_LAUNCH("dl3p_maxpool2d_bwd_bf16");
  return DL3P_OK;
}

extern "C" int dl3p_f32_to_bf16(const float* src, void* dst, size_t n, void* stream) {
  DL3P_CHECK_ARG(src && dst && al16(src) && ((uintptr_t)dst & 7u) == 0, "dl3p_f32_to_bf16: bad pointers");
  if (n == 0) return DL3P_OK;
  size_t blocks = (n / 4 + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(f32_to_bf16_k, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, (bf16*)dst, n);
  DL3P_CHECK_LAUNCH("dl3p_f32_to_bf16");
  return DL3P_OK;
}

extern "C" int dl3p_u8_to_bf16(const unsigned char* src, void* dst, size_t n, float divide_by, float subtract,
                               void* stream) {
  DL3P_CHECK_ARG(src && dst && ((uintptr_t)src % 4 == 0) && ((uintptr_t)dst & 7u) == 0, "dl3p_u8_to_bf16: bad pointers");
  if (n == 0) return DL3P_OK;
  const size_t blocks = (n / 4 + 256) / 256;
  hipLaunchKernelGGL(u8_to_bf16_k, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, (bf16*)dst, n,
                     divide_by, subtract);
  DL3P_CHECK_LAUNCH("dl3p_u8_to_bf16");
  return DL3P_OK;
}

extern "C" int dl3p_transpose_batch_bf16(const float* src, void* dst, const int* table, int n_matrices, void* stream) {
  DL3P_CHECK_ARG(src && dst && table && n_matrices > 0, "dl3p_transpose_batch_bf16: bad arguments");
  hipLaunchKernelGGL(transpose_batch_b, dim3(n_matrices, 96), dim3(256), 0, (hipStream_t)stream, src, (bf16*)dst, table);
  DL3P_CHECK_LAUNCH("dl3p_transpose_batch_bf16");
  return DL3P_OK;
}
