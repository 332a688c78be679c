// BatchNorm backward of a SMALL tensor in one launch ("remove launches, not microseconds").
//
// The three-launch chain  dl3p_bn_bwd_reduce -> dl3p_bn_bwd_finalize -> dl3p_bn_bwd_apply  (reference: the gradient of
// CustomBatchNormalization, /root/reference deeplabv3p/models/layers.py:63-70, as Keras' autodiff forms it) costs a
// 5-8 MB tensor ~5 us per launch of pure latency, three times, and reads (g, z) twice.  On the 64 x 128 .. 512 x 1024 maps of
// BASELINE.json configs[4] (and the 33 x 33 maps of configs[1..2]) the whole tensor fits the registers of ONE resident grid:
//
//   phase 1  every thread loads its <= NV row vectors of g and z (kept raw), forms g' = g * act'(z*scale+shift) and the
//            (sum g', sum g' * xhat) of its rows; the workgroup leaves one partial row for its channel slab
//   barrier  among the workgroups of the slab (they are all resident: grid <= 2 workgroups per CU by construction)
//   phase 2  every workgroup of the slab reduces the slab's partial rows (few: a slab is 8 channel lanes wide, so a slab has
//            512 / nslab workgroups) in a fixed order in double, forms (c0, c1, c2) = (gamma*invstd, sum/count, sumx/count);
//            workgroup 0 of the slab also writes dgamma, dbeta and the coefficient triple
//   phase 3  dz = c0 * (g' - c1 - xhat * c2) from the registers of phase 1 (g and z are not read again)
//
// Deterministic: fixed partition, fixed summation order, no floating-point atomics (the barrier counts arrivals with an integer).
// The barrier spins at most BARRIER_SPINS times and then raises the error word of the workspace instead of hanging.
#include "bf16.h"
#include <stdlib.h>

namespace {

constexpr int MAX_GRID = 512;          // 2 workgroups of 256 threads per CU: all resident at once (launch bound below)
constexpr int MAX_NV = 16;             // row vectors a thread keeps (g and z: 2 x 4 registers each; 8 with 8-channel bf16 lanes, whose
                                       // 16-vector instantiation does not fit 256 registers)
constexpr int SLAB_LANES = 8;          // channel lanes per slab: 128 B per pixel row, 32 pixel lanes per workgroup
constexpr unsigned BARRIER_SPINS = 2000000u;     // x >= 1.2 us: seconds
constexpr int WS_COUNTERS = 2 * MAX_GRID + 16;      // arrive / depart per slab, then the error word
constexpr int WS_ERR = 2 * MAX_GRID;

struct FusedP {
  const void* g; int ldg;
  const void* z; int ldz;
  const float* scale; const float* shift; int act;
  const float* mean; const float* invstd; const float* gamma;
  float* dgamma; float* dbeta; float* coef;
  void* dz; int lddz;
  int M, C;
  int lanes, cs, px, nslab, nbx;
  double count;
  unsigned* counters; float* rows;
  int dbg;            // DL3P_BNF_DBG (timing experiments only): 1 no barrier
};

// a row vector as loaded (packed words): the compiler must not keep the unpacked floats of phase 1 alive across the barrier
// (8-channel bf16 lanes: 4 words packed against 8 unpacked -- 16 kept vectors of g and z would not fit the register file)
template <typename T, int V> struct RawVec { uint32_t w[V * (int)sizeof(T) / 4]; };
template <typename T, int V> __device__ __forceinline__ RawVec<T, V> ld_raw(const T* p) {
  constexpr int W = V * (int)sizeof(T) / 4;
  RawVec<T, V> r;
  if (W == 4) {
    const uint4 v = *reinterpret_cast<const uint4*>(p);
    r.w[0] = v.x; r.w[1] = v.y; r.w[2 % W] = v.z; r.w[3 % W] = v.w;
  } else {
    const uint2 v = *reinterpret_cast<const uint2*>(p);
    r.w[0] = v.x; r.w[1] = v.y;
  }
  return r;
}
template <typename T, int V> __device__ __forceinline__ RawVec<T, V> raw_zero() {
  RawVec<T, V> r;
#pragma unroll
  for (int i = 0; i < V * (int)sizeof(T) / 4; ++i) r.w[i] = 0u;
  return r;
}
template <typename T, int V> __device__ __forceinline__ float raw_at(const RawVec<T, V>& r, int j);
template <> __device__ __forceinline__ float raw_at<bf16, 8>(const RawVec<bf16, 8>& r, int j) {
  return __uint_as_float((j & 1) ? (r.w[j >> 1] & 0xffff0000u) : (r.w[j >> 1] << 16));
}
template <> __device__ __forceinline__ float raw_at<bf16, 4>(const RawVec<bf16, 4>& r, int j) {
  return __uint_as_float((j & 1) ? (r.w[j >> 1] & 0xffff0000u) : (r.w[j >> 1] << 16));
}
template <> __device__ __forceinline__ float raw_at<float, 4>(const RawVec<float, 4>& r, int j) { return __uint_as_float(r.w[j]); }
template <typename T, int V> __device__ __forceinline__ void raw_pin(RawVec<T, V>& r) {
#pragma unroll
  for (int i = 0; i < V * (int)sizeof(T) / 4; ++i) asm volatile("" : "+v"(r.w[i]));
}
template <int V> __device__ __forceinline__ void st_out(bf16* p, const fvec<V>& o) { stv<V>(p, o); }
template <int V> __device__ __forceinline__ void st_out(float* p, const fvec<V>& o) { stv<V>(p, o); }

// all workgroups of one slab: integer arrivals, bounded spin.  One agent-scope release in front of the arrival and one acquire
// behind the wait; the polls themselves are relaxed (an acquire per poll invalidates the caches under every other workgroup's loads:
// measured 50-90 us per launch against 8-12)
__device__ __forceinline__ void slab_barrier(unsigned* arrive, unsigned n, unsigned* err) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    while (__hip_atomic_load(arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < n) {
      // every waiting workgroup polls the same word: back off (0.3 -> 1.2 us) so that the polls do not queue in front of the arrivals
      if (spins < 2) __builtin_amdgcn_s_sleep(8);
      else if (spins < 6) __builtin_amdgcn_s_sleep(16);
      else __builtin_amdgcn_s_sleep(32);
      if (++spins > BARRIER_SPINS) {
        __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
}

template <typename T, int V, int NV>
__global__ __launch_bounds__(256, 2) void bn_bwd_fused_kernel(FusedP p) {
  __shared__ float sm[256 * V];
  __shared__ double smd[256];
  __shared__ float cf[3][SLAB_LANES * V];
  const int slab = blockIdx.x / p.nbx, bx = blockIdx.x - slab * p.nbx;
  const int pl = threadIdx.x / p.cs, cl = threadIdx.x - pl * p.cs;
  const int lane = slab * p.cs + cl;
  const bool active = pl < p.px && lane < p.lanes;
  const int c = lane * V;
  const int width = p.cs * V;                    // channels of a slab (the last slab may hold fewer real ones)
  const T* gp = static_cast<const T*>(p.g);
  const T* zp = static_cast<const T*>(p.z);
  const int stride = p.nbx * p.px;

  // ---- phase 1: rows into registers, per-thread sums
  RawVec<T, V> g[NV], z[NV];
  fvec<V> sc, sh, mu, is;
  if (active) {
    sc = ldv_f32_or<V>(p.scale, c, 1.f); sh = ldv_f32_or<V>(p.shift, c, 0.f);
    mu = ldv_f32_or<V>(p.mean, c, 0.f); is = ldv_f32_or<V>(p.invstd, c, 1.f);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int m = bx * p.px + pl + i * stride;
      if (m < p.M) {
        g[i] = ld_raw<T, V>(gp + (size_t)m * p.ldg + c);
        z[i] = ld_raw<T, V>(zp + (size_t)m * p.ldz + c);
      } else {
        g[i] = raw_zero<T, V>();
        z[i] = raw_zero<T, V>();
      }
    }
  } else {
    sc = sh = mu = is = fzero<V>();
#pragma unroll
    for (int i = 0; i < NV; ++i) { g[i] = raw_zero<T, V>(); z[i] = raw_zero<T, V>(); }
  }
  fvec<V> acc[2] = {fzero<V>(), fzero<V>()};
#pragma unroll
  for (int i = 0; i < NV; ++i) {
#pragma unroll
    for (int j = 0; j < V; ++j) {
      const float zz = raw_at<T, V>(z[i], j);
      const float d = raw_at<T, V>(g[i], j) * act_grad(fmaf(zz, sc.v[j], sh.v[j]), p.act);      // rows past M hold g = 0
      acc[0].v[j] += d;
      acc[1].v[j] = fmaf(d, (zz - mu.v[j]) * is.v[j], acc[1].v[j]);
    }
  }
  // one partial row [2][width] per workgroup, rows of a slab contiguous
  float* row = p.rows + ((size_t)(slab * p.nbx + bx) * 2) * width;
  {
    const int P = 256 / width;                   // threads per channel (width <= 64: >= 4), each over every P-th pixel lane
    const int e = threadIdx.x % width, part = threadIdx.x / width;
    for (int v = 0; v < 2; ++v) {
      __syncthreads();
      if (pl < p.px) {
#pragma unroll
        for (int j = 0; j < V; ++j) sm[(pl * p.cs + cl) * V + j] = active ? acc[v].v[j] : 0.f;
      }
      __syncthreads();
      float a = 0.f;
      if (part < P)
        for (int q = part; q < p.px; q += P) a += sm[q * width + e];
      __syncthreads();
      if (part < P) sm[part * width + e] = a;
      __syncthreads();
      if ((int)threadIdx.x < width) {
        float t = sm[threadIdx.x];
        for (int q = 1; q < P; ++q) t += sm[q * width + threadIdx.x];
        row[v * width + threadIdx.x] = t;
      }
    }
  }

  unsigned* arrive = p.counters + 2 * slab;
  unsigned* depart = arrive + 1;
  if (!(p.dbg & 1)) slab_barrier(arrive, (unsigned)p.nbx, p.counters + WS_ERR);

  // ---- phase 2: the slab's sums (fixed order, double), coefficients
  {
    const int ne = 2 * width;                    // values per partial row (<= 128)
    const int Q = 256 / ne;                      // row groups
    const int e = threadIdx.x % ne, q = threadIdx.x / ne;
    double a = 0.0, b = 0.0;
    if (q < Q) {
      const float* base = p.rows + ((size_t)slab * p.nbx * 2) * width + e;
      int r = q;
      for (; r + Q < p.nbx; r += 2 * Q) {
        a += (double)base[(size_t)r * ne];
        b += (double)base[(size_t)(r + Q) * ne];
      }
      if (r < p.nbx) a += (double)base[(size_t)r * ne];
    }
    smd[threadIdx.x] = a + b;
    __syncthreads();
    if ((int)threadIdx.x < width) {
      double s = 0.0, sx = 0.0;
      for (int qq = 0; qq < Q; ++qq) {
        s += smd[qq * ne + threadIdx.x];
        sx += smd[qq * ne + width + threadIdx.x];
      }
      const int cc = slab * width + threadIdx.x;
      if (cc < p.C) {
        const float c0 = p.gamma[cc] * p.invstd[cc];
        const float c1 = (float)(s / p.count), c2 = (float)(sx / p.count);
        cf[0][threadIdx.x] = c0; cf[1][threadIdx.x] = c1; cf[2][threadIdx.x] = c2;
        if (bx == 0) {
          if (p.dgamma) p.dgamma[cc] = (float)sx;
          if (p.dbeta) p.dbeta[cc] = (float)s;
          if (p.coef) { p.coef[cc] = c0; p.coef[p.C + cc] = c1; p.coef[2 * p.C + cc] = c2; }
        }
      } else {
        cf[0][threadIdx.x] = 0.f; cf[1][threadIdx.x] = 0.f; cf[2][threadIdx.x] = 0.f;
      }
    }
    __syncthreads();
  }
  // the last workgroup to leave re-arms the slab's counters for the next launch (stream order makes that visible)
  if (threadIdx.x == 0 && !(p.dbg & 2)) {
    const unsigned old = __hip_atomic_fetch_add(depart, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old == (unsigned)p.nbx - 1u) {
      __hip_atomic_store(arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(depart, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }

  // ---- phase 3: dz from the registers
  if (!active) return;
  fvec<V> c0, c1, c2;
#pragma unroll
  for (int j = 0; j < V; ++j) { c0.v[j] = cf[0][cl * V + j]; c1.v[j] = cf[1][cl * V + j]; c2.v[j] = cf[2][cl * V + j]; }
  T* op = static_cast<T*>(p.dz);
#pragma unroll
  for (int i = 0; i < NV; ++i) { raw_pin<T, V>(g[i]); raw_pin<T, V>(z[i]); }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int m = bx * p.px + pl + i * stride;
    if (m < p.M) {
      fvec<V> o;
#pragma unroll
      for (int j = 0; j < V; ++j) {
        const float zz = raw_at<T, V>(z[i], j);
        const float d = raw_at<T, V>(g[i], j) * act_grad(fmaf(zz, sc.v[j], sh.v[j]), p.act);
        o.v[j] = c0.v[j] * (d - c1.v[j] - (zz - mu.v[j]) * is.v[j] * c2.v[j]);
      }
      st_out<V>(op + (size_t)m * p.lddz + c, o);
    }
  }
}

struct Split { int V, lanes, cs, px, nslab, nbx, nv; };

// esize 2: bf16 tensors (8-channel lanes when everything is 16-byte aligned, else 4); esize 4: float (4-channel lanes)
bool plan(int M, int C, int esize, bool vec8_ok, Split* s) {
  if (M <= 0 || C <= 0 || C % 4) return false;
  s->V = (esize == 2 && vec8_ok && C % 8 == 0) ? 8 : 4;
  s->lanes = C / s->V;
  s->cs = s->lanes < SLAB_LANES ? s->lanes : SLAB_LANES;
  s->px = 256 / s->cs;
  s->nslab = ceil_div(s->lanes, s->cs);
  if (s->nslab > MAX_GRID) return false;
  // as few workgroups per slab as the registers allow (8 row vectors per thread): after the barrier EVERY workgroup of the slab reads
  // all of the slab's partial rows, nbx^2 rows in total
  const int cap = MAX_GRID / s->nslab;
  const int want = ceil_div(M, s->px * 8);
  s->nbx = want < cap ? want : cap;
  s->nv = ceil_div(M, s->px * s->nbx);
  return s->nv <= (s->V == 8 ? MAX_NV / 2 : MAX_NV);
}

template <typename T, int V>
void launch_nv(const FusedP& p, int nv, hipStream_t st) {
  const dim3 grid(p.nbx * p.nslab), block(256);
  if (nv <= 2) dl3p_launch(bn_bwd_fused_kernel<T, V, 2>, grid, block, 0, st, p);
  else if (nv <= 4) dl3p_launch(bn_bwd_fused_kernel<T, V, 4>, grid, block, 0, st, p);
  else if (nv <= 8 || V == 8) dl3p_launch(bn_bwd_fused_kernel<T, V, 8>, grid, block, 0, st, p);
  else dl3p_launch(bn_bwd_fused_kernel<T, V, (V == 8 ? 8 : 16)>, grid, block, 0, st, p);
}

int launch(const char* fn, int esize, const void* g, int ldg, const void* z, int ldz, const float* scale, const float* shift, int act,
           const float* mean, const float* invstd, const float* gamma, float* dgamma, float* dbeta, float* coef, void* dz, int lddz,
           int M, int C, void* workspace, size_t workspace_bytes, void* stream) {
  DL3P_CHECK_ARG(g && z && dz && mean && invstd && gamma && workspace, "%s: null pointer", fn);
  DL3P_CHECK_ARG(workspace_bytes >= dl3p_bn_bwd_fused_workspace() && aligned16(workspace), "%s: workspace of %zu bytes (need %zu, 16-byte aligned)",
                 fn, workspace_bytes, dl3p_bn_bwd_fused_workspace());
  DL3P_CHECK_ARG(C > 0 && C % 4 == 0 && ldg % 4 == 0 && ldz % 4 == 0 && lddz % 4 == 0 && ldg >= C && ldz >= C && lddz >= C,
                 "%s: bad layout (C=%d ld %d %d %d)", fn, C, ldg, ldz, lddz);
  const uintptr_t amask = esize == 2 ? 7u : 15u;
  DL3P_CHECK_ARG(((uintptr_t)g & amask) == 0 && ((uintptr_t)z & amask) == 0 && ((uintptr_t)dz & amask) == 0, "%s: misaligned tensor", fn);
  const bool v8 = esize == 2 && C % 8 == 0 && ldg % 8 == 0 && ldz % 8 == 0 && lddz % 8 == 0 && aligned16(g) && aligned16(z) && aligned16(dz);
  Split s;
  DL3P_CHECK_ARG(plan(M, C, esize, v8, &s), "%s: %d x %d does not fit one resident grid (dl3p_bn_bwd_fused_supported)", fn, M, C);
  FusedP p = {};
  p.g = g; p.ldg = ldg; p.z = z; p.ldz = ldz; p.scale = scale; p.shift = shift; p.act = act; p.mean = mean; p.invstd = invstd;
  p.gamma = gamma; p.dgamma = dgamma; p.dbeta = dbeta; p.coef = coef; p.dz = dz; p.lddz = lddz; p.M = M; p.C = C;
  p.lanes = s.lanes; p.cs = s.cs; p.px = s.px; p.nslab = s.nslab; p.nbx = s.nbx; p.count = (double)M;
  p.counters = static_cast<unsigned*>(workspace);
  p.rows = reinterpret_cast<float*>(static_cast<unsigned*>(workspace) + WS_COUNTERS);
  static const int dbg = getenv("DL3P_BNF_DBG") ? atoi(getenv("DL3P_BNF_DBG")) : 0;
  p.dbg = dbg;
  hipStream_t st = (hipStream_t)stream;
  if (esize == 2) {
    if (s.V == 8) launch_nv<bf16, 8>(p, s.nv, st);
    else launch_nv<bf16, 4>(p, s.nv, st);
  } else {
    launch_nv<float, 4>(p, s.nv, st);
  }
  DL3P_CHECK_LAUNCH(fn);
  return DL3P_OK;
}

}  // namespace

extern "C" size_t dl3p_bn_bwd_fused_workspace(void) {
  // counters + error word, then MAX_GRID partial rows of 2 x (8 lanes x 8 channels) floats
  return sizeof(unsigned) * WS_COUNTERS + sizeof(float) * (size_t)MAX_GRID * 2 * SLAB_LANES * 8;
}

extern "C" int dl3p_bn_bwd_fused_supported(int M, int C, int esize, int ldg, int ldz, int lddz) {
  if (esize != 2 && esize != 4) return 0;
  Split s;
  const bool v8 = esize == 2 && C % 8 == 0 && ldg % 8 == 0 && ldz % 8 == 0 && lddz % 8 == 0;
  return plan(M, C, esize, v8, &s) ? 1 : 0;
}

extern "C" int dl3p_bn_bwd_fused_error(const void* workspace, void* stream) {
  unsigned e = 0;
  if (!workspace) return -1;
  if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return -1;
  if (hipMemcpy(&e, static_cast<const unsigned*>(workspace) + WS_ERR, sizeof(e), hipMemcpyDeviceToHost) != hipSuccess) return -1;
  return (int)e;
}

extern "C" int dl3p_bn_bwd_fused(const float* g, int ldg, const float* z, int ldz, const float* scale, const float* shift, int act,
                                 const float* save_mean, const float* save_invstd, const float* gamma, float* dgamma, float* dbeta,
                                 float* coef, float* dz, int lddz, int M, int C, void* workspace, size_t workspace_bytes, void* stream) {
  return launch("dl3p_bn_bwd_fused", 4, g, ldg, z, ldz, scale, shift, act, save_mean, save_invstd, gamma, dgamma, dbeta, coef, dz, lddz, M,
                C, workspace, workspace_bytes, stream);
}

extern "C" int dl3p_bn_bwd_fused_bf16(const void* g, int ldg, const void* z, int ldz, const float* scale, const float* shift, int act,
                                      const float* save_mean, const float* save_invstd, const float* gamma, float* dgamma, float* dbeta,
                                      float* coef, void* dz, int lddz, int M, int C, void* workspace, size_t workspace_bytes,
                                      void* stream) {
  return launch("dl3p_bn_bwd_fused_bf16", 2, g, ldg, z, ldz, scale, shift, act, save_mean, save_invstd, gamma, dgamma, dbeta, coef, dz,
                lddz, M, C, workspace, workspace_bytes, stream);
}
