// Shared device/host helpers for libdl3p (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/dl3p.h"

#define DL3P_NUM_CUS 256
#define DL3P_NUM_XCDS 8

void dl3p_set_error(const char* fmt, ...);

#define DL3P_CHECK_ARG(cond, ...)            \
  do {                                       \
    if (!(cond)) {                           \
      dl3p_set_error(__VA_ARGS__);           \
      return DL3P_EINVAL;                    \
    }                                        \
  } while (0)

#define DL3P_CHECK_LAUNCH(name)                                              \
  do {                                                                       \
    hipError_t e_ = hipGetLastError();                                       \
    if (e_ != hipSuccess) {                                                  \
      dl3p_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));  \
      return DL3P_ELAUNCH;                                                   \
    }                                                                        \
  } while (0)

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline long long ceil_div_ll(long long a, long long b) { return (a + b - 1) / b; }

// ---------------------------------------------------------------------------------- device
// Activations are evaluated BRANCH-FREE from wave-uniform constants derived from the activation code:
//   t = clamp(v + add, lo, hi) * mul;  out = (hswish) ? v * t : t
// (none: lo=-inf,hi=+inf; relu: 0,+inf; relu6: 0,6; hard-sigmoid / hard-swish: add 3, clamp 0..6, /6).
// A `switch` here gets loop-unswitched inside the heavily unrolled conv bodies and explodes the code
// (43k instructions for one depthwise kernel); selects on a uniform value cost a few SALU ops once.
#define DL3P_INF __builtin_huge_valf()
__device__ __forceinline__ float act_apply(float v, int act) {
  const bool h = act >= DL3P_ACT_HSWISH;
  const float lo = act == DL3P_ACT_NONE ? -DL3P_INF : 0.f;
  const float hi = (act == DL3P_ACT_NONE || act == DL3P_ACT_RELU) ? DL3P_INF : 6.f;
  const float add = h ? 3.f : 0.f;
  const float mul = h ? (1.f / 6.f) : 1.f;
  const float t = fminf(fmaxf(v + add, lo), hi) * mul;
  return act == DL3P_ACT_HSWISH ? v * t : t;
}
// derivative of the activation w.r.t. its (pre-activation) input u (TF conventions: 0 at the kinks)
__device__ __forceinline__ float act_grad(float u, int act) {
  const bool h = act >= DL3P_ACT_HSWISH;
  const float lo = act == DL3P_ACT_NONE ? -DL3P_INF : 0.f;
  const float hi = (act == DL3P_ACT_NONE || act == DL3P_ACT_RELU) ? DL3P_INF : 6.f;
  const float add = h ? 3.f : 0.f;
  const float mul = h ? (1.f / 6.f) : 1.f;
  const float t = u + add;
  const float in = (t > lo && t < hi) ? mul : 0.f;
  const float hs = fminf(fmaxf(t, lo), hi) * mul;
  return act == DL3P_ACT_HSWISH ? hs + u * in : in;
}
__device__ __forceinline__ float4 act_apply4(float4 v, int act) {
  return make_float4(act_apply(v.x, act), act_apply(v.y, act), act_apply(v.z, act), act_apply(v.w, act));
}
__device__ __forceinline__ float4 fma4(float4 a, float4 b, float4 c) {
  return make_float4(fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y), fmaf(a.z, b.z, c.z), fmaf(a.w, b.w, c.w));
}
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
// streaming store (global_store ... nt): the line is not kept in L2 for this kernel's benefit
__device__ __forceinline__ void st4_nt(float* p, float4 v) {
  typedef float f4v __attribute__((ext_vector_type(4)));
  __builtin_nontemporal_store((f4v){v.x, v.y, v.z, v.w}, reinterpret_cast<f4v*>(p));
}

// counter-based keep mask for Dropout: one 64-bit mix per element (splitmix64 finaliser)
__device__ __forceinline__ bool dropout_keep(uint64_t seed, int64_t step, uint64_t idx, float rate) {
  uint64_t z = seed + 0x9E3779B97F4A7C15ull * (uint64_t)(step + 1) + idx * 0xD1B54A32D192ED03ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  float u = (float)(z >> 40) * (1.0f / 16777216.0f);  // [0,1)
  return u >= rate;
}

// XCD-aware work split: the dispatcher deals consecutive workgroups round-robin over the 8 XCDs
// (each with a private 4 MiB L2).  Giving workgroup b the `b % 8`-th contiguous chunk of the
// (image,row,col) work range keeps every image's reads inside one L2.  Speed only, never
// correctness.  Work items [begin,end) of this workgroup's chunk, visited with stride `step`.
struct XcdRange { int begin, end, step; };
__device__ __forceinline__ XcdRange xcd_range(long long total_ll, int bx, int nbx, int lanes, int lane) {
  const int total = (int)total_ll;  // hosts reject work ranges >= 2^31
  int xcd = bx & (DL3P_NUM_XCDS - 1);
  int j = bx >> 3;
  int nbj = nbx >> 3;
  const int chunk = (total + DL3P_NUM_XCDS - 1) / DL3P_NUM_XCDS;
  const int b = chunk * xcd;
  const int e = b + chunk < total ? b + chunk : total;
  XcdRange r;
  r.begin = b + j * lanes + lane;
  r.end = e;
  r.step = nbj * lanes;
  return r;
}

// ---------------------------------------------------------------------------------- shared decomposition
// Row-major [rows][C] tensors are walked by 256-thread workgroups as (px pixel lanes) x (c4s channel
// lanes of 4 floats); channel lanes are fastest so a wave reads >= 128 B contiguous per row.
static inline void pick_lanes(int C, int* c4s, int* px, int* nslab) {
  const int c4 = C / 4;
  int best = 1, best_used = 0;
  for (int d = 1; d <= c4 && d <= 256; ++d) {
    if (c4 % d) continue;
    if (d < 32 && d < c4) continue;  // keep >= 512 B contiguous per pixel (or the whole pixel)
    int used = (256 / d) * d;
    if (used > best_used || (used == best_used && d > best)) { best = d; best_used = used; }
  }
  *c4s = best;
  *px = 256 / best;
  *nslab = c4 / best;
}

// workgroups per channel slab (a multiple of 8 = one share per XCD), ~8 workgroups per CU overall,
// never more than DL3P_MAX_STAT_ROWS (each workgroup emits one partial row)
static inline int pick_nbx(long long total, int px, int nslab, int per_cu = 8) {
  long long chunk = ceil_div_ll(total, DL3P_NUM_XCDS);
  long long need = ceil_div_ll(chunk, px);
  long long target = (DL3P_NUM_CUS * per_cu / nslab) / DL3P_NUM_XCDS;
  if (target < 1) target = 1;
  long long nbj = need < target ? need : target;
  if (nbj < 1) nbj = 1;
  if (nbj * DL3P_NUM_XCDS > DL3P_MAX_STAT_ROWS) nbj = DL3P_MAX_STAT_ROWS / DL3P_NUM_XCDS;
  return (int)nbj * DL3P_NUM_XCDS;
}

__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 mul4(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }

// per-workgroup reduction over the pixel lanes of NV float4 values held per thread; writes
// out_row[v][C] (channels of this workgroup's slab) for v in [0,NV).  All 256 threads must call.
template <int NV>
__device__ __forceinline__ void block_reduce_store(const float4 (&vals)[NV], bool active, int pl, int cl,
                                                   int c4s, int px, int cbase4, int C, float* out_row) {
  __shared__ float4 sm[256];
  for (int v = 0; v < NV; ++v) {
    __syncthreads();
    if (active) sm[pl * c4s + cl] = vals[v];
    __syncthreads();
    if ((int)threadIdx.x < c4s) {
      float4 a = sm[threadIdx.x];
      for (int q = 1; q < px; ++q) a = add4(a, sm[q * c4s + threadIdx.x]);
      st4(out_row + (size_t)v * C + (size_t)(cbase4 + threadIdx.x) * 4, a);
    }
  }
}

// pw_tiny.hip: pointwise convolutions on M <= 64 rows (behind a global pooling)
bool dl3p_pw_tiny_applies(int M);
void dl3p_pw_tiny_nt(const float* a, int lda, const float* scale, const float* shift, int act, const float* bt, int ldb,
                     const float* bias, float* y, int ldy, int accumulate, float* partials, int M, int K, int N,
                     hipStream_t st);
void dl3p_pw_tiny_wgrad(const float* x, int ldx, const float* scale, const float* shift, int act, const float* dy,
                        int lddy, float* gw, float* gb, int M, int K, int N, hipStream_t st);

// depthwise plan knobs moved by dl3p_set_option (defined in dwconv.hip): 0 = automatic
extern int dl3p_dw_force_per_cu, dl3p_dw_force_want, dl3p_dw_force_maxth, dl3p_dw_force_tw, dl3p_dw_use_table;

int dl3p_reduce_rows_impl(const float* partials, int rows, size_t n, float* out, int accumulate, hipStream_t st);
int dl3p_reduce_rows_strided_impl(const float* partials, int rows, size_t row_stride, size_t n, float* out,
                                  int accumulate, hipStream_t st);

// ---------------------------------------------------------------------------------- timed launches
// dl3p_probe_arm() leaves a (start, stop) event pair for the next launch that goes through dl3p_launch
extern thread_local hipEvent_t dl3p_probe_start, dl3p_probe_stop;
template <typename K, typename... A>
static inline void dl3p_launch(K kernel, dim3 grid, dim3 block, size_t shmem, hipStream_t st, A... args) {
  if (dl3p_probe_start) {
    hipExtLaunchKernelGGL(kernel, grid, block, shmem, st, dl3p_probe_start, dl3p_probe_stop, 0, args...);
    dl3p_probe_start = nullptr;
    dl3p_probe_stop = nullptr;
  } else {
    hipLaunchKernelGGL(kernel, grid, block, shmem, st, args...);
  }
}
