// BatchNorm statistics/backward, elementwise materialisation (residual Add, Dropout), global pooling,
// SGD and small utilities for gfx950.  All HBM-bound streaming kernels: 16-B vector accesses with
// channel lanes fastest (common.h pick_lanes), persistent workgroups for the reductions so each
// emits ONE deterministic partial row (no atomics).
//
// Reference call sites: CustomBatchNormalization layers.py:63-70; ReLU/Add/Dropout layers.py:98,161,
// deeplabv3p_mobilenetv2.py:70; AveragePooling2D layers.py:132; SGD common/model_utils.py:124.
#include "common.h"
#include <type_traits>
#include <string.h>

static thread_local char g_err[512] = "";
void dl3p_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* dl3p_last_error_string(void) { return g_err; }
extern "C" int dl3p_version(void) { return DL3P_VERSION; }
extern "C" int dl3p_device_cus(void) { return DL3P_NUM_CUS; }

// ------------------------------------------------------------------------------ measurement hook
thread_local hipEvent_t dl3p_probe_start = nullptr, dl3p_probe_stop = nullptr;
#define DL3P_PROBE_SLOTS 4096
static hipEvent_t g_probe_ev[DL3P_PROBE_SLOTS][2];
extern "C" int dl3p_probe_arm(int slot) {
  DL3P_CHECK_ARG(slot >= 0 && slot < DL3P_PROBE_SLOTS, "dl3p_probe_arm: slot %d out of range", slot);
  for (int i = 0; i < 2; ++i)
    if (!g_probe_ev[slot][i] && hipEventCreate(&g_probe_ev[slot][i]) != hipSuccess) {
      dl3p_set_error("dl3p_probe_arm: hipEventCreate failed");
      return DL3P_ELAUNCH;
    }
  dl3p_probe_start = g_probe_ev[slot][0];
  dl3p_probe_stop = g_probe_ev[slot][1];
  return DL3P_OK;
}
extern "C" int dl3p_probe_read(int slot, float* ms) {
  DL3P_CHECK_ARG(slot >= 0 && slot < DL3P_PROBE_SLOTS && ms && g_probe_ev[slot][0], "dl3p_probe_read: slot %d not armed", slot);
  if (hipEventSynchronize(g_probe_ev[slot][1]) != hipSuccess ||
      hipEventElapsedTime(ms, g_probe_ev[slot][0], g_probe_ev[slot][1]) != hipSuccess) {
    dl3p_set_error("dl3p_probe_read: event query failed (was a depthwise forward launched after arming?)");
    return DL3P_ELAUNCH;
  }
  return DL3P_OK;
}

// ------------------------------------------------------------------------------ row reducer
// out[i] (+)= sum_r partials[r][i]; double accumulation in a fixed order (deterministic)
// EL consecutive elements x RL row lanes per workgroup; each thread keeps 4 independent double
// accumulators so that 4 row loads are in flight, then the row lanes are combined through LDS.
template <int EL, int RL>
__global__ __launch_bounds__(EL * RL) void reduce_rows_kernel(const float* __restrict__ partials, int rows,
                                                              size_t row_stride, size_t n, float* __restrict__ out,
                                                              int accumulate) {
  __shared__ double sm[RL][EL];
  const int ex = threadIdx.x % EL, ry = threadIdx.x / EL;
  const size_t i = (size_t)blockIdx.x * EL + ex;
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  if (i < n) {
    int r = ry;
    for (; r + 3 * RL < rows; r += 4 * RL) {
      a0 += (double)partials[(size_t)r * row_stride + i];
      a1 += (double)partials[(size_t)(r + RL) * row_stride + i];
      a2 += (double)partials[(size_t)(r + 2 * RL) * row_stride + i];
      a3 += (double)partials[(size_t)(r + 3 * RL) * row_stride + i];
    }
    for (; r < rows; r += RL) a0 += (double)partials[(size_t)r * row_stride + i];
  }
  sm[ry][ex] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (ry == 0 && i < n) {
    double acc = 0.0;
#pragma unroll 8
    for (int q = 0; q < RL; ++q) acc += sm[q][ex];
    out[i] = (float)(accumulate ? acc + (double)out[i] : acc);
  }
}

int dl3p_reduce_rows_strided_impl(const float* partials, int rows, size_t row_stride, size_t n, float* out,
                                  int accumulate, hipStream_t st) {
  if (n == 0) return DL3P_OK;
  if (n >= 64 * 1024 || rows <= 16) {
    hipLaunchKernelGGL((reduce_rows_kernel<64, 4>), dim3((unsigned)((n + 63) / 64)), dim3(256), 0, st, partials, rows,
                       row_stride, n, out, accumulate);
  } else {
    hipLaunchKernelGGL((reduce_rows_kernel<16, 64>), dim3((unsigned)((n + 15) / 16)), dim3(1024), 0, st, partials,
                       rows, row_stride, n, out, accumulate);
  }
  DL3P_CHECK_LAUNCH("dl3p_reduce_rows");
  return DL3P_OK;
}
int dl3p_reduce_rows_impl(const float* partials, int rows, size_t n, float* out, int accumulate, hipStream_t st) {
  return dl3p_reduce_rows_strided_impl(partials, rows, n, n, out, accumulate, st);
}
extern "C" int dl3p_reduce_rows(const float* partials, int rows, size_t n, float* out, int accumulate, void* stream) {
  DL3P_CHECK_ARG(partials && out && rows >= 0, "dl3p_reduce_rows: bad arguments");
  return dl3p_reduce_rows_impl(partials, rows, n, out, accumulate, (hipStream_t)stream);
}

// Every weight gradient of a step in TWO launches instead of one per layer: the wgrad kernels leave their slabs
// (dl3p_*_bwd_weight_slabs), and each job (slab address, rows, n, destination in the flat gradient buffer) is reduced with
// exactly the arithmetic of reduce_rows_kernel<64, 4> / <16, 64> -- the same variant dl3p_reduce_rows would have picked
// for it, so the sums are bit-identical to the per-layer path.  blockmap[b] = (job, block within the job).
struct ReduceJob { const float* src; float* dst; int rows; int n; };
// The arithmetic of reduce_rows_kernel<64, 4> on 256 consecutive elements per workgroup (four 64-element chunks, one per
// row-lane group in the final add): a 64-element workgroup of a 7-slab job reads 1.8 KB and the launch becomes a stream of
// 400 000 workgroups (Xception: 730 us for 0.7 GB of slabs); here a thread has 4 chunks x 4 rows in flight.  Bit-identical sums.
#define DL3P_RB_WIDE 4
// Round 5: the workgroups are PERSISTENT (a grid of a few per CU walks the block list) and read 16 bytes per lane.  A block is 14 KB for
// a 14-slab job: as one short-lived workgroup each (dispatch, two dependent descriptor loads, one round of row loads, LDS, store) the
// launch streamed Xception's 2.2 GB of slabs at 3.3 TB/s; the descriptors of the next block are now requested while the rows of this
// one are in flight.  Per element the rows still go to the same four accumulators in the same order: bit-identical sums.
#define DL3P_RB_NV 2          // 256-element sub-blocks per block: 16-byte loads in flight per thread and row = NV
__global__ __launch_bounds__(256) void reduce_rows_batched_wide_kernel(const ReduceJob* __restrict__ jobs,
                                                                       const int2* __restrict__ blockmap, int nblocks) {
  constexpr int EL = 64, RL = 4, CH = DL3P_RB_WIDE, NV = DL3P_RB_NV, SUB = EL * CH;
  __shared__ double sm[NV * CH * RL * EL];                 // [RL][NV * 256 elements] (vector path) / [CH][RL][EL] (scalar path)
  const int ex = threadIdx.x % EL, ry = threadIdx.x / EL;
  int2 bm = blockmap[blockIdx.x];
  ReduceJob jb = jobs[bm.x];
  for (int blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
    const float* __restrict__ partials = jb.src;
    float* dst = jb.dst;
    const size_t n = (size_t)jb.n;
    const int rows = jb.rows;
    const int by = bm.y;
    const int nxt = min(blk + (int)gridDim.x, nblocks - 1);
    if ((n & 3) == 0 && ((uintptr_t)partials & 15) == 0) {
      // lane ex of row-lane group ry owns the four CONSECUTIVE elements 4 ex .. 4 ex + 3 of each 256-element sub-block (a wave reads
      // 1 KB of a row per instruction)
      size_t ec[NV];
#pragma unroll
      for (int c = 0; c < NV; ++c) {
        const size_t e0 = ((size_t)by * NV + c) * SUB + 4 * ex;
        ec[c] = e0 < n ? e0 : n - 4;                        // clamped: loads stay in the job (n % 4 == 0)
      }
      double a[NV][4][4];
#pragma unroll
      for (int c = 0; c < NV; ++c)
#pragma unroll
        for (int k = 0; k < 4; ++k) a[c][k][0] = a[c][k][1] = a[c][k][2] = a[c][k][3] = 0.0;
      int r = ry;
      bool first = true;
      for (; r + 3 * RL < rows; r += 4 * RL) {
        float4 v[NV][4];
#pragma unroll
        for (int c = 0; c < NV; ++c)
#pragma unroll
          for (int u = 0; u < 4; ++u) v[c][u] = *reinterpret_cast<const float4*>(partials + (size_t)(r + u * RL) * n + ec[c]);
        if (first) { bm = blockmap[nxt]; jb = jobs[bm.x]; first = false; }
#pragma unroll
        for (int c = 0; c < NV; ++c)
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            a[c][0][u] += (double)v[c][u].x; a[c][1][u] += (double)v[c][u].y;
            a[c][2][u] += (double)v[c][u].z; a[c][3][u] += (double)v[c][u].w;
          }
      }
      {
        // the <= 3 remaining rows of this row lane: all requested before the first is added (same accumulator, same order; clamped,
        // never conditional loads)
        float4 v[NV][3];
#pragma unroll
        for (int c = 0; c < NV; ++c)
#pragma unroll
          for (int u = 0; u < 3; ++u) v[c][u] = *reinterpret_cast<const float4*>(partials + (size_t)min(r + u * RL, rows - 1) * n + ec[c]);
        if (first) { bm = blockmap[nxt]; jb = jobs[bm.x]; }
#pragma unroll
        for (int c = 0; c < NV; ++c)
#pragma unroll
          for (int u = 0; u < 3; ++u) {
            const bool ok = r + u * RL < rows;                // (+0.0 for a skipped row: the accumulators are never -0.0)
            a[c][0][0] += ok ? (double)v[c][u].x : 0.0; a[c][1][0] += ok ? (double)v[c][u].y : 0.0;
            a[c][2][0] += ok ? (double)v[c][u].z : 0.0; a[c][3][0] += ok ? (double)v[c][u].w : 0.0;
          }
      }
#pragma unroll
      for (int c = 0; c < NV; ++c)
#pragma unroll
        for (int k = 0; k < 4; ++k) sm[(ry * NV + c) * SUB + 4 * ex + k] = (a[c][k][0] + a[c][k][1]) + (a[c][k][2] + a[c][k][3]);
      __syncthreads();
#pragma unroll
      for (int c = 0; c < NV; ++c) {
        const size_t i = ((size_t)by * NV + c) * SUB + threadIdx.x;
        if (i < n) {
          double acc = 0.0;
#pragma unroll
          for (int q = 0; q < RL; ++q) acc += sm[(q * NV + c) * SUB + threadIdx.x];
          dst[i] = (float)acc;
        }
      }
      __syncthreads();
      continue;
    }
    bm = blockmap[nxt];
    jb = jobs[bm.x];
    for (int c2 = 0; c2 < NV; ++c2) {
      const size_t i0 = ((size_t)by * NV + c2) * SUB + ex;
      if (i0 - ex >= n) break;
      size_t idx[CH];
#pragma unroll
      for (int c = 0; c < CH; ++c) idx[c] = i0 + (size_t)c * EL < n ? i0 + (size_t)c * EL : n - 1;     // clamped: loads stay in the job
      double a[CH][4];
#pragma unroll
      for (int c = 0; c < CH; ++c) a[c][0] = a[c][1] = a[c][2] = a[c][3] = 0.0;
      int r = ry;
      for (; r + 3 * RL < rows; r += 4 * RL) {
        float v[CH][4];
#pragma unroll
        for (int c = 0; c < CH; ++c)
#pragma unroll
          for (int u = 0; u < 4; ++u) v[c][u] = partials[(size_t)(r + u * RL) * n + idx[c]];
#pragma unroll
        for (int c = 0; c < CH; ++c)
#pragma unroll
          for (int u = 0; u < 4; ++u) a[c][u] += (double)v[c][u];
      }
      for (; r < rows; r += RL) {
        float v[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) v[c] = partials[(size_t)r * n + idx[c]];
#pragma unroll
        for (int c = 0; c < CH; ++c) a[c][0] += (double)v[c];
      }
#pragma unroll
      for (int c = 0; c < CH; ++c) sm[(c * RL + ry) * EL + ex] = (a[c][0] + a[c][1]) + (a[c][2] + a[c][3]);
      __syncthreads();
      // the thread group ry finishes chunk ry (CH == RL)
      const size_t i = i0 + (size_t)ry * EL;
      if (i < n) {
        double acc = 0.0;
#pragma unroll
        for (int q = 0; q < RL; ++q) acc += sm[(ry * RL + q) * EL + ex];
        dst[i] = (float)acc;
      }
      __syncthreads();
    }
  }
}

// The arithmetic of reduce_rows_kernel<16, 64> (64 row lanes, each with four interleaved accumulators, lane totals added in
// lane order) on a 64-element x 4-thread-group mapping: a thread walks 16 of the 64 lanes one after the other, so a wave
// reads 256 contiguous bytes per row instead of 64 -- the batched launch streams ~10^8 bytes of slabs and is bound by
// that, where the per-layer launches were bound by their own latency.  Bit-identical sums.
__global__ __launch_bounds__(256) void reduce_rows_batched_lanes64_kernel(const ReduceJob* __restrict__ jobs,
                                                                          const int2* __restrict__ blockmap) {
  constexpr int EL = 64, RL = 64, G = 4;
  __shared__ double sm[RL][EL];
  const int2 bm = blockmap[blockIdx.x];
  const ReduceJob jb = jobs[bm.x];
  const float* __restrict__ partials = jb.src;
  const size_t n = (size_t)jb.n;
  const int rows = jb.rows;
  const int ex = threadIdx.x % EL, g = threadIdx.x / EL;
  const size_t i = (size_t)bm.y * EL + ex;
  // four of the thread's 16 lanes at a time (16 loads in flight instead of 4); every lane's accumulators still receive
  // their rows in the same order
  const size_t ic = i < n ? i : n - 1;
  if (rows <= 3 * RL) {
    // at most three rows per lane and no lane with a full group of four (that needs row q + 192): the thread's 16 lanes at once, every
    // row requested before the first is added.  The lane total goes through the same additions as below (all rows into the first
    // accumulator in row order, then (a0 + 0.0) + (0.0 + 0.0)), so that a -0.0 comes out as it does there.
    auto few = [&](auto nr_tag) {
      constexpr int NR = decltype(nr_tag)::value;
      float v[RL / G][NR];
#pragma unroll
      for (int m = 0; m < RL / G; ++m)
#pragma unroll
        for (int t = 0; t < NR; ++t) v[m][t] = partials[(size_t)min(g + G * m + t * RL, rows - 1) * n + ic];
#pragma unroll
      for (int m = 0; m < RL / G; ++m) {
        double a0 = 0.0;
#pragma unroll
        for (int t = 0; t < NR; ++t) a0 += g + G * m + t * RL < rows ? (double)v[m][t] : 0.0;
        sm[g + G * m][ex] = (a0 + 0.0) + (0.0 + 0.0);
      }
    };
    if (rows <= RL) few(std::integral_constant<int, 1>());
    else if (rows <= 2 * RL) few(std::integral_constant<int, 2>());
    else few(std::integral_constant<int, 3>());
  } else
  for (int q0 = g; q0 < RL; q0 += 4 * G) {
    double a[4][4];
    int r[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { r[u] = q0 + G * u; a[u][0] = a[u][1] = a[u][2] = a[u][3] = 0.0; }
    while (r[3] + 3 * RL < rows) {          // the highest lane has a full group of four rows: so have the others
      float v[4][4];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < 4; ++j) v[u][j] = partials[(size_t)(r[u] + j * RL) * n + ic];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
#pragma unroll
        for (int j = 0; j < 4; ++j) a[u][j] += (double)v[u][j];
        r[u] += 4 * RL;
      }
    }
    // what is left of a lane: at most FOUR rows (the highest lane has no full group left and the others start <= 12 rows below it) --
    // one more full group or a tail of < 4 rows; all requested before the first is added (round 5: one at a time, the 42-144-row
    // jobs of the 33 x 33 layers ran at 2.5 TB/s with one load in flight per lane).  Same accumulators, same order; a skipped row
    // adds +0.0 (an accumulator that starts at +0.0 is never -0.0: x + 0.0 == x bitwise).
    float v[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int t = 0; t < 4; ++t) v[u][t] = partials[(size_t)min(r[u] + t * RL, rows - 1) * n + ic];     // (clamped, never conditional: a load inside a branch is waited for on the spot)
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const bool full = r[u] + 3 * RL < rows;          // (wave-uniform: the lane is a function of the wave and u)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const double d = r[u] + t * RL < rows ? (double)v[u][t] : 0.0;
        if (full) a[u][t] += d; else a[u][0] += d;
      }
      sm[q0 + G * u][ex] = (a[u][0] + a[u][1]) + (a[u][2] + a[u][3]);
    }
  }
  __syncthreads();
  if (g == 0 && i < n) {
    double acc = 0.0;
#pragma unroll 8
    for (int q = 0; q < RL; ++q) acc += sm[q][ex];
    jb.dst[i] = (float)acc;
  }
}

extern "C" int dl3p_reduce_rows_variant(int rows, size_t n) { return (n >= 64 * 1024 || rows <= 16) ? 0 : 1; }
extern "C" int dl3p_reduce_rows_block_elements(int variant) { return variant == 0 ? 64 * DL3P_RB_WIDE * DL3P_RB_NV : 64; }

extern "C" int dl3p_reduce_rows_batched(const void* jobs, const int* blockmap0, int blocks0, const int* blockmap1, int blocks1,
                                        void* stream) {
  DL3P_CHECK_ARG(jobs && blocks0 >= 0 && blocks1 >= 0 && (!blocks0 || blockmap0) && (!blocks1 || blockmap1),
                 "dl3p_reduce_rows_batched: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  static const int per_cu = getenv("DL3P_RB_PER_CU") ? atoi(getenv("DL3P_RB_PER_CU")) : 8;      // (0: one workgroup per block)
  if (blocks0)
    hipLaunchKernelGGL(reduce_rows_batched_wide_kernel, dim3(per_cu > 0 ? std::min(blocks0, per_cu * dl3p_device_cus()) : blocks0),
                       dim3(256), 0, st, (const ReduceJob*)jobs, (const int2*)blockmap0, blocks0);
  if (blocks1)
    hipLaunchKernelGGL(reduce_rows_batched_lanes64_kernel, dim3(blocks1), dim3(256), 0, st, (const ReduceJob*)jobs,
                       (const int2*)blockmap1);
  DL3P_CHECK_LAUNCH("dl3p_reduce_rows_batched");
  return DL3P_OK;
}

// sums[i] = sum_r partials[r][i] in double (SyncBN: these sums are all-reduced across ranks).
// 16 elements x 64 row lanes per workgroup, two rows in flight per thread.
__global__ __launch_bounds__(1024) void reduce_partials_kernel(const float* __restrict__ partials, int rows, int n,
                                                               double* __restrict__ sums) {
  __shared__ double sm[64][16];
  const int ex = threadIdx.x & 15, ry = threadIdx.x >> 4;
  const int i = blockIdx.x * 16 + ex;
  double a0 = 0.0, a1 = 0.0;
  if (i < n) {
    int r = ry;
    for (; r + 64 < rows; r += 128) {
      a0 += (double)partials[(size_t)r * n + i];
      a1 += (double)partials[(size_t)(r + 64) * n + i];
    }
    if (r < rows) a0 += (double)partials[(size_t)r * n + i];
  }
  sm[ry][ex] = a0 + a1;
  __syncthreads();
  if (ry == 0 && i < n) {
    double acc = 0.0;
#pragma unroll 8
    for (int q = 0; q < 64; ++q) acc += sm[q][ex];
    sums[i] = acc;
  }
}
extern "C" int dl3p_bn_reduce_partials(const float* partials, int rows, int C2, double* sums, void* stream) {
  DL3P_CHECK_ARG(partials && sums && rows > 0 && C2 > 0, "dl3p_bn_reduce_partials: bad arguments");
  hipLaunchKernelGGL(reduce_partials_kernel, dim3((C2 + 15) / 16), dim3(1024), 0, (hipStream_t)stream, partials, rows,
                     C2, sums);
  DL3P_CHECK_LAUNCH("dl3p_bn_reduce_partials");
  return DL3P_OK;
}

// ------------------------------------------------------------------------------ BN finalize
// 16 channels x 64 row lanes per workgroup (1024 threads); double accumulation, 2 rows in flight
#define FIN_CX 16
#define FIN_RY 64
__device__ __forceinline__ void reduce2_rows(const float* partials, int rows, const double* sums, int C, int c, int ry,
                                             double& a, double& b) {
  __shared__ double sm[2][FIN_RY][FIN_CX];
  const int cx = threadIdx.x % FIN_CX;
  double s0 = 0.0, s1 = 0.0, t0 = 0.0, t1 = 0.0;
  if (c < C) {
    if (sums) {
      if (ry == 0) { s0 = sums[c]; s1 = sums[C + c]; }
    } else {
      int r = ry;
      // the partial rows were just written by workgroups all over the chip: every load is an L2 miss, so 16 of them are
      // in flight per thread before the first add (same accumulators, same order as the two-row loop below)
      for (; r + 7 * FIN_RY < rows; r += 8 * FIN_RY) {
        float v[8][2];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          v[u][0] = partials[((size_t)(r + u * FIN_RY) * 2) * C + c];
          v[u][1] = partials[((size_t)(r + u * FIN_RY) * 2 + 1) * C + c];
        }
#pragma unroll
        for (int u = 0; u < 8; u += 2) {
          s0 += (double)v[u][0];
          s1 += (double)v[u][1];
          t0 += (double)v[u + 1][0];
          t1 += (double)v[u + 1][1];
        }
      }
      for (; r + FIN_RY < rows; r += 2 * FIN_RY) {
        s0 += (double)partials[((size_t)r * 2) * C + c];
        s1 += (double)partials[((size_t)r * 2 + 1) * C + c];
        t0 += (double)partials[((size_t)(r + FIN_RY) * 2) * C + c];
        t1 += (double)partials[((size_t)(r + FIN_RY) * 2 + 1) * C + c];
      }
      if (r < rows) {
        s0 += (double)partials[((size_t)r * 2) * C + c];
        s1 += (double)partials[((size_t)r * 2 + 1) * C + c];
      }
    }
  }
  sm[0][ry][cx] = s0 + t0;
  sm[1][ry][cx] = s1 + t1;
  __syncthreads();
  a = 0.0; b = 0.0;
  if (ry == 0) {
#pragma unroll 8
    for (int q = 0; q < FIN_RY; ++q) { a += sm[0][q][cx]; b += sm[1][q][cx]; }
  }
}

__global__ __launch_bounds__(1024) void bn_finalize_kernel(const float* partials, int rows, const double* sums, int C,
                                                          double count, const float* gamma, const float* beta,
                                                          float eps, float momentum, float* moving_mean,
                                                          float* moving_var, int update_moving, float* scale,
                                                          float* shift, float* save_mean, float* save_invstd) {
  const int cx = threadIdx.x % FIN_CX, ry = threadIdx.x / FIN_CX;
  const int c = blockIdx.x * FIN_CX + cx;
  // the per-channel parameters are fetched under the reduction, not after it
  float ga = 0.f, be = 0.f, mm = 0.f, mv = 0.f;
  if (ry == 0 && c < C) {
    ga = gamma[c];
    be = beta[c];
    if (update_moving) { mm = moving_mean[c]; mv = moving_var[c]; }
  }
  double s, ss;
  reduce2_rows(partials, rows, sums, C, c, ry, s, ss);
  if (ry == 0 && c < C) {
    double mean = s / count;
    double var = ss / count - mean * mean;
    if (var < 0.0) var = 0.0;
    float invstd = (float)(1.0 / sqrt(var + (double)eps));
    float sc = ga * invstd;
    scale[c] = sc;
    shift[c] = be - (float)mean * sc;
    save_mean[c] = (float)mean;
    save_invstd[c] = invstd;
    if (update_moving) {
      // update_moving 1: the biased batch variance feeds the moving average too (Keras' non-fused path =
      // SyncBatchNormalization, layers.py:65-66); 2: Bessel-corrected, count / (count - 1) (the fused FusedBatchNormV3 kernel
      // behind plain BatchNormalization, layers.py:68 -- what the string compare at layers.py:64 selects under the pinned
      // tensorflow==2.11.0)
      const double mvar = (update_moving == 2 && count > 1.0) ? var * (count / (count - 1.0)) : var;
      moving_mean[c] = mm * momentum + (float)mean * (1.f - momentum);
      moving_var[c] = mv * momentum + (float)mvar * (1.f - momentum);
    }
  }
}

extern "C" int dl3p_bn_finalize(const float* partials, int rows, const double* sums, int C, double count,
                                const float* gamma, const float* beta, float eps, float momentum, float* moving_mean,
                                float* moving_var, int update_moving, float* scale, float* shift, float* save_mean,
                                float* save_invstd, void* stream) {
  DL3P_CHECK_ARG((partials && rows > 0) || sums, "dl3p_bn_finalize: need partial rows or sums");
  DL3P_CHECK_ARG(C > 0 && count > 0 && gamma && beta && scale && shift && save_mean && save_invstd,
                 "dl3p_bn_finalize: bad arguments");
  DL3P_CHECK_ARG(!update_moving || (moving_mean && moving_var), "dl3p_bn_finalize: moving stats missing");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(ceil_div(C, FIN_CX)), dim3(FIN_CX * FIN_RY), 0, (hipStream_t)stream, partials, rows,
                     sums, C, count, gamma, beta, eps, momentum, moving_mean, moving_var, update_moving, scale, shift,
                     save_mean, save_invstd);
  DL3P_CHECK_LAUNCH("dl3p_bn_finalize");
  return DL3P_OK;
}

__global__ void bn_infer_coeffs_kernel(const float* gamma, const float* beta, const float* mm, const float* mv,
                                       float eps, float* scale, float* shift, float* save_mean, float* save_invstd,
                                       int C) {
  int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  float invstd = 1.f / sqrtf(mv[c] + eps);
  float sc = gamma[c] * invstd;
  scale[c] = sc;
  shift[c] = beta[c] - mm[c] * sc;
  if (save_mean) save_mean[c] = mm[c];
  if (save_invstd) save_invstd[c] = invstd;
}
extern "C" int dl3p_bn_infer_coeffs(const float* gamma, const float* beta, const float* moving_mean,
                                    const float* moving_var, float eps, float* scale, float* shift, float* save_mean,
                                    float* save_invstd, int C, void* stream) {
  DL3P_CHECK_ARG(gamma && beta && moving_mean && moving_var && scale && shift && C > 0,
                 "dl3p_bn_infer_coeffs: bad arguments");
  hipLaunchKernelGGL(bn_infer_coeffs_kernel, dim3(ceil_div(C, 256)), dim3(256), 0, (hipStream_t)stream, gamma, beta,
                     moving_mean, moving_var, eps, scale, shift, save_mean, save_invstd, C);
  DL3P_CHECK_LAUNCH("dl3p_bn_infer_coeffs");
  return DL3P_OK;
}

// ------------------------------------------------------------------------------ BN backward
struct EwParams {
  const float* a; int lda;      // g (bwd) or x (fwd)
  const float* z; int ldz;
  const float* scale; const float* shift; int act;
  const float* mean; const float* invstd; const float* coef;
  const float* r; int ldr; const float* rscale; const float* rshift; int ract;
  float* out; int ldo;
  float* partials;
  long long M; int C;
  int c4s, px, nslab, nbx;
  float rate; uint64_t seed; const int64_t* step;
  int accumulate;
};

__device__ __forceinline__ float4 opt_ld4(const float* p, int c, float4 dflt) { return p ? ld4(p + c) : dflt; }

// pass 1: partial sums of dyy and dyy*xhat
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(EwParams p) {
  const int b = blockIdx.x;
  const int slab = b / p.nbx;
  const int bx = b - slab * p.nbx;
  const int pl = threadIdx.x / p.c4s;
  const int cl = threadIdx.x - pl * p.c4s;
  const bool active = pl < p.px;
  const int cbase4 = slab * p.c4s;
  const int c = (cbase4 + cl) * 4;
  float4 acc[2] = {zero4(), zero4()};
  if (active) {
    const float4 one = make_float4(1.f, 1.f, 1.f, 1.f);
    const float4 sc = opt_ld4(p.scale, c, one), sh = opt_ld4(p.shift, c, zero4());
    const float4 mu = opt_ld4(p.mean, c, zero4()), is = opt_ld4(p.invstd, c, one);
    const int act = p.act;
    for (long long m = (long long)bx * p.px + pl; m < p.M; m += (long long)p.nbx * p.px) {
      float4 g = ld4(p.a + (size_t)m * p.lda + c);
      float4 z = ld4(p.z + (size_t)m * p.ldz + c);
      float4 u = fma4(z, sc, sh);
      float4 d = make_float4(g.x * act_grad(u.x, act), g.y * act_grad(u.y, act), g.z * act_grad(u.z, act),
                             g.w * act_grad(u.w, act));
      float4 xh = make_float4((z.x - mu.x) * is.x, (z.y - mu.y) * is.y, (z.z - mu.z) * is.z, (z.w - mu.w) * is.w);
      acc[0] = add4(acc[0], d);
      acc[1] = fma4(d, xh, acc[1]);
    }
  }
  block_reduce_store<2>(acc, active, pl, cl, p.c4s, p.px, cbase4, p.C, p.partials + (size_t)bx * 2 * p.C);
}

static int ew_setup(EwParams& p, long long M, int C, int max_rows) {
  pick_lanes(C, &p.c4s, &p.px, &p.nslab);
  long long need = ceil_div_ll(M, p.px);
  static const int ew_per_cu = getenv("DL3P_EW_PER_CU") ? atoi(getenv("DL3P_EW_PER_CU")) : 4;   // 8 -> 4: fewer partial rows for the finalize kernels, same streaming rate
  long long target = DL3P_NUM_CUS * ew_per_cu / p.nslab;
  if (target < 1) target = 1;
  long long nbx = need < target ? need : target;
  if (nbx < 1) nbx = 1;
  if (nbx > max_rows) nbx = max_rows;
  p.nbx = (int)nbx;
  p.M = M;
  p.C = C;
  return p.nbx;
}

static int check_ew(const char* fn, const void* a, int lda, int C) {
  DL3P_CHECK_ARG(a != nullptr, "%s: null pointer", fn);
  DL3P_CHECK_ARG(C > 0 && C % 4 == 0, "%s: C=%d must be a positive multiple of 4", fn, C);
  DL3P_CHECK_ARG(lda % 4 == 0 && lda >= C && aligned16(a), "%s: bad layout (ld=%d)", fn, lda);
  return DL3P_OK;
}

extern "C" int dl3p_bn_bwd_reduce(const float* g, int ldg, const float* z, int ldz, const float* scale,
                                  const float* shift, int act, const float* save_mean, const float* save_invstd,
                                  float* partials, int* rows_out, int M, int C, void* stream) {
  int rc = check_ew("dl3p_bn_bwd_reduce", g, ldg, C);
  if (rc) return rc;
  rc = check_ew("dl3p_bn_bwd_reduce", z, ldz, C);
  if (rc) return rc;
  DL3P_CHECK_ARG(partials && M > 0, "dl3p_bn_bwd_reduce: bad arguments");
  EwParams p = {};
  p.a = g; p.lda = ldg; p.z = z; p.ldz = ldz; p.scale = scale; p.shift = shift; p.act = act;
  p.mean = save_mean; p.invstd = save_invstd; p.partials = partials;
  int rows = ew_setup(p, M, C, DL3P_MAX_STAT_ROWS);
  if (rows_out) *rows_out = rows;
  hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(p.nbx * p.nslab), dim3(256), 0, (hipStream_t)stream, p);
  DL3P_CHECK_LAUNCH("dl3p_bn_bwd_reduce");
  return DL3P_OK;
}

__global__ __launch_bounds__(1024) void bn_bwd_finalize_kernel(const float* partials, int rows, const double* sums,
                                                              int C, double count, const float* gamma,
                                                              const float* invstd, const float* scale, int frozen,
                                                              float* dgamma, float* dbeta, float* coef) {
  const int cx = threadIdx.x % FIN_CX, ry = threadIdx.x / FIN_CX;
  const int c = blockIdx.x * FIN_CX + cx;
  float c0 = 0.f;
  if (ry == 0 && c < C) c0 = frozen ? scale[c] : gamma[c] * invstd[c];      // fetched under the reduction
  double s, sx;
  reduce2_rows(partials, rows, sums, C, c, ry, s, sx);
  if (ry == 0 && c < C) {
    if (frozen) {
      coef[c] = c0;
      coef[C + c] = 0.f;
      coef[2 * C + c] = 0.f;
    } else {
      if (dgamma) dgamma[c] = (float)sx;
      if (dbeta) dbeta[c] = (float)s;
      coef[c] = c0;
      coef[C + c] = (float)(s / count);
      coef[2 * C + c] = (float)(sx / count);
    }
  }
}

extern "C" int dl3p_bn_bwd_finalize(const float* partials, int rows, const double* sums, int C, double count,
                                    const float* gamma, const float* save_invstd, const float* scale, int frozen,
                                    float* dgamma, float* dbeta, float* coef, void* stream) {
  DL3P_CHECK_ARG(frozen || (partials && rows > 0) || sums, "dl3p_bn_bwd_finalize: need partial rows or sums");
  DL3P_CHECK_ARG(C > 0 && coef && (frozen ? scale != nullptr : (gamma && save_invstd && count > 0)),
                 "dl3p_bn_bwd_finalize: bad arguments");
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(ceil_div(C, FIN_CX)), dim3(FIN_CX * FIN_RY), 0, (hipStream_t)stream,
                     frozen ? nullptr : partials, frozen ? 0 : rows, frozen ? nullptr : sums, C, count, gamma,
                     save_invstd, scale, frozen, dgamma, dbeta, coef);
  DL3P_CHECK_LAUNCH("dl3p_bn_bwd_finalize");
  return DL3P_OK;
}

// pass 2: dz = coef0*(dyy - coef1 - xhat*coef2)
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(EwParams p) {
  const int b = blockIdx.x;
  const int slab = b / p.nbx;
  const int bx = b - slab * p.nbx;
  const int pl = threadIdx.x / p.c4s;
  const int cl = threadIdx.x - pl * p.c4s;
  if (pl >= p.px) return;
  const int c = (slab * p.c4s + cl) * 4;
  const float4 one = make_float4(1.f, 1.f, 1.f, 1.f);
  const float4 sc = opt_ld4(p.scale, c, one), sh = opt_ld4(p.shift, c, zero4());
  const float4 mu = opt_ld4(p.mean, c, zero4()), is = opt_ld4(p.invstd, c, one);
  const float4 c0 = p.coef ? ld4(p.coef + c) : one;
  const float4 c1 = p.coef ? ld4(p.coef + p.C + c) : zero4();
  const float4 c2 = p.coef ? ld4(p.coef + 2 * p.C + c) : zero4();
  const int act = p.act;
  for (long long m = (long long)bx * p.px + pl; m < p.M; m += (long long)p.nbx * p.px) {
    float4 g = ld4(p.a + (size_t)m * p.lda + c);
    float4 z = ld4(p.z + (size_t)m * p.ldz + c);
    float4 u = fma4(z, sc, sh);
    float4 d = make_float4(g.x * act_grad(u.x, act), g.y * act_grad(u.y, act), g.z * act_grad(u.z, act),
                           g.w * act_grad(u.w, act));
    float4 o;
    o.x = c0.x * (d.x - c1.x - (z.x - mu.x) * is.x * c2.x);
    o.y = c0.y * (d.y - c1.y - (z.y - mu.y) * is.y * c2.y);
    o.z = c0.z * (d.z - c1.z - (z.z - mu.z) * is.z * c2.z);
    o.w = c0.w * (d.w - c1.w - (z.w - mu.w) * is.w * c2.w);
    float* op = p.out + (size_t)m * p.ldo + c;
    if (p.accumulate) o = add4(o, ld4(op));
    st4(op, o);
  }
}

extern "C" int dl3p_bn_bwd_apply(const float* g, int ldg, const float* z, int ldz, const float* scale,
                                 const float* shift, int act, const float* save_mean, const float* save_invstd,
                                 const float* coef, float* dz, int lddz, int accumulate, int M, int C,
                                 void* stream) {
  int rc = check_ew("dl3p_bn_bwd_apply", g, ldg, C);
  if (rc) return rc;
  rc = check_ew("dl3p_bn_bwd_apply", z, ldz, C);
  if (rc) return rc;
  rc = check_ew("dl3p_bn_bwd_apply", dz, lddz, C);
  if (rc) return rc;
  EwParams p = {};
  p.a = g; p.lda = ldg; p.z = z; p.ldz = ldz; p.scale = scale; p.shift = shift; p.act = act;
  p.mean = save_mean; p.invstd = save_invstd; p.coef = coef; p.out = dz; p.ldo = lddz; p.accumulate = accumulate;
  ew_setup(p, M, C, 1 << 20);
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(p.nbx * p.nslab), dim3(256), 0, (hipStream_t)stream, p);
  DL3P_CHECK_LAUNCH("dl3p_bn_bwd_apply");
  return DL3P_OK;
}

// ------------------------------------------------------------------------------ materialise / residual / dropout
__global__ __launch_bounds__(256) void affine_act_kernel(EwParams p) {
  const int b = blockIdx.x;
  const int slab = b / p.nbx;
  const int bx = b - slab * p.nbx;
  const int pl = threadIdx.x / p.c4s;
  const int cl = threadIdx.x - pl * p.c4s;
  if (pl >= p.px) return;
  const int c = (slab * p.c4s + cl) * 4;
  const float4 one = make_float4(1.f, 1.f, 1.f, 1.f);
  const float4 sc = opt_ld4(p.scale, c, one), sh = opt_ld4(p.shift, c, zero4());
  const float4 rsc = opt_ld4(p.rscale, c, one), rsh = opt_ld4(p.rshift, c, zero4());
  const int64_t step = (p.rate > 0.f && p.step) ? *p.step : 0;
  const float keep_scale = p.rate > 0.f ? 1.f / (1.f - p.rate) : 1.f;
  for (long long m = (long long)bx * p.px + pl; m < p.M; m += (long long)p.nbx * p.px) {
    float4 v = act_apply4(fma4(ld4(p.a + (size_t)m * p.lda + c), sc, sh), p.act);
    if (p.rate > 0.f) {
      const uint64_t e = (uint64_t)m * p.C + c;
      v.x = dropout_keep(p.seed, step, e + 0, p.rate) ? v.x * keep_scale : 0.f;
      v.y = dropout_keep(p.seed, step, e + 1, p.rate) ? v.y * keep_scale : 0.f;
      v.z = dropout_keep(p.seed, step, e + 2, p.rate) ? v.z * keep_scale : 0.f;
      v.w = dropout_keep(p.seed, step, e + 3, p.rate) ? v.w * keep_scale : 0.f;
    }
    if (p.r) v = add4(v, act_apply4(fma4(ld4(p.r + (size_t)m * p.ldr + c), rsc, rsh), p.ract));
    st4(p.out + (size_t)m * p.ldo + c, v);
  }
}

extern "C" int dl3p_affine_act(const float* x, int ldx, const float* scale, const float* shift, int act,
                               const float* r, int ldr, const float* rscale, const float* rshift, int ract,
                               float dropout_rate, uint64_t seed, const int64_t* step_counter, float* y, int ldy,
                               int M, int C, void* stream) {
  int rc = check_ew("dl3p_affine_act", x, ldx, C);
  if (rc) return rc;
  rc = check_ew("dl3p_affine_act", y, ldy, C);
  if (rc) return rc;
  if (r) { rc = check_ew("dl3p_affine_act", r, ldr, C); if (rc) return rc; }
  DL3P_CHECK_ARG(dropout_rate >= 0.f && dropout_rate < 1.f && M > 0, "dl3p_affine_act: bad arguments");
  EwParams p = {};
  p.a = x; p.lda = ldx; p.scale = scale; p.shift = shift; p.act = act;
  p.r = r; p.ldr = ldr; p.rscale = rscale; p.rshift = rshift; p.ract = ract;
  p.rate = dropout_rate; p.seed = seed; p.step = step_counter; p.out = y; p.ldo = ldy;
  ew_setup(p, M, C, 1 << 20);
  hipLaunchKernelGGL(affine_act_kernel, dim3(p.nbx * p.nslab), dim3(256), 0, (hipStream_t)stream, p);
  DL3P_CHECK_LAUNCH("dl3p_affine_act");
  return DL3P_OK;
}

__global__ __launch_bounds__(256) void scale_mask_bwd_kernel(EwParams p) {
  const int b = blockIdx.x;
  const int slab = b / p.nbx;
  const int bx = b - slab * p.nbx;
  const int pl = threadIdx.x / p.c4s;
  const int cl = threadIdx.x - pl * p.c4s;
  if (pl >= p.px) return;
  const int c = (slab * p.c4s + cl) * 4;
  const int64_t step = (p.rate > 0.f && p.step) ? *p.step : 0;
  const float keep_scale = p.rate > 0.f ? 1.f / (1.f - p.rate) : 1.f;
  for (long long m = (long long)bx * p.px + pl; m < p.M; m += (long long)p.nbx * p.px) {
    float4 v = ld4(p.a + (size_t)m * p.lda + c);
    if (p.rate > 0.f) {
      const uint64_t e = (uint64_t)m * p.C + c;
      v.x = dropout_keep(p.seed, step, e + 0, p.rate) ? v.x * keep_scale : 0.f;
      v.y = dropout_keep(p.seed, step, e + 1, p.rate) ? v.y * keep_scale : 0.f;
      v.z = dropout_keep(p.seed, step, e + 2, p.rate) ? v.z * keep_scale : 0.f;
      v.w = dropout_keep(p.seed, step, e + 3, p.rate) ? v.w * keep_scale : 0.f;
    }
    float* o = p.out + (size_t)m * p.ldo + c;
    if (p.accumulate) v = add4(v, ld4(o));
    st4(o, v);
  }
}

extern "C" int dl3p_scale_mask_bwd(const float* gy, int ldgy, float dropout_rate, uint64_t seed,
                                   const int64_t* step_counter, float* gx, int ldgx, int accumulate, int M, int C,
                                   void* stream) {
  int rc = check_ew("dl3p_scale_mask_bwd", gy, ldgy, C);
  if (rc) return rc;
  rc = check_ew("dl3p_scale_mask_bwd", gx, ldgx, C);
  if (rc) return rc;
  EwParams p = {};
  p.a = gy; p.lda = ldgy; p.rate = dropout_rate; p.seed = seed; p.step = step_counter;
  p.out = gx; p.ldo = ldgx; p.accumulate = accumulate;
  ew_setup(p, M, C, 1 << 20);
  hipLaunchKernelGGL(scale_mask_bwd_kernel, dim3(p.nbx * p.nslab), dim3(256), 0, (hipStream_t)stream, p);
  DL3P_CHECK_LAUNCH("dl3p_scale_mask_bwd");
  return DL3P_OK;
}

__global__ void dropout_mask_kernel(float rate, uint64_t seed, const int64_t* step, float* mask, size_t n) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  mask[i] = dropout_keep(seed, step ? *step : 0, i, rate) ? 1.f : 0.f;
}
extern "C" int dl3p_dropout_mask(float dropout_rate, uint64_t seed, const int64_t* step_counter, float* mask, int M,
                                 int C, void* stream) {
  DL3P_CHECK_ARG(mask && M > 0 && C > 0, "dl3p_dropout_mask: bad arguments");
  size_t n = (size_t)M * C;
  hipLaunchKernelGGL(dropout_mask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     dropout_rate, seed, step_counter, mask, n);
  DL3P_CHECK_LAUNCH("dl3p_dropout_mask");
  return DL3P_OK;
}

__global__ void fill_kernel(float* p, float v, size_t n) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  size_t stride = (size_t)gridDim.x * 256;
  for (; i < n; i += stride) p[i] = v;
}
extern "C" int dl3p_fill(float* p, float value, size_t n, void* stream) {
  DL3P_CHECK_ARG(p || n == 0, "dl3p_fill: null pointer");
  if (n == 0) return DL3P_OK;
  size_t blocks = (n + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(fill_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, value, n);
  DL3P_CHECK_LAUNCH("dl3p_fill");
  return DL3P_OK;
}

__global__ void increment_kernel(int64_t* c) { if (threadIdx.x == 0 && blockIdx.x == 0) *c += 1; }
extern "C" int dl3p_increment_counter(int64_t* counter, void* stream) {
  DL3P_CHECK_ARG(counter, "dl3p_increment_counter: null pointer");
  hipLaunchKernelGGL(increment_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, counter);
  DL3P_CHECK_LAUNCH("dl3p_increment_counter");
  return DL3P_OK;
}

// ------------------------------------------------------------------------------ uint8 batches at the boundary
// common/data_utils.py:403-417 normalize_image (`image.astype(np.float32) / 127.5 - 1.0`) and the label cast of
// deeplabv3p/data.py:116-124 done on the device: the host hands over the bytes it decoded (a quarter of the PCIe
// traffic of float32 batches).  True IEEE division, so the result is bit-identical to NumPy's.
__global__ __launch_bounds__(256) void u8_to_float_kernel(const unsigned char* __restrict__ src, float* __restrict__ dst,
                                                          size_t n, float div, float sub) {
  const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i + 3 < n) {
    const uchar4 v = *reinterpret_cast<const uchar4*>(src + i);
    st4(dst + i, make_float4((float)v.x / div - sub, (float)v.y / div - sub, (float)v.z / div - sub, (float)v.w / div - sub));
  } else {
    for (size_t j = i; j < n; ++j) dst[j] = (float)src[j] / div - sub;
  }
}
extern "C" int dl3p_u8_to_float(const unsigned char* src, float* dst, size_t n, float divide_by, float subtract,
                                void* stream) {
  DL3P_CHECK_ARG(src && dst && ((uintptr_t)src % 4 == 0) && aligned16(dst), "dl3p_u8_to_float: bad pointers");
  if (n == 0) return DL3P_OK;
  const size_t blocks = (n / 4 + 256) / 256;
  hipLaunchKernelGGL(u8_to_float_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, dst, n, divide_by,
                     subtract);
  DL3P_CHECK_LAUNCH("dl3p_u8_to_float");
  return DL3P_OK;
}

// ------------------------------------------------------------------------------ global average pooling
// Per-image reductions over HW pixels (pooling, SE-block backward).  Workgroups = (image, channel slab, pixel
// chunk).  With one chunk the workgroup owns the whole image and stores the result; with several, each stores an
// unscaled partial row into the workspace, takes a ticket, and the workgroup that draws the last ticket of its
// (image, slab) sums the partial rows in chunk order -- the same order whichever workgroup ends up doing it, so the
// result is deterministic.  Tickets live at the head of the workspace and are reset to zero by that last workgroup.
struct PoolPlan { int c4s, px, nslab, nchunk, per; size_t ticket_floats; };
static PoolPlan pool_plan(int N, int HW, int C, bool chunked) {
  PoolPlan pl;
  const int c4 = C / 4;
  int d = 1;
  const int dmax = chunked ? 64 : 16;   // chunks supply the parallelism: whole rows per pixel lane when they fit
  for (int k = 1; k <= dmax && k <= c4; ++k) if (c4 % k == 0) d = k;
  pl.c4s = d; pl.px = 256 / d; pl.nslab = c4 / d;
  int nchunk = 1;
  if (chunked) {
    static const int want = getenv("DL3P_POOL_WGS") ? atoi(getenv("DL3P_POOL_WGS")) : 512;
    nchunk = (want + N * pl.nslab - 1) / (N * pl.nslab);
    const int most = (HW + 4 * pl.px - 1) / (4 * pl.px);          // >= 4 pixels per lane and chunk
    if (nchunk > most) nchunk = most;
    if (nchunk > 64) nchunk = 64;
    if (nchunk < 1) nchunk = 1;
  }
  pl.per = (HW + nchunk - 1) / nchunk;
  pl.nchunk = (HW + pl.per - 1) / pl.per;
  pl.ticket_floats = (((size_t)N * pl.nslab + 63) / 64) * 64;
  return pl;
}
extern "C" size_t dl3p_pool_workspace(int N, int HW, int C) {
  if (N <= 0 || HW <= 0 || C <= 0 || C % 4) return 0;
  const PoolPlan pl = pool_plan(N, HW, C, true);
  return (pl.ticket_floats + (size_t)N * pl.nchunk * C) * sizeof(float);
}
static bool pool_ws_ok(const float* ws, size_t bytes, int N, int HW, int C) {
  return ws && aligned16(ws) && bytes >= dl3p_pool_workspace(N, HW, C);
}

// All 256 threads call.  Block-reduces the pixel lanes' partial sums into row `chunk` of the workspace, takes a
// ticket, and lets the workgroup with the last ticket add the rows up.  The partial rows are written and read with
// agent-scope atomic accesses (write-through / L2-bypassing on the 8-XCD part); ONE wave per workgroup issues the agent-scope
// release / acquire pair around the ticket (a __threadfence() in every wave writes back the whole L2 four times over, which is
// ruinous next to a kernel that is streaming its output through it).
__device__ __forceinline__ void pool_finish(const float4& val, bool active, int pl, int cl, int px, float* ws,
                                            size_t ticket_floats, int n, int chunk, int slab, int nslab, int nchunk,
                                            int c4s, int cbase4, int C, float* out_row, float scale) {
  __shared__ float4 sm[256];
  __shared__ int last;
  if (active) sm[pl * c4s + cl] = val;
  __syncthreads();
  float* rows = ws + ticket_floats + (size_t)n * nchunk * C + (size_t)(cbase4 + threadIdx.x) * 4;
  if ((int)threadIdx.x < c4s) {
    float4 a = sm[threadIdx.x];
    for (int q = 1; q < px; ++q) a = add4(a, sm[q * c4s + threadIdx.x]);
    float* r = rows + (size_t)chunk * C;
    __hip_atomic_store(r + 0, a.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(r + 1, a.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(r + 2, a.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(r + 3, a.w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      // the stores above have left every lane ...
  __syncthreads();                                            // ... before the ticket is drawn
  int* tickets = reinterpret_cast<int*>(ws);
  if (threadIdx.x == 0) {
    // AGENT-scope release in front of the ticket and acquire behind it (one wave of the workgroup: the barrier above makes it
    // cumulative).  Round 6: with the workgroup-scope fence alone the ticket -- another address, another memory channel -- could
    // overtake the write-through rows on their way out of this XCD, and the workgroup with the last ticket (on another XCD) summed a
    // stale row: test_chunked_per_image_reductions failed once in ~15 suite runs with sums differing from call to call.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    last = __hip_atomic_fetch_add(&tickets[n * nslab + slab], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nchunk - 1;
    if (last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
  if (!last) return;
  if ((int)threadIdx.x < c4s) {
    float4 a = zero4();
    for (int q = 0; q < nchunk; ++q) {
      const float* r = rows + (size_t)q * C;
      a.x += __hip_atomic_load(r + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      a.y += __hip_atomic_load(r + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      a.z += __hip_atomic_load(r + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      a.w += __hip_atomic_load(r + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    st4(out_row + (size_t)(cbase4 + threadIdx.x) * 4, make_float4(a.x * scale, a.y * scale, a.z * scale, a.w * scale));
  }
  if (threadIdx.x == 0) __hip_atomic_store(&tickets[n * nslab + slab], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(256) void gap_fwd_kernel(EwParams p, int HW, float out_scale, float* ws,
                                                      size_t ticket_floats, int nchunk, int per) {
  const int chunk = blockIdx.x % nchunk;
  const int rest = blockIdx.x / nchunk;
  const int n = rest / p.nslab;
  const int slab = rest - n * p.nslab;
  const int pl = threadIdx.x / p.c4s;
  const int cl = threadIdx.x - pl * p.c4s;
  const bool active = pl < p.px;
  const int cbase4 = slab * p.c4s;
  const int c = (cbase4 + cl) * 4;
  const float inv = out_scale / (float)HW;
  float4 acc[1] = {zero4()};
  if (active) {
    const float4 one = make_float4(1.f, 1.f, 1.f, 1.f);
    const float4 sc = opt_ld4(p.scale, c, one), sh = opt_ld4(p.shift, c, zero4());
    const float* base = p.a + (size_t)n * HW * p.lda + c;
    const int i1 = min(HW, (chunk + 1) * per);
    // 8 independent accumulators: 8 row loads in flight per thread
    float4 ax[7];
#pragma unroll
    for (int q = 0; q < 7; ++q) ax[q] = zero4();
    int i = chunk * per + pl;
    for (; i + 7 * p.px < i1; i += 8 * p.px) {
      float4 v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = ld4(base + (size_t)(i + q * p.px) * p.lda);
      acc[0] = add4(acc[0], act_apply4(fma4(v[0], sc, sh), p.act));
#pragma unroll
      for (int q = 1; q < 8; ++q) ax[q - 1] = add4(ax[q - 1], act_apply4(fma4(v[q], sc, sh), p.act));
    }
    for (; i < i1; i += p.px) acc[0] = add4(acc[0], act_apply4(fma4(ld4(base + (size_t)i * p.lda), sc, sh), p.act));
    acc[0] = add4(add4(add4(acc[0], ax[0]), add4(ax[1], ax[2])), add4(add4(ax[3], ax[4]), add4(ax[5], ax[6])));
    if (nchunk == 1) acc[0] = make_float4(acc[0].x * inv, acc[0].y * inv, acc[0].z * inv, acc[0].w * inv);
  }
  if (nchunk == 1) {
    // out row n: block_reduce_store with C := ldo so that the row stride is honoured
    block_reduce_store<1>(acc, active, pl, cl, p.c4s, p.px, cbase4, p.ldo, p.out + (size_t)n * p.ldo);
    return;
  }
  pool_finish(acc[0], active, pl, cl, p.px, ws, ticket_floats, n, chunk, slab, p.nslab, nchunk, p.c4s, cbase4, p.C,
              p.out + (size_t)n * p.ldo, inv);
}

extern "C" int dl3p_global_avgpool_fwd(const float* x, int ldx, const float* in_scale, const float* in_shift,
                                       int in_act, float* y, int ldy, float out_scale, int N, int HW, int C,
                                       float* workspace, size_t workspace_bytes, void* stream) {
  int rc = check_ew("dl3p_global_avgpool_fwd", x, ldx, C);
  if (rc) return rc;
  rc = check_ew("dl3p_global_avgpool_fwd", y, ldy, C);
  if (rc) return rc;
  DL3P_CHECK_ARG(N > 0 && HW > 0, "dl3p_global_avgpool_fwd: bad dims");
  EwParams p = {};
  p.a = x; p.lda = ldx; p.scale = in_scale; p.shift = in_shift; p.act = in_act; p.out = y; p.ldo = ldy;
  // the ticketed finish costs ~5 us of serial latency at the kernel's tail (write-through store, ticket, re-read):
  // chunk only when (image, slab) workgroups alone leave most of the chip idle (33x33x960: 240 workgroups, 18 us whole
  // vs 21 us chunked; 33x33x320: 80 workgroups, 12 vs 15 us; 65x65x120: 32 workgroups, 27-52 us whole vs 23 us chunked)
  bool chunked = pool_ws_ok(workspace, workspace_bytes, N, HW, C);
  if (chunked && N * pool_plan(N, HW, C, false).nslab >= DL3P_NUM_CUS / 4) chunked = false;
  const PoolPlan pl = pool_plan(N, HW, C, chunked);
  p.c4s = pl.c4s; p.px = pl.px; p.nslab = pl.nslab; p.C = C;
  hipLaunchKernelGGL(gap_fwd_kernel, dim3(N * p.nslab * pl.nchunk), dim3(256), 0, (hipStream_t)stream, p, HW,
                     out_scale, workspace, pl.ticket_floats, pl.nchunk, pl.per);
  DL3P_CHECK_LAUNCH("dl3p_global_avgpool_fwd");
  return DL3P_OK;
}

__global__ __launch_bounds__(256) void gap_bwd_kernel(EwParams p, int HW) {
  const int b = blockIdx.x;
  const int slab = b / p.nbx;
  const int bx = b - slab * p.nbx;
  const int pl = threadIdx.x / p.c4s;
  const int cl = threadIdx.x - pl * p.c4s;
  if (pl >= p.px) return;
  const int c = (slab * p.c4s + cl) * 4;
  const float inv = 1.f / (float)HW;
  for (long long m = (long long)bx * p.px + pl; m < p.M; m += (long long)p.nbx * p.px) {
    const long long n = m / HW;
    float4 g = ld4(p.a + (size_t)n * p.lda + c);
    g = make_float4(g.x * inv, g.y * inv, g.z * inv, g.w * inv);
    float* o = p.out + (size_t)m * p.ldo + c;
    if (p.accumulate) g = add4(g, ld4(o));
    st4(o, g);
  }
}

extern "C" int dl3p_global_avgpool_bwd(const float* gy, int ldgy, float* gx, int ldgx, int accumulate, int N, int HW,
                                       int C, void* stream) {
  int rc = check_ew("dl3p_global_avgpool_bwd", gy, ldgy, C);
  if (rc) return rc;
  rc = check_ew("dl3p_global_avgpool_bwd", gx, ldgx, C);
  if (rc) return rc;
  EwParams p = {};
  p.a = gy; p.lda = ldgy; p.out = gx; p.ldo = ldgx; p.accumulate = accumulate;
  ew_setup(p, (long long)N * HW, C, 1 << 20);
  hipLaunchKernelGGL(gap_bwd_kernel, dim3(p.nbx * p.nslab), dim3(256), 0, (hipStream_t)stream, p, HW);
  DL3P_CHECK_LAUNCH("dl3p_global_avgpool_bwd");
  return DL3P_OK;
}

// ------------------------------------------------------------------------------ SE-block multiply
// y[n,p,c] = act(x*scale+shift) * act_s(s[n,c])  (deeplabv3p_mobilenetv3.py:145 Multiply); workgroups = (image,
// channel slab, pixel chunk) like the pooling kernel, so the backward's per-image reduction needs no float atomics
__global__ __launch_bounds__(256) void scale_bcast_fwd_kernel(EwParams p, int HW, const float* s, int lds, int s_act,
                                                              int nchunk, int per) {
  const int chunk = blockIdx.x % nchunk;
  const int rest = blockIdx.x / nchunk;
  const int n = rest / p.nslab;
  const int slab = rest - n * p.nslab;
  const int pl = threadIdx.x / p.c4s;
  const int cl = threadIdx.x - pl * p.c4s;
  if (pl >= p.px) return;
  const int c = (slab * p.c4s + cl) * 4;
  const float4 one = make_float4(1.f, 1.f, 1.f, 1.f);
  const float4 sc = opt_ld4(p.scale, c, one), sh = opt_ld4(p.shift, c, zero4());
  const float4 sv = act_apply4(ld4(s + (size_t)n * lds + c), s_act);
  const float* base = p.a + (size_t)n * HW * p.lda + c;
  float* ob = p.out + (size_t)n * HW * p.ldo + c;
  const int i1 = min(HW, (chunk + 1) * per);
  int i = chunk * per + pl;
  for (; i + p.px < i1; i += 2 * p.px) {
    const float4 v0 = ld4(base + (size_t)i * p.lda), v1 = ld4(base + (size_t)(i + p.px) * p.lda);
    st4(ob + (size_t)i * p.ldo, mul4(act_apply4(fma4(v0, sc, sh), p.act), sv));
    st4(ob + (size_t)(i + p.px) * p.ldo, mul4(act_apply4(fma4(v1, sc, sh), p.act), sv));
  }
  if (i < i1) st4(ob + (size_t)i * p.ldo, mul4(act_apply4(fma4(ld4(base + (size_t)i * p.lda), sc, sh), p.act), sv));
}

__global__ __launch_bounds__(256) void scale_bcast_bwd_kernel(EwParams p, int HW, const float* s, int lds, int s_act,
                                                              const float* gy, int ldgy, float* gs, int ldgs,
                                                              float* ws, size_t ticket_floats, int nchunk, int per) {
  const int chunk = blockIdx.x % nchunk;
  const int rest = blockIdx.x / nchunk;
  const int n = rest / p.nslab;
  const int slab = rest - n * p.nslab;
  const int pl = threadIdx.x / p.c4s;
  const int cl = threadIdx.x - pl * p.c4s;
  const bool active = pl < p.px;
  const int cbase4 = slab * p.c4s;
  const int c = (cbase4 + cl) * 4;
  float4 acc[1] = {zero4()};
  if (active) {
    const float4 one = make_float4(1.f, 1.f, 1.f, 1.f);
    const float4 sc = opt_ld4(p.scale, c, one), sh = opt_ld4(p.shift, c, zero4());
    const float4 sv = act_apply4(ld4(s + (size_t)n * lds + c), s_act);
    const float* xb = p.a + (size_t)n * HW * p.lda + c;
    const float* gb = gy + (size_t)n * HW * ldgy + c;
    float* ob = p.out + (size_t)n * HW * p.ldo + c;
    const int i1 = min(HW, (chunk + 1) * per);
    float4 a1 = zero4();
    int i = chunk * per + pl;
    for (; i + p.px < i1; i += 2 * p.px) {
      const int j = i + p.px;
      const float4 g0 = ld4(gb + (size_t)i * ldgy), g1 = ld4(gb + (size_t)j * ldgy);
      const float4 x0 = ld4(xb + (size_t)i * p.lda), x1 = ld4(xb + (size_t)j * p.lda);
      float4 o0 = mul4(g0, sv), o1 = mul4(g1, sv);
      if (p.accumulate) { o0 = add4(o0, ld4(ob + (size_t)i * p.ldo)); o1 = add4(o1, ld4(ob + (size_t)j * p.ldo)); }
      acc[0] = fma4(g0, act_apply4(fma4(x0, sc, sh), p.act), acc[0]);
      a1 = fma4(g1, act_apply4(fma4(x1, sc, sh), p.act), a1);
      st4(ob + (size_t)i * p.ldo, o0);
      st4(ob + (size_t)j * p.ldo, o1);
    }
    if (i < i1) {
      const float4 g = ld4(gb + (size_t)i * ldgy);
      const float4 a = act_apply4(fma4(ld4(xb + (size_t)i * p.lda), sc, sh), p.act);
      acc[0] = fma4(g, a, acc[0]);
      float4 o = mul4(g, sv);
      if (p.accumulate) o = add4(o, ld4(ob + (size_t)i * p.ldo));
      st4(ob + (size_t)i * p.ldo, o);
    }
    acc[0] = add4(acc[0], a1);
  }
  if (nchunk == 1) {
    block_reduce_store<1>(acc, active, pl, cl, p.c4s, p.px, cbase4, ldgs, gs + (size_t)n * ldgs);
    return;
  }
  pool_finish(acc[0], active, pl, cl, p.px, ws, ticket_floats, n, chunk, slab, p.nslab, nchunk, p.c4s, cbase4, p.C,
              gs + (size_t)n * ldgs, 1.f);
}

extern "C" int dl3p_scale_bcast_fwd(const float* x, int ldx, const float* scale, const float* shift, int act,
                                    const float* s, int lds, int s_act, float* y, int ldy, int N, int HW, int C,
                                    void* stream) {
  int rc = check_ew("dl3p_scale_bcast_fwd", x, ldx, C);
  if (rc) return rc;
  rc = check_ew("dl3p_scale_bcast_fwd", y, ldy, C);
  if (rc) return rc;
  rc = check_ew("dl3p_scale_bcast_fwd", s, lds, C);
  if (rc) return rc;
  DL3P_CHECK_ARG(N > 0 && HW > 0, "dl3p_scale_bcast_fwd: bad dims");
  EwParams p = {};
  p.a = x; p.lda = ldx; p.scale = scale; p.shift = shift; p.act = act; p.out = y; p.ldo = ldy;
  const PoolPlan pl = pool_plan(N, HW, C, true);
  p.c4s = pl.c4s; p.px = pl.px; p.nslab = pl.nslab; p.C = C;
  hipLaunchKernelGGL(scale_bcast_fwd_kernel, dim3(N * p.nslab * pl.nchunk), dim3(256), 0, (hipStream_t)stream, p, HW,
                     s, lds, s_act, pl.nchunk, pl.per);
  DL3P_CHECK_LAUNCH("dl3p_scale_bcast_fwd");
  return DL3P_OK;
}

extern "C" int dl3p_scale_bcast_bwd(const float* gy, int ldgy, const float* x, int ldx, const float* scale,
                                    const float* shift, int act, const float* s, int lds, int s_act, float* gx,
                                    int ldgx, int accumulate_gx, float* gs, int ldgs, int N, int HW, int C,
                                    float* workspace, size_t workspace_bytes, void* stream) {
  int rc = check_ew("dl3p_scale_bcast_bwd", gy, ldgy, C);
  if (rc) return rc;
  rc = check_ew("dl3p_scale_bcast_bwd", x, ldx, C);
  if (rc) return rc;
  rc = check_ew("dl3p_scale_bcast_bwd", gx, ldgx, C);
  if (rc) return rc;
  rc = check_ew("dl3p_scale_bcast_bwd", gs, ldgs, C);
  if (rc) return rc;
  DL3P_CHECK_ARG(N > 0 && HW > 0, "dl3p_scale_bcast_bwd: bad dims");
  EwParams p = {};
  p.a = x; p.lda = ldx; p.scale = scale; p.shift = shift; p.act = act; p.out = gx; p.ldo = ldgx;
  p.accumulate = accumulate_gx;
  const PoolPlan pl = pool_plan(N, HW, C, pool_ws_ok(workspace, workspace_bytes, N, HW, C));
  p.c4s = pl.c4s; p.px = pl.px; p.nslab = pl.nslab; p.C = C;
  hipLaunchKernelGGL(scale_bcast_bwd_kernel, dim3(N * p.nslab * pl.nchunk), dim3(256), 0, (hipStream_t)stream, p, HW,
                     s, lds, s_act, gy, ldgy, gs, ldgs, workspace, pl.ticket_floats, pl.nchunk, pl.per);
  DL3P_CHECK_LAUNCH("dl3p_scale_bcast_bwd");
  return DL3P_OK;
}

// ------------------------------------------------------------------------------ SGD momentum
__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ w, float* __restrict__ v,
                                                  const float* __restrict__ g, size_t n4, size_t n,
                                                  const float* lr_dev, float momentum, float l2, float gscale,
                                                  const float* __restrict__ l2e, const float* __restrict__ lre) {
  const float lr = *lr_dev;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256;
  for (; i < n4; i += stride) {
    float4 ww = ld4(w + i * 4), vv = ld4(v + i * 4), gg = ld4(g + i * 4);
    float4 d = l2e ? ld4(l2e + i * 4) : make_float4(l2, l2, l2, l2);
    float4 m = lre ? ld4(lre + i * 4) : make_float4(1.f, 1.f, 1.f, 1.f);
    float4 nv;
    nv.x = momentum * vv.x - lr * fmaf(gg.x, gscale, 2.f * d.x * ww.x);
    nv.y = momentum * vv.y - lr * fmaf(gg.y, gscale, 2.f * d.y * ww.y);
    nv.z = momentum * vv.z - lr * fmaf(gg.z, gscale, 2.f * d.z * ww.z);
    nv.w = momentum * vv.w - lr * fmaf(gg.w, gscale, 2.f * d.w * ww.w);
    nv.x = m.x != 0.f ? nv.x : vv.x; nv.y = m.y != 0.f ? nv.y : vv.y;
    nv.z = m.z != 0.f ? nv.z : vv.z; nv.w = m.w != 0.f ? nv.w : vv.w;
    st4(v + i * 4, nv);
    st4(w + i * 4, make_float4(m.x != 0.f ? ww.x + nv.x : ww.x, m.y != 0.f ? ww.y + nv.y : ww.y,
                               m.z != 0.f ? ww.z + nv.z : ww.z, m.w != 0.f ? ww.w + nv.w : ww.w));
  }
  if (blockIdx.x == 0 && threadIdx.x < (n - n4 * 4)) {   // scalar tail
    size_t j = n4 * 4 + threadIdx.x;
    const float d = l2e ? l2e[j] : l2;
    if (!lre || lre[j] != 0.f) {
      float nv = momentum * v[j] - lr * fmaf(g[j], gscale, 2.f * d * w[j]);
      v[j] = nv;
      w[j] += nv;
    }
  }
}

extern "C" int dl3p_sgd_momentum(float* w, float* v, const float* g, size_t n, const float* lr_dev, float momentum,
                                 float l2, float grad_scale, const float* l2_elem, const float* lr_scale_elem,
                                 void* stream) {
  DL3P_CHECK_ARG(w && v && g && lr_dev, "dl3p_sgd_momentum: null pointer");
  DL3P_CHECK_ARG(aligned16(w) && aligned16(v) && aligned16(g) && aligned16(l2_elem) && aligned16(lr_scale_elem),
                 "dl3p_sgd_momentum: buffers must be 16-byte aligned");
  if (n == 0) return DL3P_OK;
  size_t n4 = n / 4;
  size_t blocks = (n4 + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(sgd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w, v, g, n4, n, lr_dev,
                     momentum, l2, grad_scale, l2_elem, lr_scale_elem);
  DL3P_CHECK_LAUNCH("dl3p_sgd_momentum");
  return DL3P_OK;
}

// ------------------------------------------------------------------------------ Adam / RMSprop
// common/model_utils.py:118-121 (train.py --optimizer adam | rmsprop), the Keras 2.11 update rules on the same flat
// buffers as SGD: g' = g*grad_scale + 2*l2*w (the regulariser is part of the Keras loss, so it passes through the
// moments), per-element freeze mask.
//   Adam(epsilon=1e-7):        m = b1 m + (1-b1) g';  v = b2 v + (1-b2) g'^2;
//                              w -= lr * sqrt(1 - b2^t) / (1 - b1^t) * m / (sqrt(v) + eps)        t = step count
//   RMSprop(rho=.9, eps=1e-7): v = rho v + (1-rho) g'^2;  w -= lr * g' / sqrt(v + eps)            (momentum 0, not centred)
// The step count is read from the device counter the forward plan increments, so graph replay follows it.
template <bool ADAM>
__global__ __launch_bounds__(256) void adaptive_kernel(float* __restrict__ w, float* __restrict__ m1, float* __restrict__ v2,
                                                       const float* __restrict__ g, size_t n, const float* lr_dev,
                                                       const int64_t* step, float b1, float b2, float eps, float gscale,
                                                       const float* __restrict__ l2e, const float* __restrict__ lre) {
  __shared__ float alpha_s;
  if (threadIdx.x == 0) {
    float a = *lr_dev;
    if (ADAM) {
      const double t = (double)(step ? *step : 1);
      a = (float)((double)a * sqrt(1.0 - pow((double)b2, t)) / (1.0 - pow((double)b1, t)));
    }
    alpha_s = a;
  }
  __syncthreads();
  const float alpha = alpha_s;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    if (lre && lre[i] == 0.f) continue;
    const float ww = w[i];
    const float gg = fmaf(g[i], gscale, 2.f * (l2e ? l2e[i] : 0.f) * ww);
    if (ADAM) {
      const float m = b1 * m1[i] + (1.f - b1) * gg;
      const float v = b2 * v2[i] + (1.f - b2) * gg * gg;
      m1[i] = m;
      v2[i] = v;
      w[i] = ww - alpha * m / (sqrtf(v) + eps);
    } else {
      const float v = b2 * v2[i] + (1.f - b2) * gg * gg;
      v2[i] = v;
      w[i] = ww - alpha * gg * rsqrtf(v + eps);
    }
  }
}

extern "C" int dl3p_adam_step(float* w, float* m, float* v, const float* g, size_t n, const float* lr_dev,
                              const int64_t* step_counter, float beta_1, float beta_2, float epsilon, float grad_scale,
                              const float* l2_elem, const float* lr_scale_elem, void* stream) {
  DL3P_CHECK_ARG(w && m && v && g && lr_dev && step_counter, "dl3p_adam_step: null pointer");
  if (n == 0) return DL3P_OK;
  size_t blocks = (n + 1023) / 1024;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL((adaptive_kernel<true>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w, m, v, g, n,
                     lr_dev, step_counter, beta_1, beta_2, epsilon, grad_scale, l2_elem, lr_scale_elem);
  DL3P_CHECK_LAUNCH("dl3p_adam_step");
  return DL3P_OK;
}

extern "C" int dl3p_rmsprop_step(float* w, float* v, const float* g, size_t n, const float* lr_dev, float rho,
                                 float epsilon, float grad_scale, const float* l2_elem, const float* lr_scale_elem,
                                 void* stream) {
  DL3P_CHECK_ARG(w && v && g && lr_dev, "dl3p_rmsprop_step: null pointer");
  if (n == 0) return DL3P_OK;
  size_t blocks = (n + 1023) / 1024;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL((adaptive_kernel<false>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w, (float*)nullptr,
                     v, g, n, lr_dev, (const int64_t*)nullptr, 0.f, rho, epsilon, grad_scale, l2_elem, lr_scale_elem);
  DL3P_CHECK_LAUNCH("dl3p_rmsprop_step");
  return DL3P_OK;
}

