// The label tail of SegmentationGenerator.__getitem__ (deeplabv3p/data.py:116-145) on the device, for batches whose
// labels arrive as the bytes the generator decoded:
//   label.astype('int32'); label[label > num_classes - 1] = ignore_index               (data.py:116-121)
//   class_weight.compute_class_weight('balanced', classes=np.unique(label), y=label)   (data.py:134-145)
//     = len(label) / (number of distinct values * count of the value), float64, stored into a float32 array
// On the host the second part costs ~10 ms per 513x513 image (np.unique + one putmask per class) -- ten steps of the
// network; here it is two launches over 4 MB.
#include "common.h"

namespace {

constexpr int BINS = 256;

__device__ __forceinline__ unsigned remap(unsigned v, unsigned nc, unsigned ign) { return v > nc - 1u ? ign : v; }

// pass 1: remapped labels out as float32, per-image histogram of the remapped values (workgroup-private LDS bins,
// flushed with integer atomics: exact and order-independent)
__global__ __launch_bounds__(256) void label_hist_kernel(const unsigned char* __restrict__ src, float* __restrict__ dst,
                                                         unsigned* __restrict__ hist, size_t P, size_t per_wg,
                                                         unsigned nc, unsigned ign) {
  __shared__ unsigned bins[BINS];
  const int n = blockIdx.y;
  bins[threadIdx.x] = 0;
  __syncthreads();
  const unsigned char* s = src + (size_t)n * P;
  float* d = dst + (size_t)n * P;
  const size_t lo = (size_t)blockIdx.x * per_wg;
  const size_t hi = lo + per_wg < P ? lo + per_wg : P;
  // the image base is only byte aligned (P is odd for 513x513): scalar head up to the next 4-byte boundary of src
  size_t i = lo;
  const size_t head = (4 - ((uintptr_t)(s + lo) & 3)) & 3;
  const size_t body = lo + head < hi ? lo + head : hi;
  for (size_t j = i + threadIdx.x; j < body; j += 256) {
    const unsigned v = remap(s[j], nc, ign);
    if (hist) atomicAdd(&bins[v & 255u], 1u);
    d[j] = (float)v;
  }
  i = body;
  const size_t quads = (hi - i) / 4;
  for (size_t qd = threadIdx.x; qd < quads; qd += 256) {
    const uchar4 u = *reinterpret_cast<const uchar4*>(s + i + qd * 4);
    const unsigned a = remap(u.x, nc, ign), b = remap(u.y, nc, ign), c = remap(u.z, nc, ign), e = remap(u.w, nc, ign);
    if (hist) {
      atomicAdd(&bins[a & 255u], 1u); atomicAdd(&bins[b & 255u], 1u);
      atomicAdd(&bins[c & 255u], 1u); atomicAdd(&bins[e & 255u], 1u);
    }
    float* o = d + i + qd * 4;                     // dst is float32 but only 4-byte aligned at an odd image base
    o[0] = (float)a; o[1] = (float)b; o[2] = (float)c; o[3] = (float)e;
  }
  for (size_t j = i + quads * 4 + threadIdx.x; j < hi; j += 256) {
    const unsigned v = remap(s[j], nc, ign);
    if (hist) atomicAdd(&bins[v & 255u], 1u);
    d[j] = (float)v;
  }
  if (!hist) return;
  __syncthreads();
  const unsigned c = bins[threadIdx.x];
  if (c) atomicAdd(&hist[(size_t)n * BINS + threadIdx.x], c);
}

// pass 2: weight table of the image (256 entries) from its histogram, then one weight per pixel
__global__ __launch_bounds__(256) void label_weight_kernel(const unsigned char* __restrict__ src, float* __restrict__ wout,
                                                           const unsigned* __restrict__ hist, size_t P, size_t per_wg,
                                                           unsigned nc, unsigned ign) {
  __shared__ float table[BINS];
  __shared__ int present[4];
  const int n = blockIdx.y;
  const unsigned cnt = hist[(size_t)n * BINS + threadIdx.x];
  const unsigned long long m = __ballot(cnt != 0);
  if ((threadIdx.x & 63) == 0) present[threadIdx.x >> 6] = __popcll(m);
  __syncthreads();
  const int k = present[0] + present[1] + present[2] + present[3];
  // sklearn: recip_freq = len(y) / (len(classes) * bincount.astype(float64)); the generator's array is float32
  table[threadIdx.x] = cnt ? (float)((double)P / ((double)k * (double)cnt)) : 0.f;
  __syncthreads();
  const unsigned char* s = src + (size_t)n * P;
  float* w = wout + (size_t)n * P;
  const size_t lo = (size_t)blockIdx.x * per_wg;
  const size_t hi = lo + per_wg < P ? lo + per_wg : P;
  for (size_t j = lo + threadIdx.x; j < hi; j += 256) w[j] = table[remap(s[j], nc, ign) & 255u];
}

}  // namespace

extern "C" int dl3p_label_prepare(const unsigned char* labels, float* labels_out, float* weights_out, uint32_t* hist,
                                  int N, size_t P, int num_classes, int ignore_index, void* stream) {
  DL3P_CHECK_ARG(labels && labels_out && N > 0 && P > 0, "dl3p_label_prepare: bad arguments");
  DL3P_CHECK_ARG(num_classes >= 1 && num_classes <= 256 && ignore_index >= 0 && ignore_index <= 255,
                 "dl3p_label_prepare: num_classes in [1, 256] and ignore_index in [0, 255] (byte labels)");
  DL3P_CHECK_ARG(!weights_out || hist, "dl3p_label_prepare: the weights need the N x 256 histogram workspace");
  hipStream_t st = (hipStream_t)stream;
  uint32_t* h = weights_out ? hist : nullptr;
  if (h && hipMemsetAsync(h, 0, (size_t)N * BINS * sizeof(unsigned), st) != hipSuccess) {
    dl3p_set_error("dl3p_label_prepare: hipMemsetAsync failed");
    return DL3P_ELAUNCH;
  }
  // enough workgroups per image to fill the chip, at least 4 K pixels each
  size_t chunks = (size_t)(2 * DL3P_NUM_CUS + N - 1) / N;
  const size_t most = (P + 4095) / 4096;
  if (chunks > most) chunks = most;
  if (chunks < 1) chunks = 1;
  const size_t per_wg = (P + chunks - 1) / chunks;
  const dim3 grid((unsigned)((P + per_wg - 1) / per_wg), (unsigned)N);
  hipLaunchKernelGGL(label_hist_kernel, grid, dim3(256), 0, st, labels, labels_out, h, P, per_wg, (unsigned)num_classes,
                     (unsigned)ignore_index);
  DL3P_CHECK_LAUNCH("dl3p_label_prepare");
  if (weights_out) {
    hipLaunchKernelGGL(label_weight_kernel, grid, dim3(256), 0, st, labels, weights_out, h, P, per_wg,
                       (unsigned)num_classes, (unsigned)ignore_index);
    DL3P_CHECK_LAUNCH("dl3p_label_prepare");
  }
  return DL3P_OK;
}
