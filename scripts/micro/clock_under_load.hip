// the shader clock the chip actually sustains under a chip-wide matrix / vector load: s_memtime (core clock ticks) against
// s_memrealtime (a constant 100 MHz counter) around a long loop, every CU busy.  The "peak" of an MFMA roofline is quoted at the
// nominal 2.4 GHz; what a kernel can reach is that times (sustained clock / 2.4).
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/clock_under_load.hip -o scripts/micro/bin/clock_under_load
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

template <int KIND>
__global__ __launch_bounds__(256) void load_kernel(float* out, unsigned long long* ticks, int iters, float seed) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  const float a = seed + threadIdx.x * 0.37f, b = seed * 1.7f - threadIdx.x * 0.11f;
  s16x8 xa, xb;
  for (int e = 0; e < 8; ++e) { xa[e] = (short)(0x3f80 + ((threadIdx.x * 37 + e * 11) & 0x7f)); xb[e] = (short)(0xbf00 + ((threadIdx.x * 13 + e * 7) & 0x7f)); }
  float v0 = a, v1 = b, v2 = a + b, v3 = a - b;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      if (KIND == 0) acc[j & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j & 3], 0, 0, 0);
      if (KIND == 1) acc[j & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa, xb, acc[j & 3], 0, 0, 0);
      if (KIND == 2) { v0 = fmaf(v0, 1.0001f, v1); v1 = fmaf(v1, 0.9999f, v2); v2 = fmaf(v2, 1.0002f, v3); v3 = fmaf(v3, 0.9998f, v0); }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = v0 + v1 + v2 + v3;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][15];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) { ticks[blockIdx.x * 2] = t1 - t0; ticks[blockIdx.x * 2 + 1] = r1 - r0; }
}

template <int KIND>
static void run(const char* name, int iters, double flops_per_iter_per_wave) {
  const int grid = 256 * 2;
  float* out; unsigned long long* ticks;
  hipMalloc(&out, grid * 256 * 4); hipMalloc(&ticks, grid * 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int r = 0; r < 3; ++r) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(load_kernel<KIND>, dim3(grid), dim3(256), 0, 0, out, ticks, iters, 1.f + r);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(grid * 2);
    hipMemcpy(h.data(), ticks, grid * 16, hipMemcpyDeviceToHost);
    std::vector<double> mhz;
    for (int i = 0; i < grid; ++i) mhz.push_back((double)h[2 * i] / (double)h[2 * i + 1] * 100.0);
    std::sort(mhz.begin(), mhz.end());
    printf("%-30s run %d: %8.3f ms  shader clock min %.0f median %.0f max %.0f MHz", name, r, ms, mhz.front(), mhz[grid / 2], mhz.back());
    if (flops_per_iter_per_wave > 0) printf("   %.1f TFLOP/s", flops_per_iter_per_wave * iters * grid * 4 / ms / 1e9);
    printf("\n");
  }
  hipFree(out); hipFree(ticks);
}

int main() {
  run<2>("v_fma_f32 chain (VALU)", 200000, 0);
  run<0>("v_mfma_f32_32x32x2_f32", 40000, 16.0 * 32 * 32 * 2 * 2);
  run<1>("v_mfma_f32_32x32x16_bf16", 80000, 16.0 * 32 * 32 * 16 * 2);
  run<0>("v_mfma_f32_32x32x2_f32 (again)", 200000, 16.0 * 32 * 32 * 2 * 2);
  return 0;
}
