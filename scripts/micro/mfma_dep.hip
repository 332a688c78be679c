// does a chain of DEPENDENT v_mfma_f32_16x16x4_f32 (same accumulator back to back) issue at the full rate?
// NACC independent accumulators, the inner loop walks them round-robin; 1 or 2 waves per SIMD
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void k16(float* out, int iters, float a0, float b0) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0, 0, 0, 0};
  float a = a0 + threadIdx.x, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 16 / NACC * 4; ++j)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename F> void run(F f, const char* name, double flops) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f(); hipDeviceSynchronize();
  hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-44s %.3f ms  %.1f TFLOP/s\n", name, ms, flops / ms / 1e9);
}
int main() {
  float* out; hipMalloc(&out, 256 * 8 * 512 * 4);
  const int iters = 4000;
  for (int wps = 1; wps <= 4; wps *= 2) {
    dim3 grid(256), block(256 * wps);
    double n = (double)256 * 4 * wps * iters * 64 * (2.0 * 16 * 16 * 4);
    char nm[64];
#define R(N) snprintf(nm, 64, "%d accumulator(s), %d wave(s)/SIMD", N, wps); run([&] { hipLaunchKernelGGL(k16<N>, grid, block, 0, 0, out, iters, 1.f, 2.f); }, nm, n);
    R(1) R(2) R(4) R(8) R(16)
  }
  return 0;
}
