// raw f32 MFMA issue-rate microbenchmark (register-only): 16x16x4 vs 32x32x2, 1 or 2 waves per SIMD
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ void k16(float* out, int iters, float a0, float b0) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0, 0, 0, 0};
  float a = a0 + threadIdx.x, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
__global__ void k32(float* out, int iters, float a0, float b0) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0;
  float a = a0 + threadIdx.x, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][15];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename F> float run(F f, const char* name, double flops) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f(); hipDeviceSynchronize();
  hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-40s %.3f ms  %.1f TFLOP/s\n", name, ms, flops / ms / 1e9);
  return ms;
}
int main() {
  float* out; hipMalloc(&out, 256 * 8 * 512 * 4);
  const int iters = 4000;
  for (int wps = 1; wps <= 2; ++wps) {
    dim3 grid(256), block(256 * wps);   // 4*wps waves per CU
    double n16 = (double)256 * 4 * wps * iters * 4 * 16 * (2.0 * 16 * 16 * 4);
    double n32 = (double)256 * 4 * wps * iters * 4 * 4 * (2.0 * 32 * 32 * 2);
    char nm[64];
    snprintf(nm, 64, "16x16x4 16 acc, %d wave/SIMD", wps);
    run([&] { hipLaunchKernelGGL(k16<16>, grid, block, 0, 0, out, iters, 1.f, 2.f); }, nm, n16);
    snprintf(nm, 64, "32x32x2 4 acc, %d wave/SIMD", wps);
    run([&] { hipLaunchKernelGGL(k32<4>, grid, block, 0, 0, out, iters, 1.f, 2.f); }, nm, n32);
  }
  return 0;
}
