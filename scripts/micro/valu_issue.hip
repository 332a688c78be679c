// how many cycles does a SIMD spend per vector instruction with 1, 2, 4, 8 resident waves?  (independent v_fma_f32 chains, dependent chain,
// v_pk_fma_f32, v_fma beside v_mfma_f32_16x16x4_f32)      hipcc --offload-arch=gfx950 -O3 valu_issue.hip -o valu_issue
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void k(float* out, int iters, long long* cyc) {
  float a[8];
  for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 0.001f + i;
  f2 p[4] = {{1.f, 2.f}, {3.f, 4.f}, {5.f, 6.f}, {7.f, 8.f}};
  f4 acc = {0.f, 0.f, 0.f, 0.f}, acc1 = acc, acc2 = acc, acc3 = acc;
  typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
  bf8 ba, bb;
  for (int i = 0; i < 8; ++i) { ba[i] = (__bf16)(threadIdx.x * 0.01f + i); bb[i] = (__bf16)(i * 0.5f); }
  const float b = 1.0001f, c = 0.5f;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {           // 8 independent chains: 32 fma per iteration
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
    } else if (MODE == 1) {    // one dependent chain: 32 fma
#pragma unroll
      for (int r = 0; r < 32; ++r) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(b), "v"(c));
    } else if (MODE == 2) {    // packed: 32 v_pk_fma_f32 on 4 independent pairs
#pragma unroll
      for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(p[(i + 1) & 3]), "v"(p[(i + 2) & 3]));
    } else if (MODE == 4) {    // 16 mfma 16x16x4 f32 on 4 independent accumulators, nothing else
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[7], b, acc, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[6], b, acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[5], b, acc2, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4], b, acc3, 0, 0, 0);
      }
    } else if (MODE == 5) {    // 4 mfma 16x16x32 bf16 + 32 independent fma
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ba, bb, acc, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i & 3]) : "v"(b), "v"(c));
      }
    } else if (MODE == 6) {    // 16 mfma 16x16x32 bf16 on 4 accumulators, nothing else
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ba, bb, acc, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ba, bb, acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ba, bb, acc2, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ba, bb, acc3, 0, 0, 0);
      }
    } else {                   // 4 mfma 16x16x4 f32 + 32 independent fma
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[7], b, acc, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i & 3]) : "v"(b), "v"(c));
      }
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  float s = acc[0] + acc[1] + acc1[0] + acc2[1] + acc3[2] + p[0][0] + p[1][1] + p[2][0] + p[3][1];
  for (int i = 0; i < 8; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE>
void run(const char* name, int insts) {
  float* out; long long* cyc;
  hipMalloc(&out, 4 * 256 * 2048 * 8); hipMalloc(&cyc, 8 * 256 * 8);
  for (int wps = 1; wps <= 8; wps *= 2) {      // waves per SIMD: blocks of 256 threads (4 waves, one per SIMD), wps blocks per CU
    const int blocks = 256 * wps, iters = 2000;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, cyc);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    static long long h[2048]; hipMemcpy(h, cyc, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
    double mean = 0; for (int i = 0; i < blocks; ++i) mean += h[i]; mean /= blocks;
    // per SIMD: wps waves x iters x insts instructions in `mean` cycles (waves run concurrently)
    printf("%-28s waves/SIMD %d: %.2f cycles per instruction per SIMD (wave view %.2f), kernel %.3f ms\n", name, wps,
           mean / ((double)wps * iters * insts), mean / ((double)iters * insts), ms);
  }
  hipFree(out); hipFree(cyc);
}
int main() {
  run<0>("32 independent v_fma_f32", 32);
  run<1>("32 dependent v_fma_f32", 32);
  run<2>("32 v_pk_fma_f32", 32);
  run<3>("4 mfma16x16x4f32 + 32 fma", 36);
  run<4>("16 mfma16x16x4f32", 16);
  run<5>("4 mfma16x16x32bf16 + 32 fma", 36);
  run<6>("16 mfma16x16x32bf16", 16);
  return 0;
}
