// How does v_mfma_f32_16x16x32_bf16 round when it adds its 32 products to the accumulator?  Each output row gets its own test:
// D[row][col] = C[row] + sum_k A[row][k] (B = 1), with A chosen so that the exact sum sits between two floats in a known place.
// Build: hipcc --offload-arch=gfx950 -O2 scripts/micro/mfma_round.hip -o scripts/micro/bin/mfma_round     Run on the GPU box.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <string.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
static unsigned short bf(float x) { unsigned u; memcpy(&u, &x, 4); return (unsigned short)(u >> 16); }      // exact powers of two only
__global__ void k(const unsigned short* A, const float* C, float* D, int chain) {
  const int l = threadIdx.x, r = l & 15, q = l >> 4;
  s16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (short)A[r * 32 + q * 8 + i]; b[i] = (short)0x3f80; }      // B = 1.0
  f32x4 c;
  for (int j = 0; j < 4; ++j) c[j] = C[4 * q + j];
  // operand order as in the kernels: mfma(A rows -> D rows)
  for (int it = 0; it < chain; ++it) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  for (int j = 0; j < 4; ++j) D[(4 * q + j) * 16 + r] = c[j];
}
int main() {
  unsigned short hA[16 * 32]; float hC[16]; double exact[16]; const char* what[16];
  memset(hA, 0, sizeof hA);
  for (int r = 0; r < 16; ++r) { hC[r] = 1.f; exact[r] = 1.0; what[r] = ""; }
  auto put = [&](int r, int k, float v) { hA[r * 32 + k] = bf(v); exact[r] += v; };
  what[0] = "1 + 3 x 2^-25 (0.75 ulp above 1)";              for (int k = 0; k < 3; ++k) put(0, k, ldexpf(1.f, -25));
  what[1] = "1 - 5 x 2^-26 (1.25 ulp below 1, ulp 2^-24)";   for (int k = 0; k < 5; ++k) put(1, k, -ldexpf(1.f, -26));
  what[2] = "1 + 2^-24 + 2^-26 (0.625 ulp)";                 put(2, 0, ldexpf(1.f, -24)); put(2, 1, ldexpf(1.f, -26));
  what[3] = "1 + 31 x 2^-29 + 2^-28 (33/64 ulp)";            for (int k = 0; k < 31; ++k) put(3, k, ldexpf(1.f, -29)); put(3, 31, ldexpf(1.f, -28));
  what[4] = "1 + 2^-24 exactly (tie)";                       put(4, 0, ldexpf(1.f, -24));
  what[5] = "1 + 3 x 2^-24 (tie at 1.5 ulp)";                for (int k = 0; k < 3; ++k) put(5, k, ldexpf(1.f, -24));
  what[6] = "1 + 2^-24 + 2^-40 (just over a tie)";           put(6, 0, ldexpf(1.f, -24)); put(6, 1, ldexpf(1.f, -40));
  what[7] = "1 + 2^-24 + 2^-60";                             put(7, 0, ldexpf(1.f, -24)); put(7, 1, ldexpf(1.f, -60));
  what[8] = "1 - 2^-25 - 2^-40 (just over a tie below 1)";   put(8, 0, -ldexpf(1.f, -25)); put(8, 1, -ldexpf(1.f, -40));
  what[9] = "1 + 0.5 + 2^-24 + 2^-26 (products of different size)"; put(9, 0, 0.5f); put(9, 1, ldexpf(1.f, -24)); put(9, 2, ldexpf(1.f, -26));
  what[10] = "C = 0: 1 + 3 x 2^-25 as products";             hC[10] = 0.f; exact[10] = 0.0; put(10, 0, 1.f); for (int k = 1; k < 4; ++k) put(10, k, ldexpf(1.f, -25));
  what[11] = "C = 0: 2^-24 + 2^-26 + 1 (order reversed)";    hC[11] = 0.f; exact[11] = 0.0; put(11, 0, ldexpf(1.f, -24)); put(11, 1, ldexpf(1.f, -26)); put(11, 31, 1.f);
  what[12] = "1 + 16 x 2^-27 - 15 x 2^-27 ... (cancelling small terms, net 2^-27 x 1 + 2^-24)"; for (int k = 0; k < 16; ++k) put(12, k, ldexpf(1.f, -27)); for (int k = 16; k < 31; ++k) put(12, k, -ldexpf(1.f, -27)); put(12, 31, ldexpf(1.f, -24));
  unsigned short* dA; float *dC, *dD; float hD[256];
  hipMalloc(&dA, sizeof hA); hipMalloc(&dC, sizeof hC); hipMalloc(&dD, sizeof hD);
  hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dC, hC, sizeof hC, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dC, dD, 1);
  hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
  for (int r = 0; r < 13; ++r) {
    const float got = hD[r * 16], rn = (float)exact[r];
    const float lo = nextafterf(rn, rn > (float)exact[r] ? -1e30f : 1e30f);      // the other neighbour
    printf("row %2d  %-70s exact 1%+.4f ulp(2^-23)  got 1%+.4f ulp  round-to-nearest 1%+.4f  %s\n", r, what[r], (exact[r] - 1.0) * 8388608.0,
           ((double)got - 1.0) * 8388608.0, ((double)rn - 1.0) * 8388608.0, got == rn ? "= RN" : (got == lo ? "= the OTHER neighbour" : "neither"));
  }
  // a chain: the same small products added 64 times (one MFMA per step, like a K loop): drift against the exact value
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dC, dD, 64);
  hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
  for (int r = 0; r < 4; ++r) printf("chain of 64, row %d: exact 1%+.3f ulp, got 1%+.3f ulp\n", r, (exact[r] - 1.0) * 64 * 8388608.0, ((double)hD[r * 16] - 1.0) * 8388608.0);
  return 0;
}
