// the fp32 pointwise GEMM K loop with TWO LDS operand buffers and ONE barrier per K-step: the staging of step i + 1 (registers ->
// LDS, then the request for step i + 2 into the same registers) is spread between the MFMAs of step i.  v_mfma_f32_32x32x2_f32
// (64 cycles, 8 of them holding the issue port); 128 x 128 tiles, 4 waves of 64 x 64, two workgroups per CU.
// Stand-alone: times M x K x N = 266256 x 320 x 256 and checks samples against a host float64 product.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off scripts/micro/db_loop.hip -o scripts/micro/bin/db_loop
//   FLAGS bits: 2 = no MFMAs, 4 = no LDS stores, 8 = no global loads, 16 = no epilogue stores, 32 = no barrier
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <math.h>
#include <string.h>
#include <type_traits>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BKT = 32;

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// CT: 32-column tiles per wave (BN = 64 CT)
template <int FLAGS, int CT, int OCC = 2, bool UPFRONT = false>
__global__ __launch_bounds__(256, OCC) void db(const float* __restrict__ A, const float* __restrict__ Bt, float* __restrict__ Y,
                                              int M, int K, int N, int num_m_tiles) {
  constexpr int BN = 64 * CT;
  constexpr int A_TILE = BM * 32, B_TILE = BN * 32, STAGE = A_TILE + B_TILE;      // floats; rows of 128 bytes, chunks swizzled by (row >> 1) & 7
  constexpr int NAP = BM / 32, NBP = BN / 32;                                       // passes of 32 rows (256 threads x 16 bytes)
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  float* S0 = reinterpret_cast<float*>(lds);
  const int t = threadIdx.x, l = t & 63, w = t >> 6, l31 = l & 31, h = l >> 5;
  const int wr = w >> 1, wc = w & 1;
  const int n0 = blockIdx.y * BN;
  const int nk = K / BKT;
  const int my_tiles = (num_m_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  const int T = my_tiles * nk, last = T - 1;
  if (T <= 0) return;
  const char* Ab = reinterpret_cast<const char*>(A);
  const char* Bb = reinterpret_cast<const char*>(Bt);
  const int sr = t >> 3, sc = t & 7;
  int a_lds[NAP], b_lds[NBP];
  uint32_t b_off[NBP];
#pragma unroll
  for (int i = 0; i < NAP; ++i) { const int r = sr + 32 * i; a_lds[i] = r * 32 + ((sc ^ ((r >> 1) & 7)) * 4); }
#pragma unroll
  for (int i = 0; i < NBP; ++i) {
    const int r = sr + 32 * i;
    b_lds[i] = A_TILE + r * 32 + ((sc ^ ((r >> 1) & 7)) * 4);
    b_off[i] = (uint32_t)(((long long)(n0 + r) * K + sc * 4) * 4);
  }
  int xa_off[2][4], wb_off[CT][4];
#pragma unroll
  for (int c4 = 0; c4 < 4; ++c4) {
#pragma unroll
    for (int i = 0; i < 2; ++i) { const int r = wr * 64 + i * 32 + l31; xa_off[i][c4] = r * 32 + (((2 * c4 + h) ^ ((r >> 1) & 7)) * 4); }
#pragma unroll
    for (int i = 0; i < CT; ++i) { const int c = wc * 32 * CT + i * 32 + l31; wb_off[i][c4] = A_TILE + c * 32 + (((2 * c4 + h) ^ ((c >> 1) & 7)) * 4); }
  }
  f32x4 ra[NAP], rb[NBP];
  auto prefetch_a = [&](int it, int i) __attribute__((always_inline)) {
    const int kt = it % nk, m0 = (blockIdx.x + (it / nk) * gridDim.x) * BM;
    const uint32_t off = (uint32_t)min(m0 + sr + 32 * i, M - 1) * (uint32_t)K * 4u + (uint32_t)(kt * BKT + sc * 4) * 4u;
    if (!(FLAGS & 8) || it < 2) ra[i] = *reinterpret_cast<const f32x4*>(Ab + off);
  };
  auto prefetch_b = [&](int it, int i) __attribute__((always_inline)) {
    const int kt = it % nk;
    if (!(FLAGS & 8) || it < 2) rb[i] = *reinterpret_cast<const f32x4*>(Bb + (b_off[i] + (uint32_t)(kt * BKT) * 4u));
  };
  f32x16 acc[2][CT];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < CT; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

#pragma unroll
  for (int i = 0; i < NAP; ++i) prefetch_a(0, i);
#pragma unroll
  for (int i = 0; i < NBP; ++i) prefetch_b(0, i);
#pragma unroll
  for (int i = 0; i < NAP; ++i) { *reinterpret_cast<f32x4*>(S0 + a_lds[i]) = ra[i]; prefetch_a(min(1, last), i); }
#pragma unroll
  for (int i = 0; i < NBP; ++i) { *reinterpret_cast<f32x4*>(S0 + b_lds[i]) = rb[i]; prefetch_b(min(1, last), i); }
  lds_barrier();

  f32x4 xaf[4][2], wbf[4][CT];
  auto step = [&](int it, auto par) __attribute__((always_inline)) {
    constexpr int P = decltype(par)::value;
    const float* cur = S0 + P * STAGE;
    float* nxt = S0 + (P ^ 1) * STAGE;
    const int nit = min(it + 2, last);
    constexpr int NPIECE = NAP + NBP;
    if (UPFRONT && (!(FLAGS & 64) || it < 2)) {
#pragma unroll
      for (int c4 = 0; c4 < 4; ++c4) {
#pragma unroll
        for (int i = 0; i < 2; ++i) xaf[c4][i] = *reinterpret_cast<const f32x4*>(cur + xa_off[i][c4]);
#pragma unroll
        for (int i = 0; i < CT; ++i) wbf[c4][i] = *reinterpret_cast<const f32x4*>(cur + wb_off[i][c4]);
      }
    }
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4) {
      f32x4 xa[2], wb[CT];
#pragma unroll
      for (int i = 0; i < 2; ++i) xa[i] = UPFRONT ? xaf[c4][i] : *reinterpret_cast<const f32x4*>(cur + xa_off[i][c4]);
#pragma unroll
      for (int i = 0; i < CT; ++i) wb[i] = UPFRONT ? wbf[c4][i] : *reinterpret_cast<const f32x4*>(cur + wb_off[i][c4]);
      if (!(FLAGS & 2)) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < CT; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[mi][j], wb[ni][j], acc[mi][ni], 0, 0, 0);
      } else {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < CT; ++ni) acc[mi][ni][0] += xa[mi][0] + wb[ni][1] + xa[mi][2] + wb[ni][3];
      }
      // a quarter of the next step's staging behind every quarter of the MFMAs
#pragma unroll
      for (int pc = c4 * NPIECE / 4; pc < (c4 + 1) * NPIECE / 4; ++pc) {
        if (pc < NAP) {
          if (!(FLAGS & 4)) *reinterpret_cast<f32x4*>(nxt + a_lds[pc]) = ra[pc];
          prefetch_a(nit, pc);
        } else {
          if (!(FLAGS & 4)) *reinterpret_cast<f32x4*>(nxt + b_lds[pc - NAP]) = rb[pc - NAP];
          prefetch_b(nit, pc - NAP);
        }
      }
    }
    if (!(FLAGS & 32)) lds_barrier();
    if (it % nk == nk - 1 && !(FLAGS & 16)) {
      // lane = output channel (l & 31), register e = pixel row (e & 3) + 8 (e >> 2) + 4 (l >> 5) of the 32-row tile
      const int m0 = (blockIdx.x + (it / nk) * gridDim.x) * BM;
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < CT; ++ni) {
          const int mrow = m0 + wr * 64 + mi * 32 + 4 * h;
          float* yb = Y + (size_t)mrow * N + n0 + wc * 32 * CT + ni * 32 + l31;
          if (mrow + 28 < M) {
#pragma unroll
            for (int e = 0; e < 16; ++e) { yb[(size_t)((e & 3) + 8 * (e >> 2)) * N] = acc[mi][ni][e]; acc[mi][ni][e] = 0.f; }
          } else {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
              const int row = (e & 3) + 8 * (e >> 2);
              if (mrow + row < M) yb[(size_t)row * N] = acc[mi][ni][e];
              acc[mi][ni][e] = 0.f;
            }
          }
        }
    }
  };
  int it = 0;
  for (; it + 1 < T; it += 2) { step(it, std::integral_constant<int, 0>{}); step(it + 1, std::integral_constant<int, 1>{}); }
  if (it < T) step(it, std::integral_constant<int, 0>{});
  if ((FLAGS & 16) && num_m_tiles < 0) {
    float sum = 0.f;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < CT; ++ni)
#pragma unroll
        for (int e = 0; e < 16; ++e) sum += acc[mi][ni][e];
    Y[t] = sum;
  }
}

template <int FLAGS, int CT, int OCC = 2, bool UPFRONT = false>
static float run(const float* A, const float* B, float* Y, int M, int K, int N, const char* name) {
  constexpr int BN = 64 * CT;
  const int mt = (M + BM - 1) / BM;
  const int gy = N / BN, gxm = 256 * OCC / gy;
  const int gx = (mt + ((mt + gxm - 1) / gxm) - 1) / ((mt + gxm - 1) / gxm);
  const int ldsb = 2 * (BM + BN) * 32 * 4;
  hipFuncSetAttribute((const void*)db<FLAGS, CT, OCC, UPFRONT>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f;
  for (int r = 0; r < 6; ++r) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((db<FLAGS, CT, OCC, UPFRONT>), dim3(gx, gy), dim3(256), ldsb, 0, A, B, Y, M, K, N, mt);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (r > 0 && ms < best) best = ms;
  }
  printf("%-52s %7.1f us   %.1f TFLOP/s   grid %dx%d lds %d\n", name, best * 1e3, 2.0 * M * K * N / best / 1e9, gx, gy, ldsb);
  return best;
}

static void check(const std::vector<float>& hA, const std::vector<float>& hW, const float* Y, int M, int K, int N) {
  std::vector<float> hY((size_t)M * N);
  hipMemcpy(hY.data(), Y, hY.size() * 4, hipMemcpyDeviceToHost);
  double worst = 0, scale = 0;
  for (int s = 0; s < 4000; ++s) {
    const int m = s < 200 ? M - 1 - s : (int)(((long long)s * 7919 * 131) % M), n = (s * 37) % N;
    double ref = 0;
    for (int k = 0; k < K; ++k) ref += (double)hA[(size_t)m * K + k] * (double)hW[(size_t)n * K + k];
    worst = fmax(worst, fabs(ref - hY[(size_t)m * N + n])); scale = fmax(scale, fabs(ref));
  }
  printf("max |err| / max |ref| over 4000 samples: %.2e\n", worst / scale);
}

int main() {
  const int M = 266256, K = 320, N = 256;
  std::vector<float> hA((size_t)M * K), hW((size_t)N * K);
  srand(1);
  for (auto& v : hA) v = (rand() / (float)RAND_MAX - 0.5f) * 4.f;
  for (auto& v : hW) v = (rand() / (float)RAND_MAX - 0.5f) * 0.2f;
  float *A, *Y, *B;
  hipMalloc(&A, hA.size() * 4); hipMalloc(&Y, (size_t)M * N * 4); hipMalloc(&B, hW.size() * 4);
  hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(B, hW.data(), hW.size() * 4, hipMemcpyHostToDevice);
  run<0, 2>(A, B, Y, M, K, N, "128 x 128 tiles");
  check(hA, hW, Y, M, K, N);
  hipMemset(Y, 0, (size_t)M * N * 4);
  run<0, 1>(A, B, Y, M, K, N, "128 x 64 tiles");
  check(hA, hW, Y, M, K, N);
  run<16, 2>(A, B, Y, M, K, N, "128 x 128, no epilogue stores");
  run<2 | 16, 2>(A, B, Y, M, K, N, "128 x 128, no MFMAs, no stores");
  run<4 | 8 | 16, 2>(A, B, Y, M, K, N, "128 x 128, MFMAs + fragment reads only");
  run<4 | 8 | 16 | 32, 2>(A, B, Y, M, K, N, "  ... and no barrier");
  run<16, 1>(A, B, Y, M, K, N, "128 x 64, no epilogue stores");
  run<4 | 8 | 16, 1>(A, B, Y, M, K, N, "128 x 64, MFMAs + fragment reads only");
  run<0, 1, 3>(A, B, Y, M, K, N, "128 x 64, three workgroups per CU");
  check(hA, hW, Y, M, K, N);
  run<0, 2, 2, true>(A, B, Y, M, K, N, "128 x 128, fragments up front");
  check(hA, hW, Y, M, K, N);
  run<0, 1, 2, true>(A, B, Y, M, K, N, "128 x 64, fragments up front");
  run<0, 1, 3, true>(A, B, Y, M, K, N, "128 x 64, three per CU, fragments up front");
  run<4 | 8 | 16 | 32, 2, 2, true>(A, B, Y, M, K, N, "128 x 128, up front, MFMA + fragments, no barrier");
  run<4 | 8 | 16 | 32, 2, 1, true>(A, B, Y, M, K, N, "128 x 128, up front, one WG per CU, MFMA + fragments, no barrier");
  run<4 | 8 | 16 | 32 | 64, 2, 1, true>(A, B, Y, M, K, N, "128 x 128, one WG per CU, MFMA from registers only");
  run<4 | 8 | 16 | 32 | 64, 2, 2, true>(A, B, Y, M, K, N, "128 x 128, two WG per CU, MFMA from registers only");
  return 0;
}
