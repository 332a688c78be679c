// the split-bf16 GEMM K loop with ONE wave per SIMD and the staging of K-step i + 1 interleaved, instruction by instruction, with the
// MFMAs of K-step i (two LDS operand buffers, one barrier per step).  Stand-alone: times M x K x N = 266256 x 320 x 256 (the decoder
// shape, K padded to the tile) and checks a few outputs against a host float64 product.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off scripts/micro/sb2_loop.hip -o scripts/micro/bin/sb2_loop
//   FLAGS bits: 1 = sched_group_barrier pipeline, 2 = no MFMAs, 4 = no staging arithmetic/LDS stores (after the first step),
//   8 = no global loads (after the first steps), 16 = no epilogue stores, 32 = no barrier in the loop, 64 = fragments read once
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
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));

constexpr int BM = 128, BN = 128, BKT = 32;
constexpr int A_PLANE = BM * 32, B_PLANE = BN * 32;       // bf16 elements; rows of 64 bytes, 16-byte chunks swizzled by (row >> 2) & 3
constexpr int STAGE = 3 * (A_PLANE + B_PLANE);

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ void split2(float x, float y, uint32_t& h, uint32_t& m, uint32_t& l) {
  const bf16x2v hv = {(__bf16)x, (__bf16)y};
  h = __builtin_bit_cast(uint32_t, hv);
  const float rx = x - __builtin_bit_cast(float, h << 16);
  const float ry = y - __builtin_bit_cast(float, h & 0xffff0000u);
  const bf16x2v mv = {(__bf16)rx, (__bf16)ry};
  m = __builtin_bit_cast(uint32_t, mv);
  const float sx = rx - __builtin_bit_cast(float, m << 16);
  const float sy = ry - __builtin_bit_cast(float, m & 0xffff0000u);
  const bf16x2v lv = {(__bf16)sx, (__bf16)sy};
  l = __builtin_bit_cast(uint32_t, lv);
}

template <int FLAGS, int NW>
__global__ __launch_bounds__(64 * NW, 1) void sb2(const float* __restrict__ A, const unsigned short* __restrict__ Bsp, float* __restrict__ Y,
                                               int M, int K, int N, int num_m_tiles) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned short* S0 = reinterpret_cast<unsigned short*>(lds);
  const int t = threadIdx.x, l = t & 63, w = t >> 6, l15 = l & 15, q = l >> 4;
  constexpr int NTHR = 64 * NW, MT = NW == 4 ? 2 : 1, NAC = 512 / NTHR, NBC = 1536 / NTHR, RPP = NTHR / 4;
  const int wr = w >> 1, wc = w & 1;
  const int n0 = blockIdx.y * BN;
  const int nk = K / BKT;
  const int my_tiles = (num_m_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  const int T = my_tiles * nk, last = T - 1;
  const char* Ab = reinterpret_cast<const char*>(A);
  const char* Bb = reinterpret_cast<const char*>(Bsp);
  const int ar = t >> 2, ac = t & 3;
  // LDS store offsets (bf16 elements) of this thread's chunks
  int a_lds[NAC];
#pragma unroll
  for (int i = 0; i < NAC; ++i) { const int r = ar + RPP * i; a_lds[i] = r * 32 + ((ac ^ ((r >> 2) & 3)) * 8); }
  int b_lds[NBC];
  uint32_t b_off[NBC];
#pragma unroll
  for (int i = 0; i < NBC; ++i) {
    const int idx = t + NTHR * i, plane = idx / (BN * 4), rem = idx - plane * (BN * 4), r = rem >> 2, ch = rem & 3;
    b_lds[i] = 3 * A_PLANE + plane * B_PLANE + r * 32 + ((ch ^ ((r >> 2) & 3)) * 8);
    b_off[i] = (uint32_t)(((long long)plane * N * K + (long long)(n0 + r) * K + ch * 8) * 2);
  }
  // fragment read offsets (32x32x16: lane = row (l & 31), 8 k at chunk 2 kh + (l >> 5))
  const int l31 = l & 31, hh5 = l >> 5;
  int xa_off[2][2], wb_off[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
      const int r = wr * 32 * MT + (i % MT) * 32 + l31; xa_off[i][kh] = r * 32 + (((2 * kh + hh5) ^ ((r >> 2) & 3)) * 8);
      const int c = wc * 64 + i * 32 + l31; wb_off[i][kh] = 3 * A_PLANE + c * 32 + (((2 * kh + hh5) ^ ((c >> 2) & 3)) * 8);
    }
  float4 ra[2][NAC][2];
  u32x4 rb[2][NBC];
  auto prefetch_a = [&](int it, auto par, int i) __attribute__((always_inline)) {
    constexpr int P = decltype(par)::value;
    const int kt = it % nk, m0 = (blockIdx.x + (it / nk) * gridDim.x) * BM;
    const uint32_t rowb = (uint32_t)min(m0 + ar + RPP * i, M - 1) * (uint32_t)K * 4u + (uint32_t)(kt * BKT + ac * 8) * 4u;
    ra[P][i][0] = *reinterpret_cast<const float4*>(Ab + rowb);
    ra[P][i][1] = *reinterpret_cast<const float4*>(Ab + rowb + 16);
  };
  auto prefetch_b = [&](int it, auto par, int i) __attribute__((always_inline)) {
    constexpr int P = decltype(par)::value;
    const int kt = it % nk;
    rb[P][i] = *reinterpret_cast<const u32x4*>(Bb + (b_off[i] + (uint32_t)(kt * BKT) * 2u));
  };
  auto stage_a = [&](unsigned short* buf, auto par, int i) __attribute__((always_inline)) {
    constexpr int P = decltype(par)::value;
    const float4 v0 = ra[P][i][0], v1 = ra[P][i][1];
    uint4 hh, mm, ll;
    split2(v0.x, v0.y, hh.x, mm.x, ll.x);
    split2(v0.z, v0.w, hh.y, mm.y, ll.y);
    split2(v1.x, v1.y, hh.z, mm.z, ll.z);
    split2(v1.z, v1.w, hh.w, mm.w, ll.w);
    unsigned short* d = buf + a_lds[i];
    *reinterpret_cast<uint4*>(d) = hh;
    *reinterpret_cast<uint4*>(d + A_PLANE) = mm;
    *reinterpret_cast<uint4*>(d + 2 * A_PLANE) = ll;
  };
  auto stage_b = [&](unsigned short* buf, auto par, int i) __attribute__((always_inline)) {
    constexpr int P = decltype(par)::value;
    *reinterpret_cast<u32x4*>(buf + b_lds[i]) = rb[P][i];
  };
  f32x16 acc[MT][2];
#pragma unroll
  for (int mi = 0; mi < MT; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

  s16x8 xa[MT][2][3], wb[2][2][3];
  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;
  if (T <= 0) return;
  // steps 0 and 1 requested, step 0 staged, step 2 requested into the set just consumed
#pragma unroll
  for (int i = 0; i < NAC; ++i) prefetch_a(0, P0{}, i);
#pragma unroll
  for (int i = 0; i < NBC; ++i) prefetch_b(0, P0{}, i);
#pragma unroll
  for (int i = 0; i < NAC; ++i) prefetch_a(min(1, last), P1{}, i);
#pragma unroll
  for (int i = 0; i < NBC; ++i) prefetch_b(min(1, last), P1{}, i);
#pragma unroll
  for (int i = 0; i < NAC; ++i) { stage_a(S0, P0{}, i); prefetch_a(min(2, last), P0{}, i); }
#pragma unroll
  for (int i = 0; i < NBC; ++i) { stage_b(S0, P0{}, i); prefetch_b(min(2, last), P0{}, i); }
  lds_barrier();

  auto step = [&](int it, auto par) __attribute__((always_inline)) {
    constexpr int P = decltype(par)::value;
    using Q = std::integral_constant<int, P ^ 1>;
    const unsigned short* cur = S0 + P * STAGE;
    unsigned short* nxt = S0 + (P ^ 1) * STAGE;
    const int nit = min(it + 3, last);
    constexpr bool do_stage = !(FLAGS & 4), do_load = !(FLAGS & 8);
    // all 24 fragments of the step up front (96 registers; one wave per SIMD has 512)
    if (!(FLAGS & 64) || it < 2)
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
          if (i < MT) xa[i % MT][kh][pl] = *reinterpret_cast<const s16x8*>(cur + pl * A_PLANE + xa_off[i][kh]);
          wb[i][kh][pl] = *reinterpret_cast<const s16x8*>(cur + pl * B_PLANE + wb_off[i][kh]);
        }
    // groups of 6 MFMAs (k half, row tile, column tile); a piece of the next step's staging behind each
    constexpr int NG = 4 * MT;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const int kh = g / (2 * MT), mi = (g >> 1) % MT, ni = g & 1;
      if (!(FLAGS & 2)) {
        constexpr int WB[6] = {2, 0, 1, 1, 0, 0}, XA[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
        for (int pr = 0; pr < 6; ++pr)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[mi][kh][XA[pr]], wb[ni][kh][WB[pr]], acc[mi][ni], 0, 0, 0);
      } else {
        acc[mi][ni][0] += __builtin_bit_cast(float, (int)wb[ni][kh][0][0] + (int)wb[ni][kh][1][1] + (int)wb[ni][kh][2][2] + (int)xa[mi][kh][0][0] + (int)xa[mi][kh][1][1] + (int)xa[mi][kh][2][2]);
      }
      if (g < NAC) { if (do_stage) stage_a(nxt, Q{}, g); if (do_load) prefetch_a(nit, Q{}, g); }
      else if (g - NAC < NBC) {
        if (do_stage) stage_b(nxt, Q{}, g - NAC);
        if (do_load) prefetch_b(nit, Q{}, g - NAC);
      }
    }
    if (FLAGS & 1) {
      // 48 MFMAs of 32 cycles, 8 of them holding the vector issue: up to ~5 other instructions per gap
#pragma unroll
      for (int g = 0; g < 12 * MT * 2; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);      // VALU
        if (g % 4 == 1) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);   // DS write
        if (g % 4 == 3) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // VMEM read
      }
    }
    if (!(FLAGS & 32)) lds_barrier();
    if (it % nk == nk - 1 && !(FLAGS & 16)) {
      // lane = output channel (l & 31), register e = pixel row (e & 3) + 8 (e >> 2) + 4 (l >> 5) of the 32-row tile
      const int m0 = (blockIdx.x + (it / nk) * gridDim.x) * BM;
#pragma unroll
      for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          float* yb = Y + (size_t)(m0 + wr * 32 * MT + mi * 32 + 4 * hh5) * N + n0 + wc * 64 + ni * 32 + l31;
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int row = (e & 3) + 8 * (e >> 2);
            if (m0 + wr * 32 * MT + mi * 32 + 4 * hh5 + row < M) yb[(size_t)row * N] = acc[mi][ni][e];
            acc[mi][ni][e] = 0.f;
          }
        }
    }
  };
  int it = 0;
  for (; it + 1 < T; it += 2) { step(it, P0{}); step(it + 1, P1{}); }
  if (it < T) step(it, P0{});
  if ((FLAGS & 16) && num_m_tiles < 0) {      // never true: keeps the accumulators alive in the store-less variants
    float sum = 0.f;
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int e = 0; e < 16; ++e) sum += acc[mi][ni][e];
    Y[t] = sum;
  }
}

static void split_host(float x, unsigned short* h, unsigned short* m, unsigned short* l) {
  auto rn = [](float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (unsigned short)(u >> 16); };
  auto up = [](unsigned short b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; };
  *h = rn(x); float r = x - up(*h); *m = rn(r); r -= up(*m); *l = rn(r);
}

template <int FLAGS, int NW = 4>
static float run(const float* A, const unsigned short* B, float* Y, int M, int K, int N, const char* name) {
  const int mt = (M + BM - 1) / BM;
  const int gy = N / BN, gxm = 256 / gy;
  const int gx = (mt + ((mt + gxm - 1) / gxm) - 1) / ((mt + gxm - 1) / gxm);
  const int ldsb = 2 * STAGE * 2;
  hipFuncSetAttribute((const void*)sb2<FLAGS, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f;
  for (int r = 0; r < 6; ++r) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((sb2<FLAGS, NW>), dim3(gx, gy), dim3(64 * NW), ldsb, 0, A, B, Y, M, K, N, mt);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (r > 0 && ms < best) best = ms;
  }
  const double steps_per_cu = (double)mt * gy * (K / BKT) / 256.0;
  printf("%-52s %7.1f us   %.0f clk/K-step at 2.4 GHz (MFMA floor 1536)  grid %dx%d\n", name, best * 1e3, best * 1e-3 * 2.4e9 / steps_per_cu, gx, gy);
  return best;
}

int main() {
  const int M = 266256, K = 320, N = 256;
  std::vector<float> hA((size_t)M * K), hW((size_t)N * K);
  srand(1);
  for (auto& v : hA) v = (rand() / (float)RAND_MAX - 0.5f) * 4.f;
  for (auto& v : hW) v = (rand() / (float)RAND_MAX - 0.5f) * 0.2f;
  std::vector<unsigned short> hB((size_t)3 * N * K);
  for (size_t i = 0; i < (size_t)N * K; ++i) split_host(hW[i], &hB[i], &hB[(size_t)N * K + i], &hB[(size_t)2 * N * K + i]);
  float *A, *Y; unsigned short* B;
  hipMalloc(&A, hA.size() * 4); hipMalloc(&Y, (size_t)M * N * 4); hipMalloc(&B, hB.size() * 2);
  hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(B, hB.data(), hB.size() * 2, hipMemcpyHostToDevice);
  run<0>(A, B, Y, M, K, N, "compiler's own order");
  {
    std::vector<float> hY((size_t)M * N);
    hipMemcpy(hY.data(), Y, hY.size() * 4, hipMemcpyDeviceToHost);
    double worst = 0, scale = 0;
    for (int s = 0; s < 4000; ++s) {
      const int m = (int)(((long long)s * 7919 * 131) % M), n = (s * 37) % N;
      double ref = 0;
      for (int k = 0; k < K; ++k) ref += (double)hA[(size_t)m * K + k] * (double)hW[(size_t)n * K + k];
      worst = fmax(worst, fabs(ref - hY[(size_t)m * N + n])); scale = fmax(scale, fabs(ref));
    }
    printf("max |err| / max |ref| over 4000 samples: %.2e\n", worst / scale);
  }
  run<1>(A, B, Y, M, K, N, "sched_group_barrier pipeline");
  {
    std::vector<float> hY((size_t)M * N);
    hipMemcpy(hY.data(), Y, hY.size() * 4, hipMemcpyDeviceToHost);
    double worst = 0, scale = 0;
    for (int s = 0; s < 4000; ++s) {
      const int m = (int)(((long long)s * 7919 * 131) % M), n = (s * 37) % N;
      double ref = 0;
      for (int k = 0; k < K; ++k) ref += (double)hA[(size_t)m * K + k] * (double)hW[(size_t)n * K + k];
      worst = fmax(worst, fabs(ref - hY[(size_t)m * N + n])); scale = fmax(scale, fabs(ref));
    }
    printf("max |err| / max |ref| over 4000 samples: %.2e\n", worst / scale);
  }
  run<2>(A, B, Y, M, K, N, "no MFMAs");
  run<1 | 4>(A, B, Y, M, K, N, "pipeline, no staging arithmetic / LDS stores");
  run<1 | 8>(A, B, Y, M, K, N, "pipeline, no global loads");
  run<1 | 4 | 8>(A, B, Y, M, K, N, "pipeline, MFMAs + fragment reads only");
  run<4 | 8>(A, B, Y, M, K, N, "own order, MFMAs + fragment reads only");
  run<4 | 8 | 16>(A, B, Y, M, K, N, "  ... and no epilogue stores");
  run<4 | 8 | 16 | 32>(A, B, Y, M, K, N, "  ... and no barrier");
  run<4 | 8 | 16 | 32 | 64>(A, B, Y, M, K, N, "  ... and fragments from registers (MFMA only)");
  run<16>(A, B, Y, M, K, N, "everything but the epilogue stores");
  run<2 | 16>(A, B, Y, M, K, N, "no MFMAs, no epilogue stores");
  printf("-- eight waves (two per SIMD)\n");
  run<0, 8>(A, B, Y, M, K, N, "compiler's own order");
  {
    std::vector<float> hY((size_t)M * N);
    hipMemcpy(hY.data(), Y, hY.size() * 4, hipMemcpyDeviceToHost);
    double worst = 0, scale = 0;
    for (int s = 0; s < 4000; ++s) {
      const int m = (int)(((long long)s * 7919 * 131) % M), n = (s * 37) % N;
      double ref = 0;
      for (int k = 0; k < K; ++k) ref += (double)hA[(size_t)m * K + k] * (double)hW[(size_t)n * K + k];
      worst = fmax(worst, fabs(ref - hY[(size_t)m * N + n])); scale = fmax(scale, fabs(ref));
    }
    printf("max |err| / max |ref| over 4000 samples: %.2e\n", worst / scale);
  }
  run<16, 8>(A, B, Y, M, K, N, "everything but the epilogue stores");
  run<2 | 16, 8>(A, B, Y, M, K, N, "no MFMAs, no epilogue stores");
  run<4 | 8 | 16, 8>(A, B, Y, M, K, N, "MFMAs + fragment reads only, no stores");
  run<4 | 8 | 16 | 32 | 64, 8>(A, B, Y, M, K, N, "MFMA only");
  return 0;
}
