// Pointwise convolutions on a handful of rows (M <= 64): the 1x1 convs behind a global pooling -- the image-pooling
// branch of ASPP (layers.py:132-141) and the squeeze-excite bottlenecks of MobileNetV3
// (deeplabv3p_mobilenetv3.py:136-145) -- see one row per image.  The MFMA tile kernel runs them as a single 64-row
// tile walking all of K serially (50 us for 16 x 960 x 240); here the weight matrix is what gets parallelised.
//
//   pw_tiny_nt     y[m][n] = sum_k act(a[m][k]*scale[k]+shift[k]) * bt[n][k] (+ bias[n])       forward with the
//                  transposed kernel wt[N][K], and the data gradient gx[m][k] = sum_n dy[m][n] * w[k][n]
//   pw_tiny_wgrad  gw[k][n] = sum_m act(a[m][k]...) * dy[m][n],  gb[n] = sum_m dy[m][n]        written directly,
//                  no partial slabs
#include "common.h"

namespace {

struct TinyParams {
  const float* A; int lda;
  const float* scale; const float* shift; int act;
  const float* B; int ldb;          // [N][K] rows
  const float* bias;
  float* Y; int ldy;
  float* partials;                  // one partial row [2][N] (sum, sum of squares) or null
  int accumulate;
  int M, K, N;
};

constexpr int TM = 16;              // rows per pass
constexpr int TN = 4;               // output columns per workgroup

// A workgroup owns TN output columns: its 256 threads stride K in float4 steps, every thread keeps TM x TN partial dot
// products, and a transpose through LDS hands thread (wave w, lane t) the partials of threads 64w..64w+63 for output
// (m = t / TN, column t % TN); wave 0 adds the four wave sums.  Many small workgroups: the weight matrix is the only
// operand of any size (960 x 240 floats) and every workgroup streams its own 4 rows of it.
template <bool STATS>
__global__ __launch_bounds__(256) void pw_tiny_nt_kernel(TinyParams p) {
  __shared__ float red[256][TM * TN + 1];
  __shared__ float wsum[4][64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int n0 = blockIdx.x * TN;
  float s_sum = 0.f, s_sq = 0.f;
  for (int m0 = 0; m0 < p.M; m0 += TM) {
    float acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = 0.f;
    for (int k = threadIdx.x * 4; k < p.K; k += 1024) {
      float4 b[TN];
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = ld4(p.B + (size_t)(n0 + j) * p.ldb + k);
      float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = zero4();
      if (p.scale) { sc = ld4(p.scale + k); sh = ld4(p.shift + k); }
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int m = min(m0 + i, p.M - 1);             // rows past M repeat the last one and are dropped below
        const float4 a = act_apply4(fma4(ld4(p.A + (size_t)m * p.lda + k), sc, sh), p.act);
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] += a.x * b[j].x + a.y * b[j].y + a.z * b[j].z + a.w * b[j].w;
      }
    }
    __syncthreads();                                      // the previous pass has read red / wsum
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) red[threadIdx.x][i * TN + j] = acc[i][j];
    __syncthreads();
    float v = 0.f;
    for (int l = 0; l < 64; ++l) v += red[wave * 64 + l][lane];
    wsum[wave][lane] = v;
    __syncthreads();
    if (wave == 0) {
      v = (wsum[0][lane] + wsum[1][lane]) + (wsum[2][lane] + wsum[3][lane]);
      const int m = m0 + lane / TN, n = n0 + lane % TN;
      const bool ok = m < p.M;
      if (ok) {
        if (p.bias) v += p.bias[n];
        float* y = p.Y + (size_t)m * p.ldy + n;
        if (p.accumulate) v += *y;
        *y = v;
      }
      if (STATS) {
        const float u = ok ? v : 0.f;
        s_sum += u;
        s_sq += u * u;
      }
    }
  }
  if (STATS && wave == 0) {
    // lanes with the same column differ in bits 2..5
#pragma unroll
    for (int d = TN; d < 64; d <<= 1) {
      s_sum += __shfl_xor(s_sum, d);
      s_sq += __shfl_xor(s_sq, d);
    }
    if (lane < TN) {
      p.partials[n0 + lane] = s_sum;
      p.partials[p.N + n0 + lane] = s_sq;
    }
  }
}

struct TinyWgradParams {
  const float* X; int ldx;
  const float* scale; const float* shift; int act;
  const float* DY; int lddy;
  float* gw; float* gb;
  int M, K, N;
};

// one thread per (k, 4 columns): M is tiny, so there is nothing to split and nothing to reduce afterwards
__global__ __launch_bounds__(256) void pw_tiny_wgrad_kernel(TinyWgradParams p) {
  const int n4s = p.N / 4;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= p.K * n4s) return;
  const int k = idx / n4s, n = (idx - k * n4s) * 4;
  const float sc = p.scale ? p.scale[k] : 1.f, sh = p.scale ? p.shift[k] : 0.f;
  float4 acc = zero4(), colsum = zero4();
  for (int m = 0; m < p.M; ++m) {
    const float a = act_apply(p.X[(size_t)m * p.ldx + k] * sc + sh, p.act);
    const float4 g = ld4(p.DY + (size_t)m * p.lddy + n);
    acc = fma4(make_float4(a, a, a, a), g, acc);
    colsum = add4(colsum, g);
  }
  st4(p.gw + (size_t)k * p.N + n, acc);
  if (p.gb && k == 0) st4(p.gb + n, colsum);
}

int tiny_max_rows() {
  static const int v = getenv("DL3P_PW_TINY_ROWS") ? atoi(getenv("DL3P_PW_TINY_ROWS")) : 64;
  return v;
}

}  // namespace

// entry points for pwconv.hip's dispatch (same translation-unit-local checks have run there)
bool dl3p_pw_tiny_applies(int M) { return M <= tiny_max_rows(); }

void dl3p_pw_tiny_nt(const float* a, int lda, const float* scale, const float* shift, int act, const float* bt, int ldb,
                     const float* bias, float* y, int ldy, int accumulate, float* partials, int M, int K, int N,
                     hipStream_t st) {
  TinyParams p = {};
  p.A = a; p.lda = lda; p.scale = scale; p.shift = shift; p.act = act; p.B = bt; p.ldb = ldb; p.bias = bias;
  p.Y = y; p.ldy = ldy; p.partials = partials; p.accumulate = accumulate; p.M = M; p.K = K; p.N = N;
  const dim3 grid(N / TN), block(256);
  if (partials) dl3p_launch(pw_tiny_nt_kernel<true>, grid, block, 0, st, p);
  else dl3p_launch(pw_tiny_nt_kernel<false>, grid, block, 0, st, p);
}

void dl3p_pw_tiny_wgrad(const float* x, int ldx, const float* scale, const float* shift, int act, const float* dy,
                        int lddy, float* gw, float* gb, int M, int K, int N, hipStream_t st) {
  TinyWgradParams p = {};
  p.X = x; p.ldx = ldx; p.scale = scale; p.shift = shift; p.act = act; p.DY = dy; p.lddy = lddy; p.gw = gw; p.gb = gb;
  p.M = M; p.K = K; p.N = N;
  const long long threads = (long long)K * (N / 4);
  dl3p_launch(pw_tiny_wgrad_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, p);
}
