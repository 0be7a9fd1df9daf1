// K15: y = x W^T for a handful of rows: the last convolution of the PatchGAN ladder covers its whole
// 4x4 map (reference layers/discriminator.py:110-111, Conv2d(8 ndf, ndf, 4, 1, 0) under spectral_norm), i.e. it is a
// [B, 8192] x [8192, 64] product with B = 4 .. 32 rows.  rocBLAS picks a 16x16x256 macro-tile kernel for it (16-17 us per
// call, four forward calls and eight gradient products per iteration); the problem is one pass over a 2 MB weight.
// A linear map is closed under differentiation like the convolutions of K11:
//     F  y  [M,N] = x [M,K] W[N,K]^T        D  gx [M,K] = gy [M,N] W[N,K]        G  gW [N,K] = gy[M,N]^T x[M,K]
// (backward of D wrt (gy, W) = (F, G); of G wrt (gy, x) = (F, D)).  F and G are kernels here (11.8 us / 5 us against 17 / 8 us);
// D exists as a kernel too (round 3: up to 16 rows, 64 columns x 4 row runs per workgroup; a first, column-walking attempt in
// round 2 measured 18-30 us) but rocBLAS' 5 us product stays the default: the kernel costs the iteration 0.7 % (opt-in).  Plain fp32 FMAs, fixed summation order, no atomics.
#include "tp_common.h"

namespace {
constexpr int kT = 256;
constexpr int kRows = 8;                       // rows of x handled per pass
using f32x4 = __attribute__((ext_vector_type(4))) float;

// F: one workgroup per output column n: thread t takes k = t, t + 256, ...; block tree over the 256 partial sums
__global__ __launch_bounds__(kT) void skinny_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ y,
                                                        int M, int N, int K) {
  __shared__ float red[kRows][kT];
  const int n = blockIdx.x, t = threadIdx.x;
  const float* wr = w + (size_t)n * K;
  for (int m0 = 0; m0 < M; m0 += kRows) {
    float acc[kRows];
#pragma unroll
    for (int r = 0; r < kRows; ++r) acc[r] = 0.f;
    // 16-byte loads, eight k-blocks in flight per thread (the loop is a few iterations of pure latency otherwise)
    const int K4 = (K % 4 == 0) ? K / 4 : 0;
#pragma unroll 8
    for (int k4 = t; k4 < K4; k4 += kT) {
      const f32x4 wv = *reinterpret_cast<const f32x4*>(wr + 4 * k4);
#pragma unroll
      for (int r = 0; r < kRows; ++r)
        if (m0 + r < M) {
          const f32x4 xv = *reinterpret_cast<const f32x4*>(x + (size_t)(m0 + r) * K + 4 * k4);
          acc[r] = fmaf(xv[3], wv[3], fmaf(xv[2], wv[2], fmaf(xv[1], wv[1], fmaf(xv[0], wv[0], acc[r]))));
        }
    }
    for (int k = 4 * K4 + t; k < K; k += kT) {
      const float wv = wr[k];
#pragma unroll
      for (int r = 0; r < kRows; ++r)
        if (m0 + r < M) acc[r] = fmaf(x[(size_t)(m0 + r) * K + k], wv, acc[r]);
    }
#pragma unroll
    for (int r = 0; r < kRows; ++r) red[r][t] = acc[r];
    __syncthreads();
    for (int s = kT >> 1; s > 0; s >>= 1) {
      if (t < s)
#pragma unroll
        for (int r = 0; r < kRows; ++r) red[r][t] += red[r][t + s];
      __syncthreads();
    }
    if (t < kRows && m0 + t < M) y[(size_t)(m0 + t) * N + n] = red[t][0];
    __syncthreads();
  }
}

// G: thread per k, loop over the N rows of gW: gW[n][k] = sum_m gy[m][n] x[m][k]  (x column kept in registers per pass)
__global__ __launch_bounds__(kT) void skinny_wgrad_kernel(const float* __restrict__ gy, const float* __restrict__ x, float* __restrict__ gw,
                                                          int M, int N, int K, int n_per_block) {
  extern __shared__ float g[];                 // [M][n_per_block] slice of gy
  const int k = blockIdx.x * kT + threadIdx.x, n0 = blockIdx.y * n_per_block, nn = min(n_per_block, N - n0);
  for (int e = threadIdx.x; e < M * nn; e += kT) g[e] = gy[(size_t)(e / nn) * N + n0 + e % nn];
  __syncthreads();
  if (k >= K) return;
  for (int j = 0; j < nn; ++j) {
    float acc = 0.f;
    for (int m = 0; m < M; ++m) acc = fmaf(g[m * nn + j], x[(size_t)m * K + k], acc);
    gw[(size_t)(n0 + j) * K + k] = acc;
  }
}
// D: gx[m][k] = sum_n gy[m][n] w[n][k].  A workgroup owns 64 columns k; its four waves split the N rows of w into four runs, so a
// wave reads 64 consecutive floats of one row per load (coalesced), N / 4 loads per thread, all independent and in flight together;
// gy [M,N] is read with scalar loads (the row run is wave-uniform).  The four partial sums of a column are added in wave order (fixed).  M <= kDgRows.
constexpr int kDgRows = 16;
__global__ __launch_bounds__(kT) void skinny_dgrad_kernel(const float* __restrict__ gy, const float* __restrict__ w, float* __restrict__ gx,
                                                          int M, int N, int K) {
  extern __shared__ float part[];              // partials [4][M][64]
  const int t = threadIdx.x, kc = t & 63, k = blockIdx.x * 64 + kc;
  const int wv = __builtin_amdgcn_readfirstlane(t >> 6);          // wave-uniform: gy is read with scalar loads
  const int per = (N + 3) / 4, n0 = wv * per, n1 = min(N, n0 + per);
  float acc[kDgRows];
#pragma unroll
  for (int m = 0; m < kDgRows; ++m) acc[m] = 0.f;
  if (k < K) {
    const float* wc = w + k;
#pragma unroll 16
    for (int n = n0; n < n1; ++n) {
      const float wv_ = wc[(size_t)n * K];
#pragma unroll
      for (int m = 0; m < kDgRows; ++m)
        if (m < M) acc[m] = fmaf(gy[m * N + n], wv_, acc[m]);
    }
  }
#pragma unroll
  for (int m = 0; m < kDgRows; ++m)
    if (m < M) part[(wv * M + m) * 64 + kc] = acc[m];
  __syncthreads();
  for (int e = t; e < M * 64; e += kT) {
    const int m = e >> 6, c = e & 63, kk = blockIdx.x * 64 + c;
    if (kk < K) gx[(size_t)m * K + kk] = ((part[(0 * M + m) * 64 + c] + part[(1 * M + m) * 64 + c]) + part[(2 * M + m) * 64 + c]) + part[(3 * M + m) * 64 + c];
  }
}
}  // namespace

extern "C" {
int tp_skinny_linear_dgrad(const float* gy, const float* w, float* gx, int M, int N, int K, tp_stream_t stream) {
  TP_REQUIRE(gy && w && gx && M > 0 && M <= kDgRows && N > 0 && K > 0, "bad arguments (at most 16 rows)");
  const size_t lds = (4u * M * 64u) * sizeof(float);
  hipLaunchKernelGGL(skinny_dgrad_kernel, dim3((K + 63) / 64), dim3(kT), lds, (hipStream_t)stream, gy, w, gx, M, N, K);
  return tp::check_launch("tp_skinny_linear_dgrad");
}
int tp_skinny_linear_fwd(const float* x, const float* w, float* y, int M, int N, int K, tp_stream_t stream) {
  TP_REQUIRE(x && w && y && M > 0 && N > 0 && K > 0, "bad arguments");
  hipLaunchKernelGGL(skinny_fwd_kernel, dim3(N), dim3(kT), 0, (hipStream_t)stream, x, w, y, M, N, K);
  return tp::check_launch("tp_skinny_linear_fwd");
}
int tp_skinny_linear_wgrad(const float* gy, const float* x, float* gw, int M, int N, int K, tp_stream_t stream) {
  TP_REQUIRE(gy && x && gw && M > 0 && M <= 256 && N > 0 && K > 0, "bad arguments (at most 256 rows)");
  const int kb = (K + kT - 1) / kT;
  int nsplit = (512 + kb - 1) / kb;             // aim at >= 512 workgroups
  if (nsplit > N) nsplit = N;
  if (nsplit < 1) nsplit = 1;
  const int npb = (N + nsplit - 1) / nsplit;
  hipLaunchKernelGGL(skinny_wgrad_kernel, dim3(kb, (N + npb - 1) / npb), dim3(kT), (size_t)M * npb * sizeof(float), (hipStream_t)stream, gy, x,
                     gw, M, N, K, npb);
  return tp::check_launch("tp_skinny_linear_wgrad");
}
}
