// K9: InstanceNorm2d (affine = False) + LeakyReLU as ONE launch per derivative order.
// reference layers/discriminator.py:94-115: every stride-2 SN-conv of the PatchGAN ladder is followed by
// nn.InstanceNorm2d(c) and nn.LeakyReLU(0.2).  PyTorch runs that pair as batch-norm kernels + elementwise kernels, and
// its DOUBLE backward -- needed by the R1 penalty, model/nerf_adapt_st_gan.py:794-807 -- as ~40 elementwise / reduction
// launches per call.  Here one wavefront owns one (image, channel) instance of hw values:
//     xhat = (x - mean) * rstd,  rstd = 1 / sqrt(var + eps)  (biased variance),   y = xhat * s,  s = xhat > 0 ? 1 : slope
//   backward        a = gy * s;   gx = rstd * P(a),   P(v) = v - mean(v) - xhat * mean(v * xhat)
//   double backward (cotangent u of gx):  d/d gy = s * rstd * P(u)
//                   d/d x  = -(rstd^2 / n) xhat (A - n C D) - rstd^2 (D (u - mean u) + C (a - mean a) - 2 C D xhat),
//                            A = sum(u a) - n mean(u) mean(a),  C = mean(u xhat),  D = mean(a xhat)
// (derived with d rstd / d x_i = -(rstd^2 / n) xhat_i, d xhat_j / d x_i = rstd (delta_ij - 1/n) - (rstd / n) xhat_j xhat_i;
// checked against torch autograd in fp64, tests/test_host_logic_cpu.py).  Fixed-order wave reductions: deterministic.
#include "tp_common.h"

namespace {
constexpr int kWave = 64;
constexpr int kMaxPerLane = 64;                    // hw <= 4096 (a 64x64 map): values of an instance stay in registers

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, kWave);
  return v;
}

__global__ __launch_bounds__(256) void inorm_lrelu_fwd_kernel(const float* __restrict__ x, int64_t n_inst, int hw, float eps,
                                                               float slope, float* __restrict__ y, float* __restrict__ xhat,
                                                               float* __restrict__ rstd) {
  const int lane = threadIdx.x & 63;
  const int64_t inst = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (inst >= n_inst) return;
  const float* xi = x + inst * hw;
  float s = 0.f;
  for (int e = lane; e < hw; e += kWave) s += xi[e];
  const float mean = wave_sum(s) / (float)hw;
  float q = 0.f;
  for (int e = lane; e < hw; e += kWave) { const float d = xi[e] - mean; q += d * d; }
  const float r = 1.0f / sqrtf(wave_sum(q) / (float)hw + eps);
  for (int e = lane; e < hw; e += kWave) {
    const float h = (xi[e] - mean) * r;
    xhat[inst * hw + e] = h;
    y[inst * hw + e] = h > 0.f ? h : h * slope;
  }
  if (lane == 0) rstd[inst] = r;
}

struct InBwdP { const float* xhat; const float* rstd; const float* gy; int64_t n_inst; int hw; float slope; const float* addend; float* gx; };

// (pa, pb, na: two independent problems in one launch, workgroups [0, na) on the first -- see csrc/patch_conv.hip conv4s2_fwd_in_kernel)
__global__ __launch_bounds__(256) void inorm_lrelu_bwd_kernel(InBwdP pa, InBwdP pb, int na) {
  const bool second = na >= 0 && (int)blockIdx.x >= na;
  const InBwdP& p = second ? pb : pa;
  const float* __restrict__ xhat = p.xhat; const float* __restrict__ rstd = p.rstd; const float* __restrict__ gy = p.gy;
  const float* __restrict__ addend = p.addend; float* __restrict__ gx = p.gx;
  const int64_t n_inst = p.n_inst;
  const int hw = p.hw;
  const float slope = p.slope;
  const int lane = threadIdx.x & 63;
  const int64_t inst = (int64_t)((int)blockIdx.x - (second ? na : 0)) * 4 + (threadIdx.x >> 6);
  if (inst >= n_inst) return;
  const float* h = xhat + inst * hw;
  const float* g = gy + inst * hw;
  float sa = 0.f, sah = 0.f;
  for (int e = lane; e < hw; e += kWave) {
    const float a = g[e] * (h[e] > 0.f ? 1.0f : slope);
    sa += a;
    sah += a * h[e];
  }
  const float ma = wave_sum(sa) / (float)hw, mah = wave_sum(sah) / (float)hw, r = rstd[inst];
  for (int e = lane; e < hw; e += kWave) {
    const float a = g[e] * (h[e] > 0.f ? 1.0f : slope);
    const float v = r * (a - ma - h[e] * mah);
    gx[inst * hw + e] = addend ? v + addend[inst * hw + e] : v;     // (a second cotangent of x: the R1 path's, K16)
  }
}

__global__ __launch_bounds__(256) void inorm_lrelu_bwd_bwd_kernel(const float* __restrict__ xhat, const float* __restrict__ rstd,
                                                                   const float* __restrict__ gy, const float* __restrict__ ggx,
                                                                   int64_t n_inst, int hw, float slope, float* __restrict__ g_gy,
                                                                   float* __restrict__ g_x) {
  const int lane = threadIdx.x & 63;
  const int64_t inst = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (inst >= n_inst) return;
  const float* h = xhat + inst * hw;
  const float* g = gy + inst * hw;
  const float* u = ggx + inst * hw;
  float su = 0.f, sa = 0.f, sua = 0.f, suh = 0.f, sah = 0.f;
  for (int e = lane; e < hw; e += kWave) {
    const float a = g[e] * (h[e] > 0.f ? 1.0f : slope);
    su += u[e]; sa += a; sua += u[e] * a; suh += u[e] * h[e]; sah += a * h[e];
  }
  const float n = (float)hw;
  const float mu = wave_sum(su) / n, ma = wave_sum(sa) / n, C = wave_sum(suh) / n, D = wave_sum(sah) / n;
  const float A = wave_sum(sua) - n * mu * ma;
  const float r = rstd[inst], r2 = r * r;
  for (int e = lane; e < hw; e += kWave) {
    const float sl = h[e] > 0.f ? 1.0f : slope, a = g[e] * sl;
    g_gy[inst * hw + e] = sl * r * (u[e] - mu - h[e] * C);
    g_x[inst * hw + e] = -(r2 / n) * h[e] * (A - n * C * D) - r2 * (D * (u[e] - mu) + C * (a - ma) - 2.0f * C * D * h[e]);
  }
}
}  // namespace

extern "C" int tp_inorm_lrelu_fwd(const float* x, int64_t n_inst, int hw, float eps, float slope, float* y, float* xhat,
                                  float* rstd, tp_stream_t stream) {
  TP_REQUIRE(x && y && xhat && rstd, "null pointer");
  TP_REQUIRE(n_inst > 0 && hw > 0 && hw <= kWave * kMaxPerLane, "bad sizes");
  hipLaunchKernelGGL(inorm_lrelu_fwd_kernel, dim3((unsigned)((n_inst + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, n_inst, hw,
                     eps, slope, y, xhat, rstd);
  return tp::check_launch("tp_inorm_lrelu_fwd");
}

extern "C" int tp_inorm_lrelu_bwd(const float* xhat, const float* rstd, const float* gy, int64_t n_inst, int hw, float slope,
                                  const float* addend, float* gx, tp_stream_t stream) {
  TP_REQUIRE(xhat && rstd && gy && gx, "null pointer");
  TP_REQUIRE(n_inst > 0 && hw > 0, "bad sizes");
  const InBwdP p{xhat, rstd, gy, n_inst, hw, slope, addend, gx};
  hipLaunchKernelGGL(inorm_lrelu_bwd_kernel, dim3((unsigned)((n_inst + 3) / 4)), dim3(256), 0, (hipStream_t)stream, p, p, -1);
  return tp::check_launch("tp_inorm_lrelu_bwd");
}

extern "C" int tp_inorm_lrelu_bwd_pair(const tp_inorm_bwd_args* a, const tp_inorm_bwd_args* b, tp_stream_t stream) {
  TP_REQUIRE(a && b && a->xhat && a->rstd && a->gy && a->gx && b->xhat && b->rstd && b->gy && b->gx, "null pointer");
  TP_REQUIRE(a->n_inst > 0 && a->hw > 0 && b->n_inst > 0 && b->hw > 0 && a->gx != b->gx, "bad sizes");
  const InBwdP pa{a->xhat, a->rstd, a->gy, a->n_inst, a->hw, a->slope, a->addend, a->gx};
  const InBwdP pb{b->xhat, b->rstd, b->gy, b->n_inst, b->hw, b->slope, b->addend, b->gx};
  const unsigned ga = (unsigned)((a->n_inst + 3) / 4), gb = (unsigned)((b->n_inst + 3) / 4);
  hipLaunchKernelGGL(inorm_lrelu_bwd_kernel, dim3(ga + gb), dim3(256), 0, (hipStream_t)stream, pa, pb, (int)ga);
  return tp::check_launch("tp_inorm_lrelu_bwd_pair");
}

extern "C" int tp_inorm_lrelu_bwd_bwd(const float* xhat, const float* rstd, const float* gy, const float* ggx, int64_t n_inst,
                                      int hw, float slope, float* g_gy, float* g_x, tp_stream_t stream) {
  TP_REQUIRE(xhat && rstd && gy && ggx && g_gy && g_x, "null pointer");
  TP_REQUIRE(n_inst > 0 && hw > 0, "bad sizes");
  hipLaunchKernelGGL(inorm_lrelu_bwd_bwd_kernel, dim3((unsigned)((n_inst + 3) / 4)), dim3(256), 0, (hipStream_t)stream, xhat, rstd,
                     gy, ggx, n_inst, hw, slope, g_gy, g_x);
  return tp::check_launch("tp_inorm_lrelu_bwd_bwd");
}
