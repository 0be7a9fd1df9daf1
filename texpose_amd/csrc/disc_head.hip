// K14: the scale-conditioned head of the PatchGAN, one launch per derivative order.
// reference layers/discriminator.py:30-40,112-115: the ladder's [B,ndf,1,1] output z is concatenated with the positional
// encoding of the patch scale, enc = [sin(s 2^l pi), cos(s 2^l pi)]_{l<L}, and the scale itself, then
//     a = [z, enc, s];  t0 = lrelu(a);  t1 = lrelu(W1 t0);  t2 = lrelu(W2 t1);  out = W3 t2        (three 1x1 SN-convs, no bias)
// PyTorch runs this as 11 launches forward, ~9 backward and ~20 in the R1 double backward (model/nerf_adapt_st_gan.py:
// 794-807) -- per discriminator call, three calls per iteration: ~95 of the 297 launches of a captured B=4 iteration for a
// few hundred kFLOP.  LeakyReLU is piecewise linear, so with d(t) = (t > 0 ? 1 : slope):
//   backward       e2 = d(t2) W3^T g;  e1 = d(t1) W2^T e2;  e0 = d(t0) W1^T e1;   gz = e0[:C];
//                  gW3 = sum_b g t2;   gW2 = sum_b e2 (x) t1;   gW1 = sum_b e1 (x) t0
//   double backward (cotangent c of gz; the masks are constants almost everywhere, as in torch's LeakyReluBackwardBackward)
//                  a0 = d(t0) [c, 0];  a1 = d(t1) W1 a0;  a2 = d(t2) W2 a1;   d/d g = W3 a2;
//                  d/d W3 = sum_b g a2;   d/d W2 = sum_b e2 (x) a1;   d/d W1 = sum_b e1 (x) a0
// One workgroup: the three weights (37 KB at ndf = 64) sit in LDS, samples are processed four at a time, weight gradients
// are summed over the samples in a fixed order.
#include "tp_common.h"

namespace {
constexpr int kT = 256;
constexpr int kNB = 4;                         // samples per pass

struct HeadP {
  const float* z;        // [B,C]
  const float* scale;    // [B]
  const float* W1;       // [H,Cin]   Cin = C + 2L + 1
  const float* W2;       // [H,H]
  const float* W3;       // [H]
  const float* g;        // [B]        cotangent of out (backward, double backward)
  const float* c;        // [B,C]      cotangent of gz (double backward)
  float* t0; float* t1; float* t2;     // [B,Cin], [B,H], [B,H]  saved activations (forward: out; others: in)
  float* e1; float* e2;                // [B,H]    backward: out; double backward: in
  float* out;            // forward: [B];  backward: gz [B,C];  double backward: d/d g [B]
  float* gW1; float* gW2; float* gW3;  // weight gradients (backward / double backward)
  int B, C, L, H, Cin;
  float slope;
  int accumulate;        // backward: weight gradients are ADDED to gW1..3 (the double backward's were written there first)
};

__device__ __forceinline__ float lrelu(float v, float s) { return v > 0.f ? v : v * s; }
__device__ __forceinline__ float dl(float t, float s) { return t > 0.f ? 1.f : s; }

extern __shared__ float smem[];

__device__ __forceinline__ void load_weights(const HeadP& p, float* w1, float* w2, float* w3) {
  for (int i = threadIdx.x; i < p.H * p.Cin; i += kT) w1[i] = p.W1[i];
  for (int i = threadIdx.x; i < p.H * p.H; i += kT) w2[i] = p.W2[i];
  for (int i = threadIdx.x; i < p.H; i += kT) w3[i] = p.W3[i];
}

__global__ __launch_bounds__(kT) void head_fwd_kernel(HeadP p) {
  float* w1 = smem; float* w2 = w1 + p.H * p.Cin; float* w3 = w2 + p.H * p.H;
  float* a0 = w3 + p.H; float* a1 = a0 + kNB * p.Cin; float* a2 = a1 + kNB * p.H;
  load_weights(p, w1, w2, w3);
  for (int b0 = 0; b0 < p.B; b0 += kNB) {
    const int nb = min(kNB, p.B - b0);
    __syncthreads();
    for (int i = threadIdx.x; i < nb * p.Cin; i += kT) {
      const int b = i / p.Cin, j = i - b * p.Cin;
      const float s = p.scale[b0 + b];
      float v;
      if (j < p.C) v = p.z[(size_t)(b0 + b) * p.C + j];
      else if (j < p.C + 2 * p.L) {
        const int l = (j - p.C) % p.L;
        const float arg = tp::mul_rn(s, tp::mul_rn((float)(1 << l), 3.14159265358979323846f));     // s * (2^l pi rounded to fp32)
        v = tp::sincos_sel(arg, j - p.C >= p.L ? 1 : 0);
      } else v = s;
      v = lrelu(v, p.slope);
      a0[i] = v;
      p.t0[(size_t)(b0 + b) * p.Cin + j] = v;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nb * p.H; i += kT) {
      const int b = i / p.H, o = i - b * p.H;
      float acc = 0.f;
      for (int j = 0; j < p.Cin; ++j) acc += w1[o * p.Cin + j] * a0[b * p.Cin + j];
      acc = lrelu(acc, p.slope);
      a1[i] = acc;
      p.t1[(size_t)(b0 + b) * p.H + o] = acc;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nb * p.H; i += kT) {
      const int b = i / p.H, o = i - b * p.H;
      float acc = 0.f;
      for (int j = 0; j < p.H; ++j) acc += w2[o * p.H + j] * a1[b * p.H + j];
      acc = lrelu(acc, p.slope);
      a2[i] = acc;
      p.t2[(size_t)(b0 + b) * p.H + o] = acc;
    }
    __syncthreads();
    if ((int)threadIdx.x < nb) {
      float acc = 0.f;
      for (int j = 0; j < p.H; ++j) acc += w3[j] * a2[threadIdx.x * p.H + j];
      p.out[b0 + threadIdx.x] = acc;
    }
  }
}

// weight gradients: out[i][j] = sum_b u[b][i] v[b][j] (rows H, cols n_col), fixed order over b
__device__ __forceinline__ void outer_sum(float* out, const float* u, int ldu, const float* v, int ldv, int rows, int cols, int B,
                                          bool accumulate) {
  for (int e = threadIdx.x; e < rows * cols; e += kT) {
    const int i = e / cols, j = e - i * cols;
    float acc = 0.f;
    for (int b = 0; b < B; ++b) acc += u[(size_t)b * ldu + i] * v[(size_t)b * ldv + j];
    out[e] = accumulate ? out[e] + acc : acc;
  }
}

__global__ __launch_bounds__(kT) void head_bwd_kernel(HeadP p) {
  float* w1 = smem; float* w2 = w1 + p.H * p.Cin; float* w3 = w2 + p.H * p.H;
  float* s2 = w3 + p.H; float* s1 = s2 + kNB * p.H;
  load_weights(p, w1, w2, w3);
  for (int b0 = 0; b0 < p.B; b0 += kNB) {
    const int nb = min(kNB, p.B - b0);
    __syncthreads();
    for (int i = threadIdx.x; i < nb * p.H; i += kT) {
      const int b = i / p.H, o = i - b * p.H;
      const float v = dl(p.t2[(size_t)(b0 + b) * p.H + o], p.slope) * (w3[o] * p.g[b0 + b]);
      s2[i] = v;
      p.e2[(size_t)(b0 + b) * p.H + o] = v;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nb * p.H; i += kT) {
      const int b = i / p.H, j = i - b * p.H;
      float acc = 0.f;
      for (int o = 0; o < p.H; ++o) acc += w2[o * p.H + j] * s2[b * p.H + o];
      acc *= dl(p.t1[(size_t)(b0 + b) * p.H + j], p.slope);
      s1[i] = acc;
      p.e1[(size_t)(b0 + b) * p.H + j] = acc;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nb * p.C; i += kT) {          // only the z part of e0 is anybody's gradient
      const int b = i / p.C, j = i - b * p.C;
      float acc = 0.f;
      for (int o = 0; o < p.H; ++o) acc += w1[o * p.Cin + j] * s1[b * p.H + o];
      p.out[(size_t)(b0 + b) * p.C + j] = acc * dl(p.t0[(size_t)(b0 + b) * p.Cin + j], p.slope);
    }
  }
  if (p.gW1 == nullptr) return;                                   // data gradient only (frozen weights; the R1 penalty's first pass)
  __threadfence_block();
  __syncthreads();
  const bool add = p.accumulate != 0;
  for (int j = threadIdx.x; j < p.H; j += kT) {
    float acc = 0.f;
    for (int b = 0; b < p.B; ++b) acc += p.g[b] * p.t2[(size_t)b * p.H + j];
    p.gW3[j] = add ? p.gW3[j] + acc : acc;
  }
  outer_sum(p.gW2, p.e2, p.H, p.t1, p.H, p.H, p.H, p.B, add);
  outer_sum(p.gW1, p.e1, p.H, p.t0, p.Cin, p.H, p.Cin, p.B, add);
}

__global__ __launch_bounds__(kT) void head_bwd_bwd_kernel(HeadP p) {
  float* w1 = smem; float* w2 = w1 + p.H * p.Cin; float* w3 = w2 + p.H * p.H;
  float* a0 = w3 + p.H; float* a1 = a0 + kNB * p.Cin; float* a2 = a1 + kNB * p.H;
  load_weights(p, w1, w2, w3);
  for (int b0 = 0; b0 < p.B; b0 += kNB) {
    const int nb = min(kNB, p.B - b0);
    __syncthreads();
    for (int i = threadIdx.x; i < nb * p.Cin; i += kT) {
      const int b = i / p.Cin, j = i - b * p.Cin;
      const float v = j < p.C ? dl(p.t0[(size_t)(b0 + b) * p.Cin + j], p.slope) * p.c[(size_t)(b0 + b) * p.C + j] : 0.f;
      a0[i] = v;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nb * p.H; i += kT) {
      const int b = i / p.H, o = i - b * p.H;
      float acc = 0.f;
      for (int j = 0; j < p.C; ++j) acc += w1[o * p.Cin + j] * a0[b * p.Cin + j];
      a1[i] = acc * dl(p.t1[(size_t)(b0 + b) * p.H + o], p.slope);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nb * p.H; i += kT) {
      const int b = i / p.H, o = i - b * p.H;
      float acc = 0.f;
      for (int j = 0; j < p.H; ++j) acc += w2[o * p.H + j] * a1[b * p.H + j];
      a2[i] = acc * dl(p.t2[(size_t)(b0 + b) * p.H + o], p.slope);
    }
    __syncthreads();
    if ((int)threadIdx.x < nb) {
      float acc = 0.f;
      for (int j = 0; j < p.H; ++j) acc += w3[j] * a2[threadIdx.x * p.H + j];
      p.out[b0 + threadIdx.x] = acc;
    }
    // this pass's contribution to the weight gradients (b ascending over the passes: fixed order)
    for (int j = threadIdx.x; j < p.H; j += kT) {
      float acc = b0 == 0 ? 0.f : p.gW3[j];
      for (int b = 0; b < nb; ++b) acc += p.g[b0 + b] * a2[b * p.H + j];
      p.gW3[j] = acc;
    }
    for (int e = threadIdx.x; e < p.H * p.H; e += kT) {
      const int i = e / p.H, j = e - i * p.H;
      float acc = b0 == 0 ? 0.f : p.gW2[e];
      for (int b = 0; b < nb; ++b) acc += p.e2[(size_t)(b0 + b) * p.H + i] * a1[b * p.H + j];
      p.gW2[e] = acc;
    }
    for (int e = threadIdx.x; e < p.H * p.Cin; e += kT) {
      const int i = e / p.Cin, j = e - i * p.Cin;
      float acc = b0 == 0 ? 0.f : p.gW1[e];
      for (int b = 0; b < nb; ++b) acc += p.e1[(size_t)(b0 + b) * p.H + i] * a0[b * p.Cin + j];
      p.gW1[e] = acc;
    }
  }
}

size_t lds_bytes(const HeadP& p) { return sizeof(float) * ((size_t)p.H * p.Cin + (size_t)p.H * p.H + p.H + (size_t)kNB * (p.Cin + 2 * p.H)); }

int fill(HeadP* q, const tp_disc_head_args* a, const char* what) {
  if (!a || a->B <= 0 || a->C <= 0 || a->L < 0 || a->L > 24 || a->H <= 0) { tp::set_error("%s: bad sizes", what); return -1; }
  q->B = a->B; q->C = a->C; q->L = a->L; q->H = a->H; q->Cin = a->C + 2 * a->L + 1; q->slope = a->slope;
  q->z = a->z; q->scale = a->scale; q->W1 = a->W1; q->W2 = a->W2; q->W3 = a->W3; q->g = a->g_out; q->c = a->c_gz;
  q->t0 = a->t0; q->t1 = a->t1; q->t2 = a->t2; q->e1 = a->e1; q->e2 = a->e2; q->out = a->out;
  q->gW1 = a->gW1; q->gW2 = a->gW2; q->gW3 = a->gW3; q->accumulate = a->accumulate_gw;
  if (!q->W1 || !q->W2 || !q->W3 || !q->t0 || !q->t1 || !q->t2 || !q->out) { tp::set_error("%s: null pointer", what); return -1; }
  if (lds_bytes(*q) > 150 * 1024) { tp::set_error("%s: head too wide for one workgroup's LDS", what); return -1; }
  return 0;
}

template <class K>
int launch(K kernel, const HeadP& q, tp_stream_t stream, const char* what, unsigned long long& flags) {
  const size_t lds = lds_bytes(q);
  if (lds > 48 * 1024 && tp::first_use_on_device(flags) &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess) {
    tp::set_error("%s: cannot raise the LDS limit", what);
    return -1;
  }
  hipLaunchKernelGGL(kernel, dim3(1), dim3(kT), lds, (hipStream_t)stream, q);
  return tp::check_launch(what);
}
}  // namespace

extern "C" {
int tp_disc_head_fwd(const tp_disc_head_args* a, tp_stream_t stream) {
  static unsigned long long flags = 0;
  HeadP q{};
  if (int rc = fill(&q, a, "tp_disc_head_fwd")) return rc;
  TP_REQUIRE(q.z && q.scale, "z / scale missing");
  return launch(head_fwd_kernel, q, stream, "tp_disc_head_fwd", flags);
}
int tp_disc_head_bwd(const tp_disc_head_args* a, tp_stream_t stream) {
  static unsigned long long flags = 0;
  HeadP q{};
  if (int rc = fill(&q, a, "tp_disc_head_bwd")) return rc;
  TP_REQUIRE(q.g && q.e1 && q.e2, "operand missing");
  TP_REQUIRE((q.gW1 && q.gW2 && q.gW3) || (!q.gW1 && !q.gW2 && !q.gW3), "gW1..3: all or none");
  return launch(head_bwd_kernel, q, stream, "tp_disc_head_bwd", flags);
}
int tp_disc_head_bwd_bwd(const tp_disc_head_args* a, tp_stream_t stream) {
  static unsigned long long flags = 0;
  HeadP q{};
  if (int rc = fill(&q, a, "tp_disc_head_bwd_bwd")) return rc;
  TP_REQUIRE(q.g && q.c && q.e1 && q.e2 && q.gW1 && q.gW2 && q.gW3, "operand missing");
  return launch(head_bwd_bwd_kernel, q, stream, "tp_disc_head_bwd_bwd", flags);
}
}
