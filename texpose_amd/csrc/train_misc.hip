// K13: the small per-iteration pieces of a training step that PyTorch spreads over dozens of 1-5 us launches each.  In a
// hipGraph replay of the B=4 GAN iteration every launch costs ~4 us whatever it computes, and 385 of 579 launches were such
// elementwise / fill / copy kernels on a few hundred values (profiles/r2): these kernels replace the longest chains, one
// launch per direction, reproducing torch's fp32 operation order where a golden pins the values.
//   tp_patch_coords      FlexPatchSampler.__call__ (reference tools/patch_sampler.py:80-114): 17 launches
//   tp_bce_logits_*      binary_cross_entropy_with_logits(d, const target), mean (model/nerf_adapt_st_gan.py:809-823,
//                        compute_gan_loss 'standard'): 8 launches forward, 4 backward, three times per iteration
//   tp_feat_inputs_*     the two (fake, real) pairs of the feature loss, masked, concatenated and ImageNet-normalised
//                        (model/nerf_adapt_st_gan.py:758-766 + layers/perceptual_loss.py:19-20,31-37): 14 + 4 launches
//   tp_step_flags        range flag + loss finiteness -> sticky gate words (reference model/base.py:153-154 asserts per term)
//   tp_adam_step         optim_nerf.step() (model/nerf_adapt_st_gan.py:62-68,125: torch.optim.Adam) for every tensor at once, gated
//   tp_disc_inputs       the real / fake patch stacks of the discriminator step (model/nerf_adapt_st_gan.py:478-497): 10 launches
#include "tp_common.h"
#include "step_prologue.h"
#include <stdlib.h>

namespace {
constexpr int kBlock = 256;

// sum over the B threads of the workgroup (the same value in every thread): a butterfly inside each wavefront, then the wavefront sums in
// wavefront order -- 2 barriers instead of the 9-11 of an LDS tree (these one-workgroup kernels sit on the chain that bounds the B=4
// training iteration).  `red`: >= B / 64 floats.
template <int B>
__device__ __forceinline__ float block_total(float v, float* red) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = red[0];
#pragma unroll
  for (int i = 1; i < B / 64; ++i) r += red[i];
  __syncthreads();
  return r;
}

// ---- patch coordinates: s = u0 * (hi - lo) + lo;  x = lattice_j * s + (u1 * 2 - 1) * (1 - s);  y likewise with u2
// u == NULL: the three uniforms of image b are drawn here, Philox4x32-10 with key = seed and counter (b, c_lo, 'patc', c_hi),
// c = *counter (the step counter of a captured training step) -- words x, y, z -> scale, x shift, y shift.
// (the arithmetic: step_prologue.h, shared with the ray-generation launch of a captured training step)
__global__ __launch_bounds__(kBlock) void patch_coords_kernel(tp_prologue::Sampler q) {
  const int e = blockIdx.x * kBlock + threadIdx.x;
  if (e >= q.B * q.p * q.p) return;
  float x, y;
  tp_prologue::patch_coord(q, e, x, y);
}

// ---- mean_i [ (1 - t) x_i - log_sigmoid(x_i) ]  (torch's formula), one workgroup, fixed-order tree
__global__ __launch_bounds__(kBlock) void bce_logits_fwd_kernel(const float* __restrict__ x, int n, float target, float* __restrict__ out) {
  __shared__ float red[kBlock];
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += kBlock) {
    const float v = x[i];
    const float ls = fminf(v, 0.f) - log1pf(expf(-fabsf(v)));      // log_sigmoid
    acc += (1.f - target) * v - ls;
  }
  const float tot = block_total<kBlock>(acc, red);
  if (threadIdx.x == 0) out[0] = tot / (float)n;
}
__global__ __launch_bounds__(kBlock) void bce_logits_bwd_kernel(const float* __restrict__ x, int n, float target, const float* __restrict__ g,
                                                                 float* __restrict__ gx) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const float sg = 1.f / (1.f + expf(-x[i]));
  gx[i] = (sg - target) * g[0] / (float)n;
}

// ---- feature-loss inputs.  rgb [B,P,3] (render layout), gathered [B,14,P] (patch gather: image 0..2, synthetic image 3..5,
// object mask 12, synthetic mask 13).  out [4B,3,P]: rows 0..B-1 fake1 = rgb, B..2B-1 fake2 = rgb m + image (1 - m),
// 2B..3B-1 real1 = image m + image_syn pad, 3B..4B-1 real2 = image;  pad = (mask_syn == 1 && m == 0);  then (v - mean) / std.
struct FeatP { const float* rgb; const float* gathered; float* out; const float* g_out; float* g_rgb; int B, P, c_img, c_syn, c_mask, c_msyn, n_ch;
               float mean[3], stdv[3]; };
__global__ __launch_bounds__(kBlock) void feat_inputs_fwd_kernel(FeatP a) {
  const int e = blockIdx.x * kBlock + threadIdx.x;
  if (e >= a.B * a.P) return;
  const int b = e / a.P, p = e - b * a.P;
  const float* gp = a.gathered + (size_t)b * a.n_ch * a.P + p;
  const float m = gp[(size_t)a.c_mask * a.P], ms = gp[(size_t)a.c_msyn * a.P];
  const float pad = (ms == 1.f && m == 0.f) ? 1.f : 0.f;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float r = a.rgb[(size_t)e * 3 + c], im = gp[(size_t)(a.c_img + c) * a.P], sy = gp[(size_t)(a.c_syn + c) * a.P];
    const float v[4] = {r, tp::add_rn(tp::mul_rn(r, m), tp::mul_rn(im, tp::sub_rn(1.f, m))),
                        tp::add_rn(tp::mul_rn(im, m), tp::mul_rn(sy, pad)), im};
#pragma unroll
    for (int k = 0; k < 4; ++k)
      a.out[((size_t)(k * a.B + b) * 3 + c) * a.P + p] = tp::div_rn(tp::sub_rn(v[k], a.mean[c]), a.stdv[c]);
  }
}
// d/d rgb = (g[fake1] + g[fake2] * m) / std
__global__ __launch_bounds__(kBlock) void feat_inputs_bwd_kernel(FeatP a) {
  const int e = blockIdx.x * kBlock + threadIdx.x;
  if (e >= a.B * a.P) return;
  const int b = e / a.P, p = e - b * a.P;
  const float m = a.gathered[((size_t)b * a.n_ch + a.c_mask) * a.P + p];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float g1 = tp::div_rn(a.g_out[((size_t)b * 3 + c) * a.P + p], a.stdv[c]);
    const float g2 = tp::div_rn(a.g_out[((size_t)(a.B + b) * 3 + c) * a.P + p], a.stdv[c]);
    a.g_rgb[(size_t)e * 3 + c] = tp::add_rn(g1, tp::mul_rn(g2, m));
  }
}

// ---- discriminator inputs (no gradient): real = image m + rgb pad, fake = rgb, both followed by the six geometry channels
__global__ __launch_bounds__(kBlock) void disc_inputs_kernel(const float* __restrict__ rgb, const float* __restrict__ gathered, int B, int P,
                                                             int geo, float* __restrict__ real, float* __restrict__ fake) {
  const int e = blockIdx.x * kBlock + threadIdx.x;
  if (e >= B * P) return;
  const int b = e / P, p = e - b * P, nc = geo ? 9 : 3;
  const float* gp = gathered + (size_t)b * 14 * P + p;
  const float m = gp[12 * (size_t)P], ms = gp[13 * (size_t)P];
  const float pad = (ms == 1.f && m == 0.f) ? 1.f : 0.f;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float r = rgb[(size_t)e * 3 + c];
    real[((size_t)b * nc + c) * P + p] = tp::add_rn(tp::mul_rn(gp[(size_t)c * P], m), tp::mul_rn(r, pad));
    fake[((size_t)b * nc + c) * P + p] = r;
  }
  if (geo)
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      const float v = gp[(size_t)(6 + c) * P];
      real[((size_t)b * nc + 3 + c) * P + p] = v;
      fake[((size_t)b * nc + 3 + c) * P + p] = v;
    }
}

// ---- the per-iteration host -> device state of a replayed training step in ONE launch: the batch into the static input tensors
// (up to TP_STEP_INPUTS_MAX_COPIES contiguous copies, 16-byte lanes; torch: one multi-tensor copy per dtype), the host-computed
// scalars (annealed patch-scale bound, discriminator progress; torch: one fill each) and the sticky gate words out to pinned
// host memory (torch: a copyBuffer).  Five launches between two replays become one.
struct StepInputs {
  char* dst[TP_STEP_INPUTS_MAX_COPIES];
  const char* src[TP_STEP_INPUTS_MAX_COPIES];
  int64_t end16[TP_STEP_INPUTS_MAX_COPIES];       // running end of the copies in 16-byte units (every size is padded up to 16)
  int64_t bytes[TP_STEP_INPUTS_MAX_COPIES];
  int n;
  float* sdst[TP_STEP_INPUTS_MAX_SCALARS];
  float sval[TP_STEP_INPUTS_MAX_SCALARS];
  int n_scalars;
  const int* words_src;
  int* words_dst;                                 // pinned host memory (device-visible), or NULL
  int n_words;
};
__global__ __launch_bounds__(kBlock) void step_inputs_kernel(StepInputs t, int64_t total16) {
  if (blockIdx.x == 0) {
    if ((int)threadIdx.x < t.n_scalars) t.sdst[threadIdx.x][0] = t.sval[threadIdx.x];
    if (t.words_dst != nullptr && (int)threadIdx.x >= 64 && (int)threadIdx.x < 64 + t.n_words)
      __hip_atomic_store(t.words_dst + (threadIdx.x - 64), t.words_src[threadIdx.x - 64], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  int k = 0;
  for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < total16; e += (int64_t)gridDim.x * kBlock) {
    while (e >= t.end16[k]) ++k;
    const int64_t off = (e - (k == 0 ? 0 : t.end16[k - 1])) * 16;
    if (off + 16 <= t.bytes[k]) {
      *reinterpret_cast<uint4*>(t.dst[k] + off) = *reinterpret_cast<const uint4*>(t.src[k] + off);
    } else {
      for (int64_t b = off; b < t.bytes[k]; ++b) t.dst[k][b] = t.src[k][b];
    }
  }
}

// ---- step gate: fold this step's range flag / loss finiteness into the sticky words, then snapshot them for the optimiser
__global__ void step_flags_kernel(const int* status, const float* total, int* bad, int n_bad, int word_status, int word_finite, int* snapshot) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (status != nullptr && (status[0] & 1)) bad[word_status] |= 1;
  const float v = total[0];
  if (!(v - v == 0.f)) bad[word_finite] |= 1;                  // NaN or +-Inf
  for (int k = 0; k < n_bad; ++k) snapshot[k] = bad[k];
}

// ---- Adam (torch.optim.Adam: no weight decay, no amsgrad), every tensor of the step in one launch.  Scalars in double like
// the reference's single-tensor implementation (bias_correction = 1 - beta^step, step_size = lr / bias_correction1); the
// weights 1 - beta are formed in double and THEN rounded (1.0f - 0.999f is off by 1.3e-5 relative).
struct AdamTable {
  float* p[TP_ADAM_MAX_TENSORS];
  const float* g[TP_ADAM_MAX_TENSORS];
  float* m[TP_ADAM_MAX_TENSORS];
  float* v[TP_ADAM_MAX_TENSORS];
  float* step[TP_ADAM_MAX_TENSORS];          // steps taken BEFORE this one (the last block to finish adds 1)
  int64_t end[TP_ADAM_MAX_TENSORS];
  int n;
};
// The step counters are read by every block and must advance only after the last of them has: the block that finishes last (a
// ticket in CALLER-OWNED device memory, one zero-filled word per stream that may run this entry point, reset by that block; agent-scope
// accesses as in nerf_losses.hip) adds 1 to each distinct counter.
__global__ __launch_bounds__(kBlock) void adam_kernel(AdamTable t, const float* lr_dev, double lr_host, double log_beta1, double log_beta2, float b2,
                                                      float eps, float w1, float w2, int64_t total, const int* gate, int n_gate, unsigned int* ticket) {
  __shared__ bool last;
  {
    // the gate words in ONE round trip (lane k of every wavefront loads word k; a scalar loop with an early exit was a dependent
    // load per word in front of everything else)
    const int lane = threadIdx.x & 63;
    const int word = lane < n_gate ? gate[lane] : 0;
    if (__builtin_amdgcn_ballot_w64(word != 0) != 0ull) return;
  }
  const double lr = lr_dev != nullptr ? (double)*lr_dev : lr_host;
  // the bias corrections are per tensor: thread k of every workgroup forms tensor k's once (beta^step as exp(step ln beta) in double, ln
  // beta from the host: a generic double pow() is ~6x the instructions of exp()), everybody reads them from LDS
  __shared__ float s_step_size[TP_ADAM_MAX_TENSORS], s_bc2_sqrt[TP_ADAM_MAX_TENSORS];
  // the tensor table goes to LDS once per workgroup: indexed per LANE out of the kernel-argument segment it was a dependent global
  // load per step of the search below and per pointer (up to 18 + 4 round trips in front of every element's own loads)
  __shared__ int64_t s_end[TP_ADAM_MAX_TENSORS];
  __shared__ float* s_p[TP_ADAM_MAX_TENSORS];
  __shared__ const float* s_g[TP_ADAM_MAX_TENSORS];
  __shared__ float* s_m[TP_ADAM_MAX_TENSORS];
  __shared__ float* s_v[TP_ADAM_MAX_TENSORS];
  if ((int)threadIdx.x < TP_ADAM_MAX_TENSORS) {
    const int k = threadIdx.x;
    s_end[k] = t.end[k]; s_p[k] = t.p[k]; s_g[k] = t.g[k]; s_m[k] = t.m[k]; s_v[k] = t.v[k];      // (entries >= n: end = total, null pointers)
  }
  if ((int)threadIdx.x < t.n) {
    const double step = (double)t.step[threadIdx.x][0] + 1.0;
    const double bc1 = 1.0 - exp(step * log_beta1), bc2 = 1.0 - exp(step * log_beta2);
    s_step_size[threadIdx.x] = (float)(lr / bc1); s_bc2_sqrt[threadIdx.x] = (float)sqrt(bc2);
  }
  __syncthreads();
  // four elements per thread and round, their sixteen loads in flight together (one element per round paid a memory latency per element:
  // 18 us for 0.63 M parameters, at the end of the render's backward chain)
  constexpr int kU = 4;
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t e0 = (int64_t)blockIdx.x * kBlock + threadIdx.x; e0 < total; e0 += kU * stride) {
    int kk[kU]; int64_t ii[kU]; bool live[kU];
    float g[kU], m[kU], v[kU], pv[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int64_t e = e0 + u * stride;
      live[u] = e < total;
      int k = 0;
      if (live[u]) {                                      // first tensor whose running end lies beyond e: five LDS reads
        int hi = TP_ADAM_MAX_TENSORS - 1;
        while (k < hi) {
          const int mid = (k + hi) >> 1;
          if (e >= s_end[mid]) k = mid + 1; else hi = mid;
        }
      }
      kk[u] = k; ii[u] = live[u] ? e - (k == 0 ? 0 : s_end[k - 1]) : 0;
      if (live[u]) { g[u] = s_g[k][ii[u]]; m[u] = s_m[k][ii[u]]; v[u] = s_v[k][ii[u]]; pv[u] = s_p[k][ii[u]]; }
      else { g[u] = m[u] = v[u] = pv[u] = 0.0f; }
    }
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      if (!live[u]) continue;
      const int k = kk[u]; const int64_t i = ii[u];
      const float step_size = s_step_size[k], bc2_sqrt = s_bc2_sqrt[k];
      float mm = m[u], vv = v[u];
      mm = tp::add_rn(mm, tp::mul_rn(tp::sub_rn(g[u], mm), w1));                         // exp_avg.lerp_(grad, 1 - beta1): w1 = (float)(1 - beta1)
      vv = tp::add_rn(tp::mul_rn(vv, b2), tp::mul_rn(tp::mul_rn(w2, g[u]), g[u]));      // mul_(beta2).addcmul_(g, g, 1 - beta2): w2 = (float)(1 - beta2)
      s_m[k][i] = mm;
      s_v[k][i] = vv;
      const float denom = tp::add_rn(tp::div_rn(sqrtf(vv), bc2_sqrt), eps);
      s_p[k][i] = tp::add_rn(pv[u], tp::mul_rn(-step_size, tp::div_rn(mm, denom)));   // addcdiv_(exp_avg, denom, value = -step_size)
    }
  }
  __syncthreads();                                         // every thread of this block has read its counters
  if (threadIdx.x == 0) last = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
  __syncthreads();
  if (!last) return;
  if (threadIdx.x == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // (tensors of one optimiser usually share nothing, but two entries may name the same counter: add once per distinct pointer)
  for (int kk = threadIdx.x; kk < t.n; kk += blockDim.x) {
    bool first = true;
    for (int j = 0; j < kk; ++j) first = first && t.step[j] != t.step[kk];
    if (first) t.step[kk][0] += 1.0f;
  }
}

// ---- data parallel (no reference counterpart: options.py:112 asserts one GPU): the gradients of one optimiser step into the flat
// all-reduce buffer, pre-scaled by 1 / world (the SUM all-reduce then leaves the average), and the step-gate words into the buffer's
// tail as 0 / 1 floats -- STICKY: a tail word that is non-zero stays non-zero, so after the all-reduce the tail is the job-wide gate
// of this and every later step until the host clears it.  One launch (torch: a multi-tensor copy, a scale, a compare + copy).
struct GradPack {
  const float* src[TP_GRAD_PACK_MAX_TENSORS];     // NULL: this rank has no gradient for the tensor (zeros)
  int64_t end[TP_GRAD_PACK_MAX_TENSORS];
  int n;
};
__global__ __launch_bounds__(kBlock) void grad_pack_kernel(GradPack t, float* __restrict__ flat, int64_t total, float scale,
                                                           const int* __restrict__ words, int n_words, float* __restrict__ tail) {
  if (blockIdx.x == 0 && (int)threadIdx.x < n_words) {
    const bool set = (words != nullptr && words[threadIdx.x] != 0) || tail[threadIdx.x] != 0.f;
    tail[threadIdx.x] = set ? 1.f : 0.f;
  }
  constexpr int kU = 4;
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t e0 = (int64_t)blockIdx.x * kBlock + threadIdx.x; e0 < total; e0 += kU * stride) {
    float g[kU]; bool live[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int64_t e = e0 + u * stride;
      live[u] = e < total;
      int k = 0;
      while (live[u] && e >= t.end[k]) ++k;
      const float* s = live[u] ? t.src[k] : nullptr;
      g[u] = s != nullptr ? s[e - (k == 0 ? 0 : t.end[k - 1])] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < kU; ++u)
      if (live[u]) flat[e0 + u * stride] = tp::mul_rn(g[u], scale);
  }
}

// ---- feature loss of the generator step (model/nerf_adapt_st_gan.py:762-766; layers/perceptual_loss.py:39-45):
// feat [4n] = features of [fake1 | fake2 | real1 | real2]; l1 = mean((fake1 - real1)^2), l2 = mean((fake2 - real2)^2),
// out = {l1 + w2 l2, l1, l2}.  One workgroup, fixed-order tree (torch: two mse_loss launches x 2 + mul + add).
constexpr int kRedBlock = 1024;
__global__ __launch_bounds__(kRedBlock) void feat_pair_loss_fwd_kernel(const float* __restrict__ feat, int64_t n, float w2, float* __restrict__ out) {
  __shared__ float red[2][kRedBlock];
  float a1 = 0.f, a2 = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += kRedBlock) {
    const float d1 = feat[i] - feat[2 * n + i], d2 = feat[n + i] - feat[3 * n + i];
    a1 += d1 * d1;
    a2 += d2 * d2;
  }
  red[0][threadIdx.x] = a1; red[1][threadIdx.x] = a2;
  __syncthreads();
  for (int s = kRedBlock >> 1; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) { red[0][threadIdx.x] += red[0][threadIdx.x + s]; red[1][threadIdx.x] += red[1][threadIdx.x + s]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float l1 = red[0][0] / (float)n, l2 = red[1][0] / (float)n;
    out[0] = l1 + w2 * l2; out[1] = l1; out[2] = l2;
  }
}
// d/d feat: 2 (fake1 - real1) g / n | 2 w2 (fake2 - real2) g / n | 0 | 0   (the targets are detached)
__global__ __launch_bounds__(kBlock) void feat_pair_loss_bwd_kernel(const float* __restrict__ feat, int64_t n, float w2, const float* __restrict__ g,
                                                                     float* __restrict__ g_feat) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const float s = 2.f * g[0] / (float)n;
  g_feat[i] = s * (feat[i] - feat[2 * n + i]);
  g_feat[n + i] = s * w2 * (feat[n + i] - feat[3 * n + i]);
  g_feat[2 * n + i] = 0.f;
  g_feat[3 * n + i] = 0.f;
}

// ---- cotangent of the fake patch stack wrt the rendered colours: g_rgb [B,P,3] = g_fake [B,nc,P][:, 0:3] transposed
__global__ __launch_bounds__(kBlock) void fake_patch_bwd_kernel(const float* __restrict__ g_fake, int B, int P, int nc, float* __restrict__ g_rgb) {
  const int e = blockIdx.x * kBlock + threadIdx.x;
  if (e >= B * P) return;
  const int b = e / P, p = e - b * P;
#pragma unroll
  for (int c = 0; c < 3; ++c) g_rgb[(size_t)e * 3 + c] = g_fake[((size_t)b * nc + c) * P + p];
}

// ---- R1 penalty value (model/nerf_adapt_st_gan.py:794-807 + the .mean() of its caller): out = sum(g^2) / B for g [B, m];
// backward: 2 g cot / B.  One workgroup forward (fixed order).
__global__ __launch_bounds__(kRedBlock) void sumsq_mean_fwd_kernel(const float* __restrict__ g, int64_t n, int B, float* __restrict__ out) {
  __shared__ float red[kRedBlock];
  float a = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += kRedBlock) a += g[i] * g[i];
  const float tot = block_total<kRedBlock>(a, red);
  if (threadIdx.x == 0) out[0] = tot / (float)B;
}
__global__ __launch_bounds__(kBlock) void sumsq_mean_bwd_kernel(const float* __restrict__ g, int64_t n, int B, const float* __restrict__ cot,
                                                                 float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < n) out[i] = 2.f * cot[0] / (float)B * g[i];
}

// ---- MaxPool2d(2, 2) of the feature network (layers/perceptual_loss.py:8-18, torchvision VGG19 "M" entries), x [n, H, W] (n = images
// x channels, H and W even) -> y [n, H/2, W/2] and the window position of the maximum (first one in row-major order on ties, as
// torch's kernel: strict > and NaN wins).  The backward writes all four inputs of a window: no zero fill before it.
__global__ __launch_bounds__(kBlock) void maxpool2_fwd_kernel(const float* __restrict__ x, int64_t n_out, int H, int W, float* __restrict__ y,
                                                               uint8_t* __restrict__ arg) {
  const int64_t o = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (o >= n_out) return;
  const int ow = W / 2, oh = H / 2;
  const int64_t img = o / ((int64_t)oh * ow);
  const int r = (int)(o - img * oh * ow), oy = r / ow, ox = r - oy * ow;
  const float* p = x + (img * H + 2 * oy) * W + 2 * ox;
  float best = p[0];
  int k = 0;
  const float v1 = p[1], v2 = p[W], v3 = p[W + 1];
  if (v1 > best || v1 != v1) { best = v1; k = 1; }
  if (v2 > best || v2 != v2) { best = v2; k = 2; }
  if (v3 > best || v3 != v3) { best = v3; k = 3; }
  y[o] = best;
  arg[o] = (uint8_t)k;
}
__global__ __launch_bounds__(kBlock) void maxpool2_bwd_kernel(const float* __restrict__ gy, const uint8_t* __restrict__ arg, int64_t n_out, int H,
                                                               int W, float* __restrict__ gx) {
  const int64_t o = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (o >= n_out) return;
  const int ow = W / 2, oh = H / 2;
  const int64_t img = o / ((int64_t)oh * ow);
  const int r = (int)(o - img * oh * ow), oy = r / ow, ox = r - oy * ow;
  float* p = gx + (img * H + 2 * oy) * W + 2 * ox;
  const float g = gy[o];
  const int k = arg[o];
  *reinterpret_cast<float2*>(p) = make_float2(k == 0 ? g : 0.f, k == 1 ? g : 0.f);
  *reinterpret_cast<float2*>(p + W) = make_float2(k == 2 ? g : 0.f, k == 3 ? g : 0.f);
}

// ---- the same two in ONE launch for the explicit discriminator-step schedule (K16): every block writes its part of
// out_g = 2 w g / B, block 0 also reduces the value (same fixed-order tree as above)
__global__ __launch_bounds__(kRedBlock) void sumsq_mean_fwd_bwd_kernel(const float* __restrict__ g, int64_t n, int B, float w,
                                                                        float* __restrict__ out, float* __restrict__ out_g) {
  const float k = 2.f * w / (float)B;
  for (int64_t i = (int64_t)blockIdx.x * kRedBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kRedBlock) out_g[i] = k * g[i];
  if (blockIdx.x != 0) return;
  __shared__ float red[kRedBlock];
  float a = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += kRedBlock) a += g[i] * g[i];
  const float tot = block_total<kRedBlock>(a, red);
  if (threadIdx.x == 0) { out[0] = tot / (float)B; out[1] = w * out[0]; }      // (the reference logs the WEIGHTED penalty, :151-153)
}

// ---- both GAN-loss terms of the discriminator step (model/nerf_adapt_st_gan.py:139-160) and their weighted cotangents in one
// launch: out2 = {bce(d_real, 1), bce(d_fake, 0)} (means), g_real = w_real (sigmoid(d_real) - 1) / n, g_fake = w_fake sigmoid(d_fake) / n.
// One workgroup: wave-free fixed-order tree per term, exactly bce_logits_fwd_kernel's arithmetic.
__global__ __launch_bounds__(kBlock) void gan_disc_losses_kernel(const float* __restrict__ d_real, const float* __restrict__ d_fake, int n,
                                                                  float w_real, float w_fake, float* __restrict__ out2,
                                                                  float* __restrict__ g_real, float* __restrict__ g_fake) {
  __shared__ float red[kBlock];
  for (int term = 0; term < 2; ++term) {
    const float* x = term == 0 ? d_real : d_fake;
    const float target = term == 0 ? 1.f : 0.f, w = term == 0 ? w_real : w_fake;
    float* gx = term == 0 ? g_real : g_fake;
    float acc = 0.f;
    for (int i = threadIdx.x; i < n; i += kBlock) {
      const float v = x[i];
      const float ls = fminf(v, 0.f) - log1pf(expf(-fabsf(v)));
      acc += (1.f - target) * v - ls;
      const float sg = 1.f / (1.f + expf(-v));
      gx[i] = (sg - target) * w / (float)n;
    }
    const float tot = block_total<kBlock>(acc, red);
    if (threadIdx.x == 0) out2[term] = tot / (float)n;
  }
}

// ---- total = sum_k w_k term_k over up to 16 scalar device tensors (reference model/base.py:145-157: the weighted loss sum)
struct TermTable { const float* t[16]; float w[16]; int n; };
__global__ void weighted_sum_kernel(TermTable tb, float* __restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float acc = 0.f;
  for (int k = 0; k < tb.n; ++k) acc += tb.t[k][0] * tb.w[k];          // ascending k, like torch.dot on the stacked terms
  out[0] = acc;
}

// ---- the loss total and the step gate in one launch: weighted_sum_kernel followed by step_flags_kernel on its result
__global__ void weighted_sum_flags_kernel(TermTable tb, float* __restrict__ out, const int* status, int* bad, int n_bad, int word_status,
                                          int word_finite, int* snapshot, unsigned long long* step_counter) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (step_counter != nullptr) step_counter[0] += 1;              // this step's random draws are behind us on this stream
  float acc = 0.f;
  for (int k = 0; k < tb.n; ++k) acc += tb.t[k][0] * tb.w[k];
  out[0] = acc;
  if (status != nullptr && (status[0] & 1)) bad[word_status] |= 1;
  if (!(acc - acc == 0.f)) bad[word_finite] |= 1;
  for (int k = 0; k < n_bad; ++k) snapshot[k] = bad[k];
}


// ---- per-image latent rows (model/nerf_adapt_st_gan.py:589-593: Embedding.weight[var.idx]) of BOTH tables in one launch,
// and their gradient: dense [n_rows, C] tables with g[r] = sum over the images b with idx[b] == r, in ascending b (no
// atomics, no zero-fill launch; torch: index_select x 2 forward, zeros + index_add_ x 2 backward)
__global__ __launch_bounds__(kBlock) void latent_rows_fwd_kernel(tp_prologue::Rows q) {
  tp_prologue::latent_row_element(q, blockIdx.x * kBlock + threadIdx.x);
}
__global__ __launch_bounds__(kBlock) void latent_rows_bwd_kernel(const float* __restrict__ gt, const float* __restrict__ gl, const int64_t* __restrict__ idx,
                                                                  int B, int n_rows, int Ct, int Cl, float* __restrict__ gwt, float* __restrict__ gwl) {
  const int e = blockIdx.x * kBlock + threadIdx.x, C = Ct + Cl;
  if (e >= n_rows * C) return;
  const int r = e / C, c = e - r * C;
  float acc = 0.f;
  for (int b = 0; b < B; ++b)
    if (idx[b] == r) acc += c < Ct ? gt[b * Ct + c] : gl[b * Cl + (c - Ct)];
  if (c < Ct) gwt[r * Ct + c] = acc;
  else gwl[r * Cl + (c - Ct)] = acc;
}
}  // namespace

extern "C" {
int tp_step_flags(const int32_t* mlp_status, const float* total, int32_t* bad, int n_bad, int word_status, int word_finite,
                  int32_t* snapshot, tp_stream_t stream) {
  TP_REQUIRE(total && bad && snapshot && n_bad > 0 && word_finite >= 0 && word_finite < n_bad && (!mlp_status || (word_status >= 0 && word_status < n_bad)),
             "bad arguments");
  hipLaunchKernelGGL(step_flags_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, mlp_status, total, bad, n_bad, word_status, word_finite, snapshot);
  return tp::check_launch("tp_step_flags");
}

int tp_step_inputs(const tp_step_copy* copies, int n_copies, float* const* scalar_dst, const float* scalar_val, int n_scalars,
                   const int32_t* words_src, int32_t* words_dst, int n_words, tp_stream_t stream) {
  TP_REQUIRE(n_copies >= 0 && n_copies <= TP_STEP_INPUTS_MAX_COPIES && n_scalars >= 0 && n_scalars <= TP_STEP_INPUTS_MAX_SCALARS, "too many entries");
  TP_REQUIRE(n_words >= 0 && n_words <= 64 && (n_words == 0 || words_dst == nullptr || words_src != nullptr), "bad gate-word arguments");
  StepInputs t{};
  int64_t total16 = 0;
  for (int k = 0; k < n_copies; ++k) {
    TP_REQUIRE(copies[k].dst && copies[k].src && copies[k].bytes > 0, "null copy");
    TP_REQUIRE(((uintptr_t)copies[k].dst & 15) == 0 && ((uintptr_t)copies[k].src & 15) == 0, "copies must be 16-byte aligned");
    t.dst[k] = (char*)copies[k].dst; t.src[k] = (const char*)copies[k].src; t.bytes[k] = copies[k].bytes;
    total16 += (copies[k].bytes + 15) / 16;
    t.end16[k] = total16;
  }
  for (int k = n_copies; k < TP_STEP_INPUTS_MAX_COPIES; ++k) t.end16[k] = total16;
  t.n = n_copies;
  for (int k = 0; k < n_scalars; ++k) { TP_REQUIRE(scalar_dst[k] != nullptr, "null scalar"); t.sdst[k] = scalar_dst[k]; t.sval[k] = scalar_val[k]; }
  t.n_scalars = n_scalars;
  t.words_src = words_src; t.words_dst = n_words > 0 ? words_dst : nullptr; t.n_words = n_words;
  int64_t blocks = (total16 + kBlock - 1) / kBlock;
  if (blocks < 1) blocks = 1;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(step_inputs_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, t, total16);
  return tp::check_launch("tp_step_inputs");
}

int tp_grad_pack(const float* const* grads, const int64_t* numel, int n, float* flat, float scale, const int32_t* words, int n_words,
                 float* tail, tp_stream_t stream) {
  TP_REQUIRE(grads != nullptr && numel != nullptr && flat != nullptr && n > 0 && n <= TP_GRAD_PACK_MAX_TENSORS, "bad tensor table");
  TP_REQUIRE(n_words >= 0 && n_words <= 64 && (n_words == 0 || tail != nullptr), "bad gate-word arguments");
  GradPack t;
  int64_t total = 0;
  for (int k = 0; k < n; ++k) {
    TP_REQUIRE(numel[k] > 0, "empty tensor");
    t.src[k] = grads[k];
    total += numel[k];
    t.end[k] = total;
  }
  for (int k = n; k < TP_GRAD_PACK_MAX_TENSORS; ++k) { t.src[k] = nullptr; t.end[k] = total; }
  t.n = n;
  int64_t blocks = (total + 4 * kBlock - 1) / (4 * kBlock);
  if (blocks < 1) blocks = 1;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(grad_pack_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, t, flat, total, scale, words, n_words, tail);
  return tp::check_launch("tp_grad_pack");
}

// Diagnostic: the device's constant-rate clock (100 MHz) into *slot, as a launch of its own -- placed between the segments of a captured
// step it gives the step's timeline without a tracer slowing the host down (tools/linear_timeline.py).
__global__ void stamp_kernel(unsigned long long* slot) { *slot = wall_clock64(); }

// Diagnostic: the shader clock next to the constant 100 MHz clock over `windows` consecutive windows of `window_ticks` 100-MHz ticks,
// from ONE sleeping wave (out[2w] = shader cycles, out[2w+1] = 100-MHz ticks of window w).  Launched on a side stream in front of a
// render it tells what clock the chip HOLDS under that load (bench.py: roofline.clock_ghz; the f16x3 forward is power-limited well
// below the 2.4 GHz the peaks are quoted at).  The wave holds one SIMD slot of one CU for the whole span.
__global__ void clock_probe_kernel(unsigned long long* out, int windows, unsigned long long window_ticks) {
  for (int w = 0; w < windows; ++w) {
    const unsigned long long r0 = wall_clock64(), c0 = clock64();
    unsigned long long r1 = r0;
    while (r1 - r0 < window_ticks) {
      __builtin_amdgcn_s_sleep(64);
      r1 = wall_clock64();
    }
    out[2 * w] = clock64() - c0;
    out[2 * w + 1] = r1 - r0;
  }
}

int tp_clock_probe(uint64_t* out, int windows, int64_t window_us, tp_stream_t stream) {
  TP_REQUIRE(out != nullptr && windows > 0 && windows <= 4096 && window_us > 0 && window_us <= 1000000, "bad arguments");
  hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (unsigned long long*)out, windows,
                     (unsigned long long)window_us * 100ull);
  return tp::check_launch("tp_clock_probe");
}

// Diagnostic: nodes (kernel launches, fills, copies) recorded so far in the hipGraph `stream` is capturing into; -1 when it is not
// capturing.  Asked of the HIP runtime THIS library is linked against (the one the capturing process already runs on).
int64_t tp_capture_node_count(tp_stream_t stream) {
  hipStreamCaptureStatus status = hipStreamCaptureStatusNone;
  unsigned long long id = 0;
  hipGraph_t graph = nullptr;
  const hipGraphNode_t* deps = nullptr;
  size_t n_deps = 0, n = 0;
  if (hipStreamGetCaptureInfo_v2((hipStream_t)stream, &status, &id, &graph, &deps, &n_deps) != hipSuccess) { (void)hipGetLastError(); return -1; }
  if (status != hipStreamCaptureStatusActive || graph == nullptr) return -1;
  if (hipGraphGetNodes(graph, nullptr, &n) != hipSuccess) { (void)hipGetLastError(); return -1; }
  return (int64_t)n;
}

int tp_stamp(uint64_t* slot, tp_stream_t stream) {
  TP_REQUIRE(slot != nullptr, "null slot");
  hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (unsigned long long*)slot);
  return tp::check_launch("tp_stamp");
}

int tp_adam_step(const tp_adam_tensor* tensors, int n, const float* lr_dev, double lr_host, double beta1, double beta2, double eps,
                 const int32_t* gate, int n_gate, uint32_t* ticket, tp_stream_t stream) {
  TP_REQUIRE(tensors != nullptr && n > 0 && n <= TP_ADAM_MAX_TENSORS, "bad tensor table");
  TP_REQUIRE(ticket != nullptr, "ticket (a zero-filled device word owned by the calling stream) is required");
  TP_REQUIRE(n_gate == 0 || gate != nullptr, "gate words missing");
  TP_REQUIRE(n_gate >= 0 && n_gate <= 64, "at most 64 gate words");
  AdamTable t;
  int64_t total = 0;
  for (int k = 0; k < n; ++k) {
    TP_REQUIRE(tensors[k].param && tensors[k].grad && tensors[k].exp_avg && tensors[k].exp_avg_sq && tensors[k].step && tensors[k].numel > 0, "null tensor");
    t.p[k] = tensors[k].param; t.g[k] = tensors[k].grad; t.m[k] = tensors[k].exp_avg; t.v[k] = tensors[k].exp_avg_sq; t.step[k] = tensors[k].step;
    total += tensors[k].numel;
    t.end[k] = total;
  }
  for (int k = n; k < TP_ADAM_MAX_TENSORS; ++k) { t.p[k] = nullptr; t.g[k] = nullptr; t.m[k] = nullptr; t.v[k] = nullptr; t.step[k] = nullptr; t.end[k] = total; }
  t.n = n;
  int64_t blocks = (total + kBlock - 1) / kBlock;
  if (blocks > 256) blocks = 256;          // (one arrival per block on ONE counter word: 1,700 same-address atomics cost 16 us; 512 / 256 / 128
                                           //  workgroups: 915-925 / 931-932 / 926-929 it/s of the B=4 GAN iteration on one box)
  { static const int forced = [] { const char* e = getenv("TP_ADAM_BLOCKS"); return e ? atoi(e) : 0; }(); if (forced > 0 && blocks > forced) blocks = forced; }
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, t, lr_dev, lr_host, log(beta1), log(beta2),
                     (float)beta2, (float)eps, (float)(1.0 - beta1), (float)(1.0 - beta2), total, gate, n_gate, ticket);
  return tp::check_launch("tp_adam_step");
}

int tp_disc_inputs(const float* rgb, const float* gathered, int B, int P, int geo, float* real, float* fake, tp_stream_t stream) {
  TP_REQUIRE(rgb && gathered && real && fake && B > 0 && P > 0, "bad arguments");
  const int n = B * P;
  hipLaunchKernelGGL(disc_inputs_kernel, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, (hipStream_t)stream, rgb, gathered, B, P, geo, real, fake);
  return tp::check_launch("tp_disc_inputs");
}

int tp_patch_coords(const float* u, int B, int p, const float* lattice, const float* lo_dev, float lo_host, float span_host, float hi,
                    int random_scale, int random_shift, uint64_t seed, const uint64_t* counter, float* coords, float* scales,
                    tp_stream_t stream) {
  TP_REQUIRE(lattice && coords && scales && B > 0 && p > 0, "bad arguments");
  const int n = B * p * p;
  tp_prologue::Sampler q;
  q.u = u; q.lattice = lattice; q.lo_dev = lo_dev; q.counter = counter; q.coords = coords; q.scales = scales; q.seed = seed;
  q.lo_host = lo_host; q.span_host = span_host; q.hi = hi; q.B = B; q.p = p; q.random_scale = random_scale; q.random_shift = random_shift;
  hipLaunchKernelGGL(patch_coords_kernel, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, (hipStream_t)stream, q);
  return tp::check_launch("tp_patch_coords");
}

int tp_bce_logits_fwd(const float* x, int n, float target, float* out, tp_stream_t stream) {
  TP_REQUIRE(x && out && n > 0, "bad arguments");
  hipLaunchKernelGGL(bce_logits_fwd_kernel, dim3(1), dim3(kBlock), 0, (hipStream_t)stream, x, n, target, out);
  return tp::check_launch("tp_bce_logits_fwd");
}
int tp_bce_logits_bwd(const float* x, int n, float target, const float* g, float* gx, tp_stream_t stream) {
  TP_REQUIRE(x && g && gx && n > 0, "bad arguments");
  hipLaunchKernelGGL(bce_logits_bwd_kernel, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, (hipStream_t)stream, x, n, target, g, gx);
  return tp::check_launch("tp_bce_logits_bwd");
}

static int feat_fill(FeatP* q, const tp_feat_inputs_args* a) {
  TP_REQUIRE(a && a->rgb && a->gathered && a->B > 0 && a->P > 0 && a->n_channels > 0, "bad arguments");
  q->rgb = a->rgb; q->gathered = a->gathered; q->B = a->B; q->P = a->P; q->n_ch = a->n_channels;
  q->c_img = a->c_image; q->c_syn = a->c_image_syn; q->c_mask = a->c_mask; q->c_msyn = a->c_mask_syn;
  for (int c = 0; c < 3; ++c) { q->mean[c] = a->mean[c]; q->stdv[c] = a->std[c]; }
  return 0;
}
int tp_feat_inputs_fwd(const tp_feat_inputs_args* a, float* out, tp_stream_t stream) {
  FeatP q{};
  if (int rc = feat_fill(&q, a)) return rc;
  TP_REQUIRE(out, "out missing");
  q.out = out;
  const int n = a->B * a->P;
  hipLaunchKernelGGL(feat_inputs_fwd_kernel, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, (hipStream_t)stream, q);
  return tp::check_launch("tp_feat_inputs_fwd");
}
int tp_feat_inputs_bwd(const tp_feat_inputs_args* a, const float* g_out, float* g_rgb, tp_stream_t stream) {
  FeatP q{};
  if (int rc = feat_fill(&q, a)) return rc;
  TP_REQUIRE(g_out && g_rgb, "gradient pointers missing");
  q.g_out = g_out; q.g_rgb = g_rgb;
  const int n = a->B * a->P;
  hipLaunchKernelGGL(feat_inputs_bwd_kernel, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, (hipStream_t)stream, q);
  return tp::check_launch("tp_feat_inputs_bwd");
}

int tp_feat_pair_loss_fwd(const float* feat, int64_t n, float w2, float* out3, tp_stream_t stream) {
  TP_REQUIRE(feat && out3 && n > 0, "bad arguments");
  hipLaunchKernelGGL(feat_pair_loss_fwd_kernel, dim3(1), dim3(kRedBlock), 0, (hipStream_t)stream, feat, n, w2, out3);
  return tp::check_launch("tp_feat_pair_loss_fwd");
}
int tp_feat_pair_loss_bwd(const float* feat, int64_t n, float w2, const float* g, float* g_feat, tp_stream_t stream) {
  TP_REQUIRE(feat && g && g_feat && n > 0, "bad arguments");
  hipLaunchKernelGGL(feat_pair_loss_bwd_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, (hipStream_t)stream, feat, n, w2, g, g_feat);
  return tp::check_launch("tp_feat_pair_loss_bwd");
}
int tp_fake_patch_bwd(const float* g_fake, int B, int P, int nc, float* g_rgb, tp_stream_t stream) {
  TP_REQUIRE(g_fake && g_rgb && B > 0 && P > 0 && nc >= 3, "bad arguments");
  const int n = B * P;
  hipLaunchKernelGGL(fake_patch_bwd_kernel, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, (hipStream_t)stream, g_fake, B, P, nc, g_rgb);
  return tp::check_launch("tp_fake_patch_bwd");
}
int tp_sumsq_mean_fwd(const float* g, int64_t n, int B, float* out, tp_stream_t stream) {
  TP_REQUIRE(g && out && n > 0 && B > 0, "bad arguments");
  hipLaunchKernelGGL(sumsq_mean_fwd_kernel, dim3(1), dim3(kRedBlock), 0, (hipStream_t)stream, g, n, B, out);
  return tp::check_launch("tp_sumsq_mean_fwd");
}
int tp_maxpool2_fwd(const float* x, int64_t n, int H, int W, float* y, uint8_t* arg, tp_stream_t stream) {
  TP_REQUIRE(x && y && arg && n > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, "bad arguments (even H, W)");
  const int64_t n_out = n * (H / 2) * (W / 2);
  hipLaunchKernelGGL(maxpool2_fwd_kernel, dim3((unsigned)((n_out + kBlock - 1) / kBlock)), dim3(kBlock), 0, (hipStream_t)stream, x, n_out, H, W, y, arg);
  return tp::check_launch("tp_maxpool2_fwd");
}
int tp_maxpool2_bwd(const float* gy, const uint8_t* arg, int64_t n, int H, int W, float* gx, tp_stream_t stream) {
  TP_REQUIRE(gy && gx && arg && n > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, "bad arguments (even H, W)");
  const int64_t n_out = n * (H / 2) * (W / 2);
  hipLaunchKernelGGL(maxpool2_bwd_kernel, dim3((unsigned)((n_out + kBlock - 1) / kBlock)), dim3(kBlock), 0, (hipStream_t)stream, gy, arg, n_out, H, W, gx);
  return tp::check_launch("tp_maxpool2_bwd");
}
int tp_sumsq_mean_fwd_bwd(const float* g, int64_t n, int B, float w, float* out, float* out_g, tp_stream_t stream) {
  TP_REQUIRE(g && out && out_g && n > 0 && B > 0, "bad arguments");
  const int64_t blocks = (n + kRedBlock - 1) / kRedBlock;
  hipLaunchKernelGGL(sumsq_mean_fwd_bwd_kernel, dim3((unsigned)(blocks < 1024 ? blocks : 1024)), dim3(kRedBlock), 0, (hipStream_t)stream, g, n, B, w,
                     out, out_g);
  return tp::check_launch("tp_sumsq_mean_fwd_bwd");
}
int tp_gan_disc_losses(const float* d_real, const float* d_fake, int n, float w_real, float w_fake, float* out2, float* g_real,
                       float* g_fake, tp_stream_t stream) {
  TP_REQUIRE(d_real && d_fake && out2 && g_real && g_fake && n > 0, "bad arguments");
  hipLaunchKernelGGL(gan_disc_losses_kernel, dim3(1), dim3(kBlock), 0, (hipStream_t)stream, d_real, d_fake, n, w_real, w_fake, out2, g_real,
                     g_fake);
  return tp::check_launch("tp_gan_disc_losses");
}
int tp_sumsq_mean_bwd(const float* g, int64_t n, int B, const float* cot, float* out, tp_stream_t stream) {
  TP_REQUIRE(g && cot && out && n > 0 && B > 0, "bad arguments");
  hipLaunchKernelGGL(sumsq_mean_bwd_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, (hipStream_t)stream, g, n, B, cot, out);
  return tp::check_launch("tp_sumsq_mean_bwd");
}
int tp_latent_rows_fwd(const float* w_trans, const float* w_light, const int64_t* idx, int B, int C_trans, int C_light, float* out_trans,
                       float* out_light, int64_t* idx_copy, tp_stream_t stream) {
  TP_REQUIRE(w_trans && w_light && idx && out_trans && out_light && B > 0 && C_trans > 0 && C_light > 0, "bad arguments");
  const int n = B * (C_trans + C_light);
  tp_prologue::Rows q;
  q.wt = w_trans; q.wl = w_light; q.idx = idx; q.ot = out_trans; q.ol = out_light; q.idx_copy = idx_copy; q.B = B; q.Ct = C_trans; q.Cl = C_light;
  hipLaunchKernelGGL(latent_rows_fwd_kernel, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, (hipStream_t)stream, q);
  return tp::check_launch("tp_latent_rows_fwd");
}
int tp_latent_rows_bwd(const float* g_trans, const float* g_light, const int64_t* idx, int B, int n_rows, int C_trans, int C_light,
                       float* gw_trans, float* gw_light, tp_stream_t stream) {
  TP_REQUIRE(g_trans && g_light && idx && gw_trans && gw_light && B > 0 && n_rows > 0 && C_trans > 0 && C_light > 0, "bad arguments");
  const int n = n_rows * (C_trans + C_light);
  hipLaunchKernelGGL(latent_rows_bwd_kernel, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, (hipStream_t)stream, g_trans, g_light, idx, B,
                     n_rows, C_trans, C_light, gw_trans, gw_light);
  return tp::check_launch("tp_latent_rows_bwd");
}
int tp_weighted_sum_flags(const float* const* terms, const float* weights, int n, float* out, const int32_t* mlp_status, int32_t* bad,
                          int n_bad, int word_status, int word_finite, int32_t* snapshot, uint64_t* step_counter, tp_stream_t stream) {
  TP_REQUIRE(terms && weights && out && n > 0 && n <= 16, "1..16 terms expected");
  TP_REQUIRE(bad && snapshot && n_bad > 0 && word_finite >= 0 && word_finite < n_bad && (!mlp_status || (word_status >= 0 && word_status < n_bad)),
             "bad gate arguments");
  TermTable tb;
  for (int k = 0; k < 16; ++k) { tb.t[k] = k < n ? terms[k] : nullptr; tb.w[k] = k < n ? weights[k] : 0.f; }
  for (int k = 0; k < n; ++k) TP_REQUIRE(terms[k] != nullptr, "null term");
  tb.n = n;
  hipLaunchKernelGGL(weighted_sum_flags_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, tb, out, mlp_status, bad, n_bad, word_status,
                     word_finite, snapshot, (unsigned long long*)step_counter);
  return tp::check_launch("tp_weighted_sum_flags");
}
int tp_weighted_sum(const float* const* terms, const float* weights, int n, float* out, tp_stream_t stream) {
  TP_REQUIRE(terms && weights && out && n > 0 && n <= 16, "1..16 terms expected");
  TermTable tb;
  for (int k = 0; k < 16; ++k) { tb.t[k] = k < n ? terms[k] : nullptr; tb.w[k] = k < n ? weights[k] : 0.f; }
  for (int k = 0; k < n; ++k) TP_REQUIRE(terms[k] != nullptr, "null term");
  tb.n = n;
  hipLaunchKernelGGL(weighted_sum_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, tb, out);
  return tp::check_launch("tp_weighted_sum");
}
}
