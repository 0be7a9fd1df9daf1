// K8: the three render-consuming loss terms of the generator step in one launch each way:
//   render    = sum mask (image - rgb)^2 / uncert^2 / (sum mask + 1e-5)        (nerf.mask_obj, :747-752)
//   uncert    = 5 + mean(log uncert^2) / 2                                      (:757-758)
//   trans_reg = mean sigma_transient                                            (:759-760)
// reference model/nerf_adapt_st_gan.py:712-776 (compute_loss, train_step == 'nerf').  PyTorch issues ~30 elementwise /
// reduction kernels for the forward and as many for the backward; here: one pass producing four sums (fixed-order
// two-stage reduction: deterministic), and one pass producing d/d rgb, d/d uncert, d/d density.
// Inputs are read where the render and the patch gather left them: rgb [B,P,3] / uncert [B,P] / density [B,P,N,2]
// from the composite / MLP outputs, image and mask as channels 0..2 and 12 of the gather output [B,14,P].
#include "tp_common.h"

namespace {
constexpr int kBlock = 256;

__device__ __forceinline__ void block_reduce4(float (&v)[4], float* red) {   // red: [4][kBlock]
  const int tid = threadIdx.x;
#pragma unroll
  for (int k = 0; k < 4; ++k) red[k * kBlock + tid] = v[k];
  __syncthreads();
  for (int s = kBlock >> 1; s > 0; s >>= 1) {
    if (tid < s) {
#pragma unroll
      for (int k = 0; k < 4; ++k) red[k * kBlock + tid] += red[k * kBlock + tid + s];
    }
    __syncthreads();
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) v[k] = red[k * kBlock];
}

// fixed-order reduction of the per-block partials: thread t adds partials t, t+256, ... in order, then a block tree
__device__ __forceinline__ void nerf_losses_finalize(const float* partial, int n_blocks, double* sums, float* losses, float n_pix, float n_den,
                                                     double (*red)[kBlock]) {
  const int tid = threadIdx.x;
  double v[4] = {0.0, 0.0, 0.0, 0.0};
  for (int i = tid; i < n_blocks; i += kBlock) {
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] += (double)__hip_atomic_load(partial + i * 4 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) red[k][tid] = v[k];
  __syncthreads();
  for (int s = kBlock >> 1; s > 0; s >>= 1) {
    if (tid < s) {
#pragma unroll
      for (int k = 0; k < 4; ++k) red[k][tid] += red[k][tid + s];
    }
    __syncthreads();
  }
  if (tid < 4) sums[tid] = red[tid][0];
  if (tid == 0 && losses != nullptr) {
    const float s0 = (float)red[0][0], s1 = (float)red[1][0], s2 = (float)red[2][0], s3 = (float)red[3][0];
    losses[0] = tp::div_rn(s0, tp::add_rn(s1, 1e-5f));
    losses[1] = tp::add_rn(5.f, tp::div_rn(tp::div_rn(s2, n_pix), 2.f));
    losses[2] = tp::div_rn(s3, n_den);
  }
}

// The block that finishes LAST reduces the partials of all of them (a ticket in CALLER-OWNED device memory -- `args.ticket`, one
// word per stream that may run this entry point, zero between launches, reset by that block: two launches that overlap on
// different streams must not count each other's arrivals): the forward is one launch, and the order of the final reduction is
// the same whichever block runs it.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "the last-block hand-over below relies on the gfx950 agent-scope store / load contract (csrc/patch_conv.hip)"
#endif
__global__ __launch_bounds__(kBlock) void nerf_losses_fwd_kernel(tp_nerf_losses_args a, float* partial, float n_pix_f, float n_den_f) {
  __shared__ float red[4 * kBlock];
  __shared__ double red_d[4][kBlock];
  __shared__ bool last;
  const int64_t n_pix = (int64_t)a.B * a.P, n_den = n_pix * a.N;
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  for (int64_t q = (int64_t)blockIdx.x * kBlock + threadIdx.x; q < n_pix; q += (int64_t)gridDim.x * kBlock) {
    const int64_t b = q / a.P, p = q - b * a.P;
    const float* gp = a.gathered + b * 14 * a.P + p;
    const float m = gp[12 * (int64_t)a.P], u = a.uncert[q];
    float se = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float d = gp[c * (int64_t)a.P] - a.rgb[q * 3 + c];
      se += d * d;
    }
    v[0] += m * (se / (u * u));
    v[1] += m;
    v[2] += logf(u * u);
  }
  const float2* den = reinterpret_cast<const float2*>(a.density);
  for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < n_den; e += (int64_t)gridDim.x * kBlock) v[3] += den[e].y;
  block_reduce4(v, red);
  // hand-over without device-scope fences (the gfx950 contract of csrc/patch_conv.hip reduce_tiles: agent-scope stores, the storing
  // lane drained, ONE agent-scope counter add; the last arriver loads agent-scope)
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < 4; ++k) __hip_atomic_store(partial + blockIdx.x * 4 + k, v[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    last = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
  }
  __syncthreads();
  if (!last) return;
  if (threadIdx.x == 0) __hip_atomic_store(a.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  nerf_losses_finalize(partial, (int)gridDim.x, a.sums, a.losses, n_pix_f, n_den_f, red_d);
}


// tp_nerf_losses_bwd_total: the generator step's loss total + step gate (csrc/train_misc.hip weighted_sum_flags_kernel, same arithmetic
// and order of side effects) as a side job of this launch's first thread -- the total is not an input of any gradient, only of the gate
// the optimiser launch reads, so it needs no launch of its own on the render's backward chain
constexpr int kMaxBadWords = 8;
struct TotalJob {
  const float* t[16]; float w[16]; int n;
  float* out; const int* status; int* bad; int* snapshot; unsigned long long* step_counter;
  int n_bad, word_status, word_finite;
};

// g_render / g_unc / g_trans: upstream gradients of (render, uncert, trans_reg), one device scalar each (NULL = 0)
__global__ __launch_bounds__(kBlock) void nerf_losses_bwd_kernel(tp_nerf_losses_args a, const double* sums, const float* g_render,
                                                                 const float* g_unc, const float* g_trans,
                                                                 float* g_rgb, float* g_uncert, float* g_density, TotalJob job) {
  if (job.n > 0 && blockIdx.x == 0 && threadIdx.x == 0) {
    // every load of the job issued before the first use (the terms, the status word, the gate words, the counter: up to 26 independent
    // loads in flight instead of a chain of ~20 round trips -- this thread's job IS the launch's duration at the B=4 training size)
    float tv[16];
    int bw[kMaxBadWords];
#pragma unroll
    for (int k = 0; k < 16; ++k) tv[k] = k < job.n ? job.t[k][0] : 0.f;
#pragma unroll
    for (int k = 0; k < kMaxBadWords; ++k) bw[k] = k < job.n_bad ? job.bad[k] : 0;
    const int st = job.status != nullptr ? job.status[0] : 0;
    const unsigned long long cnt = job.step_counter != nullptr ? job.step_counter[0] : 0ull;
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) if (k < job.n) acc += tv[k] * job.w[k];
    job.out[0] = acc;
    if (job.step_counter != nullptr) job.step_counter[0] = cnt + 1;
    const bool ranged = (st & 1) != 0, nonfinite = !(acc - acc == 0.f);
#pragma unroll
    for (int k = 0; k < kMaxBadWords; ++k) {
      if (k >= job.n_bad) break;
      const int v = bw[k] | ((ranged && k == job.word_status) || (nonfinite && k == job.word_finite) ? 1 : 0);
      if (v != bw[k]) job.bad[k] = v;
      job.snapshot[k] = v;
    }
  }
  const int64_t n_pix = (int64_t)a.B * a.P, n_den = n_pix * a.N;
  const float inv_den = (float)(1.0 / (sums[1] + 1e-5));
  const float gr = (g_render ? g_render[0] : 0.f) * inv_den, gu = (g_unc ? g_unc[0] : 0.f) / (float)n_pix,
              gt = (g_trans ? g_trans[0] : 0.f) / (float)n_den;
  for (int64_t q = (int64_t)blockIdx.x * kBlock + threadIdx.x; q < n_pix; q += (int64_t)gridDim.x * kBlock) {
    const int64_t b = q / a.P, p = q - b * a.P;
    const float* gp = a.gathered + b * 14 * a.P + p;
    const float m = gp[12 * (int64_t)a.P], u = a.uncert[q];
    const float iu2 = 1.0f / (u * u);
    float se = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float d = a.rgb[q * 3 + c] - gp[c * (int64_t)a.P];
      se += d * d;
      g_rgb[q * 3 + c] = gr * 2.0f * m * d * iu2;
    }
    g_uncert[q] = -2.0f * gr * m * se * iu2 / u + gu / u;
  }
  float2* gd = reinterpret_cast<float2*>(g_density);
  for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < n_den; e += (int64_t)gridDim.x * kBlock) gd[e] = make_float2(0.0f, gt);
}

int grid_for(const tp_nerf_losses_args* a) {
  const int64_t n = (int64_t)a->B * a->P * a->N;
  const int64_t g = (n + kBlock - 1) / kBlock;
  return (int)(g < 1 ? 1 : (g > TP_NERF_LOSSES_MAX_BLOCKS ? TP_NERF_LOSSES_MAX_BLOCKS : g));
}
int check(const tp_nerf_losses_args* a, const char* what) {
  if (!a || !a->rgb || !a->uncert || !a->density || !a->gathered || !a->workspace || !a->sums) { tp::set_error("%s: null pointer", what); return -1; }
  if (a->B <= 0 || a->P <= 0 || a->N <= 0) { tp::set_error("%s: bad sizes", what); return -1; }
  return 0;
}
}  // namespace

extern "C" int tp_nerf_losses_fwd(const tp_nerf_losses_args* a, tp_stream_t stream) {
  if (int rc = check(a, "tp_nerf_losses_fwd")) return rc;
  const int g = grid_for(a);
  if (!a->ticket) { tp::set_error("tp_nerf_losses_fwd: args.ticket (a zero-filled device word owned by the calling stream) is required"); return -1; }
  hipLaunchKernelGGL(nerf_losses_fwd_kernel, dim3(g), dim3(kBlock), 0, (hipStream_t)stream, *a, (float*)a->workspace,
                     (float)((int64_t)a->B * a->P), (float)((int64_t)a->B * a->P * a->N));
  return tp::check_launch("tp_nerf_losses_fwd");
}

extern "C" int tp_nerf_losses_bwd(const tp_nerf_losses_args* a, const float* g_render, const float* g_unc, const float* g_trans,
                                  float* g_rgb, float* g_uncert, float* g_density, tp_stream_t stream) {
  if (int rc = check(a, "tp_nerf_losses_bwd")) return rc;
  if (!g_rgb || !g_uncert || !g_density) { tp::set_error("tp_nerf_losses_bwd: null gradient pointer"); return -1; }
  hipLaunchKernelGGL(nerf_losses_bwd_kernel, dim3(grid_for(a)), dim3(kBlock), 0, (hipStream_t)stream, *a, (const double*)a->sums,
                     g_render, g_unc, g_trans, g_rgb, g_uncert, g_density, TotalJob{});
  return tp::check_launch("tp_nerf_losses_bwd");
}

extern "C" int tp_nerf_losses_bwd_total(const tp_nerf_losses_args* a, const float* g_render, const float* g_unc, const float* g_trans,
                                        float* g_rgb, float* g_uncert, float* g_density, const float* const* terms, const float* weights,
                                        int n, float* out, const int32_t* mlp_status, int32_t* bad, int n_bad, int word_status,
                                        int word_finite, int32_t* snapshot, uint64_t* step_counter, tp_stream_t stream) {
  if (int rc = check(a, "tp_nerf_losses_bwd_total")) return rc;
  if (!g_rgb || !g_uncert || !g_density) { tp::set_error("tp_nerf_losses_bwd_total: null gradient pointer"); return -1; }
  TP_REQUIRE(terms && weights && out && n > 0 && n <= 16, "1..16 terms expected");
  TP_REQUIRE(n_bad <= kMaxBadWords, "at most 8 gate words");
  TP_REQUIRE(bad && snapshot && n_bad > 0 && word_finite >= 0 && word_finite < n_bad && (!mlp_status || (word_status >= 0 && word_status < n_bad)),
             "bad gate arguments");
  TotalJob job{};
  for (int k = 0; k < n; ++k) { TP_REQUIRE(terms[k] != nullptr, "null term"); job.t[k] = terms[k]; job.w[k] = weights[k]; }
  job.n = n; job.out = out; job.status = mlp_status; job.bad = bad; job.snapshot = snapshot; job.step_counter = (unsigned long long*)step_counter;
  job.n_bad = n_bad; job.word_status = word_status; job.word_finite = word_finite;
  hipLaunchKernelGGL(nerf_losses_bwd_kernel, dim3(grid_for(a)), dim3(kBlock), 0, (hipStream_t)stream, *a, (const double*)a->sums,
                     g_render, g_unc, g_trans, g_rgb, g_uncert, g_density, job);
  return tp::check_launch("tp_nerf_losses_bwd_total");
}
