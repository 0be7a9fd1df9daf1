// K7: spectral-norm weight normalisation of the PatchGAN's six convolutions in a handful of launches.
//
// The reference wraps every discriminator conv in torch.nn.utils.spectral_norm (layers/discriminator.py:45-115).  In
// training mode every forward runs one power iteration per weight,
//     v <- normalize(W^T u),  u <- normalize(W v),  sigma = u . (W v),  W_sn = W / sigma       (W = weight.view(out, -1))
// which PyTorch issues as ~15 tiny kernels per weight (mv, norm, clamp, div, dot, clones), i.e. ~90 per discriminator
// forward and as many again in the backward -- ~600 of the ~1,350 launches of a training iteration.  Here all weights
// of the module are processed together: five launches per forward (two in eval mode + scale) -- or three with arrival counters
// (A + A2, B + B2 fused: the last workgroup of a weight normalises; measured 1.4 % slower per GAN iteration: off by default) --,
// two per backward,
//     fwd :  A  t_part[slab] = W[slab rows]^T u      (column blocks x row slabs, coalesced rows)
//            A2 v = normalize(sum_slabs t_part)       (one workgroup per weight; updates the module's v buffer)
//            B  s = W v                               (one wave per row, lanes stride the columns)
//            B2 u = normalize(s), sigma = u . s       (one workgroup per weight; updates the module's u buffer)
//            C  W_sn = W / sigma
//     bwd :  D  partial sums of <G, W_sn>;   E  dW = (G - <G, W_sn> u v^T) / sigma        (u, v constants, as in torch)
// All reductions run in a fixed order (no atomics): the training run is reproducible.
#include "tp_common.h"

namespace {
constexpr int kMaxW = TP_SN_MAX_WEIGHTS;
constexpr int kSlabRows = 64;

struct Batch {
  tp_sn_weight w[kMaxW];
  int n;
  int blk_a[kMaxW + 1];   // prefix sums of workgroups per weight for the kernel being launched
  unsigned int* tickets;  // fused kernels: [2][kMaxW] arrival counters (A, B), zero between launches
};

__device__ __forceinline__ int find_weight(const Batch& b, int blk, int& local) {
  int i = 0;
  while (i + 1 < b.n && blk >= b.blk_a[i + 1]) ++i;
  local = blk - b.blk_a[i];
  return i;
}

__device__ __forceinline__ float block_sum(float v, float* red) {
  const int tid = threadIdx.x;
  red[tid] = v;
  __syncthreads();
  for (int s = blockDim.x >> 1; s > 0; s >>= 1) {
    if (tid < s) red[tid] += red[tid + s];
    __syncthreads();
  }
  const float r = red[0];
  __syncthreads();
  return r;
}

// A: t_part[slab][c] = sum_{r in slab} W[r][c] u[r]
__global__ __launch_bounds__(256) void sn_wtu_kernel(Batch b) {
  __shared__ float us[kSlabRows];
  int local;
  const tp_sn_weight& w = b.w[find_weight(b, blockIdx.x, local)];
  const int cblocks = (w.cols + 255) / 256;
  const int slab = local / cblocks, cb = local - slab * cblocks;
  const int r0 = slab * kSlabRows, nr = min(kSlabRows, w.rows - r0);
  if (threadIdx.x < nr) us[threadIdx.x] = w.u[r0 + threadIdx.x];
  __syncthreads();
  const int c = cb * 256 + threadIdx.x;
  if (c >= w.cols) return;
  float acc = 0.0f;
#pragma unroll 8
  for (int r = 0; r < nr; ++r) acc = fmaf(w.weight[(int64_t)(r0 + r) * w.cols + c], us[r], acc);   // 8 loads in flight
  w.work[(int64_t)slab * w.cols + c] = acc;
}

// A2: v = normalize(sum_slabs t_part)
__global__ __launch_bounds__(1024) void sn_v_kernel(Batch b) {
  __shared__ float red[1024];
  const tp_sn_weight& w = b.w[blockIdx.x];
  const int slabs = (w.rows + kSlabRows - 1) / kSlabRows;
  float ss = 0.0f;
  for (int c = threadIdx.x; c < w.cols; c += 1024) {
    float t = 0.0f;
    for (int s = 0; s < slabs; ++s) t += w.work[(int64_t)s * w.cols + c];
    w.v[c] = t;
    ss = fmaf(t, t, ss);
  }
  const float nrm = fmaxf(sqrtf(block_sum(ss, red)), 1e-12f);
  for (int c = threadIdx.x; c < w.cols; c += 1024) w.v[c] = w.v[c] / nrm;
}

// B: s[r] = sum_c W[r][c] v[c]   (one wave per row)
__global__ __launch_bounds__(256) void sn_wv_kernel(Batch b) {
  int local;
  const tp_sn_weight& w = b.w[find_weight(b, blockIdx.x, local)];
  const int r = local * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= w.rows) return;
  const float* row = w.weight + (int64_t)r * w.cols;
  float acc = 0.0f;
#pragma unroll 8
  for (int c = lane; c < w.cols; c += 64) acc = fmaf(row[c], w.v[c], acc);                          // 16 loads in flight
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if (lane == 0) w.work[(int64_t)TP_SN_MAX_SLABS * w.cols + r] = acc;
}

// B2: u = normalize(s) (training) ; sigma = u . s
__global__ __launch_bounds__(256) void sn_u_kernel(Batch b, int training) {
  __shared__ float red[256];
  const tp_sn_weight& w = b.w[blockIdx.x];
  const float* s = w.work + (int64_t)TP_SN_MAX_SLABS * w.cols;
  if (training) {
    float ss = 0.0f;
    for (int r = threadIdx.x; r < w.rows; r += 256) ss = fmaf(s[r], s[r], ss);
    const float nrm = fmaxf(sqrtf(block_sum(ss, red)), 1e-12f);
    for (int r = threadIdx.x; r < w.rows; r += 256) w.u[r] = s[r] / nrm;
    __syncthreads();
  }
  float d = 0.0f;
  for (int r = threadIdx.x; r < w.rows; r += 256) d = fmaf(w.u[r], s[r], d);
  d = block_sum(d, red);
  if (threadIdx.x == 0) *w.sigma = d;
}

// ---- A + A2 and B + B2 as ONE launch each (tp_sn_fwd with `tickets`): the workgroup of a weight that arrives LAST normalises.
// Hand-over by the gfx950 contract of csrc/patch_conv.hip (reduce_tiles): handed-off words are stored and loaded agent-scope,
// the storing lanes drained, one agent-scope counter add per workgroup behind a barrier; no device-scope fence (on gfx950 that
// is a write-back + invalidate of the XCD's whole L2).  The reductions keep a fixed order: whichever workgroup is last runs the
// same code over the same partials.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "the fused spectral-norm kernels rely on the gfx950 agent-scope store / load hand-over (csrc/patch_conv.hip)"
#endif
__device__ __forceinline__ bool arrive_last(unsigned int* ticket, int n_blocks) {
  __shared__ int last;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this thread's stores are complete
  __syncthreads();
  if (threadIdx.x == 0) {
    last = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(n_blocks - 1);
    if (last) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch
  }
  __syncthreads();
  return last != 0;
}
__device__ __forceinline__ float ld_agent(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__global__ __launch_bounds__(256) void sn_wtu_v_kernel(Batch b) {
  __shared__ float us[kSlabRows];
  __shared__ float red[256];
  int local;
  const int wi = find_weight(b, blockIdx.x, local);
  const tp_sn_weight& w = b.w[wi];
  const int cblocks = (w.cols + 255) / 256;
  const int slab = local / cblocks, cb = local - slab * cblocks;
  const int r0 = slab * kSlabRows, nr = min(kSlabRows, w.rows - r0);
  if (threadIdx.x < nr) us[threadIdx.x] = w.u[r0 + threadIdx.x];
  __syncthreads();
  const int c = cb * 256 + threadIdx.x;
  if (c < w.cols) {
    float acc = 0.0f;
#pragma unroll 8
    for (int r = 0; r < nr; ++r) acc = fmaf(w.weight[(int64_t)(r0 + r) * w.cols + c], us[r], acc);
    st_agent(w.work + (int64_t)slab * w.cols + c, acc);
  }
  if (!arrive_last(b.tickets + wi, b.blk_a[wi + 1] - b.blk_a[wi])) return;
  // A2 (one workgroup per weight): v = normalize(sum_slabs t_part).  The agent-scope loads travel past the L2 (~2 us each): 32 of
  // them (4 columns x 8 slabs) are issued before the first is used; summed slab-ascending as in sn_v_kernel (absent slabs add 0).
  const int slabs = (w.rows + kSlabRows - 1) / kSlabRows;
  float ss = 0.0f;
  for (int c0 = threadIdx.x; c0 < w.cols; c0 += 256 * 4) {
    float part[4][TP_SN_MAX_SLABS];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int sl = 0; sl < TP_SN_MAX_SLABS; ++sl)
        part[j][sl] = (c0 + 256 * j < w.cols && sl < slabs) ? ld_agent(w.work + (int64_t)sl * w.cols + c0 + 256 * j) : 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (c0 + 256 * j >= w.cols) continue;
      float t = 0.0f;
#pragma unroll
      for (int sl = 0; sl < TP_SN_MAX_SLABS; ++sl) t += part[j][sl];
      w.v[c0 + 256 * j] = t;
      ss = fmaf(t, t, ss);
    }
  }
  const float nrm = fmaxf(sqrtf(block_sum(ss, red)), 1e-12f);
  for (int cc = threadIdx.x; cc < w.cols; cc += 256) w.v[cc] = w.v[cc] / nrm;
}

__global__ __launch_bounds__(256) void sn_wv_u_kernel(Batch b, int training) {
  __shared__ float red[256];
  int local;
  const int wi = find_weight(b, blockIdx.x, local);
  const tp_sn_weight& w = b.w[wi];
  const int r = local * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  float* s = w.work + (int64_t)TP_SN_MAX_SLABS * w.cols;
  if (r < w.rows) {
    const float* row = w.weight + (int64_t)r * w.cols;
    float acc = 0.0f;
#pragma unroll 8
    for (int c = lane; c < w.cols; c += 64) acc = fmaf(row[c], w.v[c], acc);
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (lane == 0) st_agent(s + r, acc);
  }
  if (!arrive_last(b.tickets + kMaxW + wi, b.blk_a[wi + 1] - b.blk_a[wi])) return;
  // B2 (one workgroup per weight): u = normalize(s) (training); sigma = u . s   (rows <= 512: two values per thread, loaded once)
  const int r_a = threadIdx.x, r_b = threadIdx.x + 256;
  const float s_a = r_a < w.rows ? ld_agent(s + r_a) : 0.0f, s_b = r_b < w.rows ? ld_agent(s + r_b) : 0.0f;
  float u_a = r_a < w.rows ? w.u[r_a] : 0.0f, u_b = r_b < w.rows ? w.u[r_b] : 0.0f;
  if (training) {
    const float ss = fmaf(s_b, s_b, fmaf(s_a, s_a, 0.0f));
    const float nrm = fmaxf(sqrtf(block_sum(ss, red)), 1e-12f);
    u_a = s_a / nrm; u_b = s_b / nrm;
    if (r_a < w.rows) w.u[r_a] = u_a;
    if (r_b < w.rows) w.u[r_b] = u_b;
  }
  float d = fmaf(u_b, s_b, fmaf(u_a, s_a, 0.0f));
  d = block_sum(d, red);
  if (threadIdx.x == 0) *w.sigma = d;
}

// C: W_sn = W / sigma
__global__ __launch_bounds__(256) void sn_scale_kernel(Batch b) {
  int local;
  const tp_sn_weight& w = b.w[find_weight(b, blockIdx.x, local)];
  const int64_t n = (int64_t)w.rows * w.cols;
  const float sg = *w.sigma;
  for (int64_t i = (int64_t)local * 1024 + threadIdx.x; i < min(n, (int64_t)(local + 1) * 1024); i += 256)
    w.weight_sn[i] = w.weight[i] / sg;
  if (local == 0) {                     // this forward's u / v for its backward (one workgroup per weight)
    if (w.u_out) for (int r = threadIdx.x; r < w.rows; r += 256) w.u_out[r] = w.u[r];
    if (w.v_out) for (int c = threadIdx.x; c < w.cols; c += 256) w.v_out[c] = w.v[c];
  }
}

// D: per-workgroup partial of <G, W_sn>
__global__ __launch_bounds__(256) void sn_dot_kernel(Batch b) {
  __shared__ float red[256];
  int local;
  const tp_sn_weight& w = b.w[find_weight(b, blockIdx.x, local)];
  const int64_t n = (int64_t)w.rows * w.cols;
  float acc = 0.0f;
  for (int64_t i = (int64_t)local * 4096 + threadIdx.x; i < min(n, (int64_t)(local + 1) * 4096); i += 256)
    acc = fmaf(w.grad_sn[i], w.weight_sn[i], acc);
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) w.work[local] = acc;
}

// E: dW = (G - <G, W_sn> u v^T) / sigma
__global__ __launch_bounds__(256) void sn_grad_kernel(Batch b) {
  __shared__ float red[256];
  int local;
  const tp_sn_weight& w = b.w[find_weight(b, blockIdx.x, local)];
  const int64_t n = (int64_t)w.rows * w.cols;
  const int parts = (int)((n + 4095) / 4096);
  float d = 0.0f;
  for (int i = threadIdx.x; i < parts; i += 256) d += w.work[i];
  d = block_sum(d, red);
  const float sg = *w.sigma;
  for (int64_t i = (int64_t)local * 1024 + threadIdx.x; i < min(n, (int64_t)(local + 1) * 1024); i += 256) {
    const int r = (int)(i / w.cols), c = (int)(i - (int64_t)r * w.cols);
    const float gv = (w.grad_sn[i] - d * w.u[r] * w.v[c]) / sg;
    w.grad[i] = w.accumulate ? w.grad[i] + gv : gv;
  }
}

int fill(Batch& b, const tp_sn_weight* ws, int n, int (*blocks)(const tp_sn_weight&)) {
  b.n = n;
  b.blk_a[0] = 0;
  for (int i = 0; i < n; ++i) { b.w[i] = ws[i]; b.blk_a[i + 1] = b.blk_a[i] + blocks(ws[i]); }
  return b.blk_a[n];
}
int check(const tp_sn_weight* ws, int n, bool bwd, const char* what) {
  if (ws == nullptr || n <= 0 || n > kMaxW) { tp::set_error("%s: 1..%d weights expected", what, kMaxW); return -1; }
  for (int i = 0; i < n; ++i) {
    const tp_sn_weight& w = ws[i];
    if (w.rows <= 0 || w.cols <= 0 || w.rows > TP_SN_MAX_SLABS * kSlabRows) { tp::set_error("%s: bad sizes (rows <= %d)", what, TP_SN_MAX_SLABS * kSlabRows); return -1; }
    if (!w.u || !w.v || !w.sigma || !w.weight_sn || !w.work) { tp::set_error("%s: null pointer", what); return -1; }
    if (!bwd && !w.weight) { tp::set_error("%s: null weight", what); return -1; }
    if (bwd && (!w.grad_sn || !w.grad)) { tp::set_error("%s: null gradient pointer", what); return -1; }
  }
  return 0;
}
}  // namespace

extern "C" int64_t tp_sn_work_floats(int rows, int cols) {
  const int64_t fwd = (int64_t)TP_SN_MAX_SLABS * cols + rows, bwd = ((int64_t)rows * cols + 4095) / 4096;
  return fwd > bwd ? fwd : bwd;
}

extern "C" int tp_sn_fwd(const tp_sn_weight* ws, int n, int training, uint32_t* tickets, tp_stream_t stream) {
  if (int rc = check(ws, n, false, "tp_sn_fwd")) return rc;
  hipStream_t st = (hipStream_t)stream;
  Batch b;
  b.tickets = tickets;
  if (training) {
    int g = fill(b, ws, n, [](const tp_sn_weight& w) { return ((w.cols + 255) / 256) * ((w.rows + kSlabRows - 1) / kSlabRows); });
    if (tickets) {
      hipLaunchKernelGGL(sn_wtu_v_kernel, dim3(g), dim3(256), 0, st, b);
    } else {
      hipLaunchKernelGGL(sn_wtu_kernel, dim3(g), dim3(256), 0, st, b);
      hipLaunchKernelGGL(sn_v_kernel, dim3(n), dim3(1024), 0, st, b);
    }
  }
  int g = fill(b, ws, n, [](const tp_sn_weight& w) { return (w.rows + 3) / 4; });
  if (tickets) {
    hipLaunchKernelGGL(sn_wv_u_kernel, dim3(g), dim3(256), 0, st, b, training);
  } else {
    hipLaunchKernelGGL(sn_wv_kernel, dim3(g), dim3(256), 0, st, b);
    hipLaunchKernelGGL(sn_u_kernel, dim3(n), dim3(256), 0, st, b, training);
  }
  g = fill(b, ws, n, [](const tp_sn_weight& w) { return (int)(((int64_t)w.rows * w.cols + 1023) / 1024); });
  hipLaunchKernelGGL(sn_scale_kernel, dim3(g), dim3(256), 0, st, b);
  return tp::check_launch("tp_sn_fwd");
}

extern "C" int tp_sn_bwd(const tp_sn_weight* ws, int n, tp_stream_t stream) {
  if (int rc = check(ws, n, true, "tp_sn_bwd")) return rc;
  hipStream_t st = (hipStream_t)stream;
  Batch b;
  b.tickets = nullptr;
  int g = fill(b, ws, n, [](const tp_sn_weight& w) { return (int)(((int64_t)w.rows * w.cols + 4095) / 4096); });
  hipLaunchKernelGGL(sn_dot_kernel, dim3(g), dim3(256), 0, st, b);
  g = fill(b, ws, n, [](const tp_sn_weight& w) { return (int)(((int64_t)w.rows * w.cols + 1023) / 1024); });
  hipLaunchKernelGGL(sn_grad_kernel, dim3(g), dim3(256), 0, st, b);
  return tp::check_launch("tp_sn_bwd");
}
