// K7: spectral-norm weight normalisation of the PatchGAN's six convolutions in a handful of launches.
//
// The reference wraps every discriminator conv in torch.nn.utils.spectral_norm (layers/discriminator.py:45-115).  In
// training mode every forward runs one power iteration per weight,
//     v <- normalize(W^T u),  u <- normalize(W v),  sigma = u . (W v),  W_sn = W / sigma       (W = weight.view(out, -1))
// which PyTorch issues as ~15 tiny kernels per weight (mv, norm, clamp, div, dot, clones), i.e. ~90 per discriminator
// forward and as many again in the backward -- ~600 of the ~1,350 launches of a training iteration.  Here all weights
// of the module are processed together: three launches per forward (two in eval mode), two per backward,
//     fwd :  A  t_part[slab] = W[slab rows]^T u      (column blocks x row slabs, coalesced rows)
//            B  v = normalize(sum_slabs t_part) in LDS, then s = W v for the workgroup's 8 rows   (every workgroup of a weight
//               normalises for itself -- the slab sums are L2 hits --; the first one writes the module's v buffer)
//            C  u = normalize(s), sigma = u . s, then W_sn = W / sigma for the workgroup's 4096 elements   (every workgroup
//               reduces the <= 512 values of s itself, identically; the first one writes the module's u buffer and sigma)
//     bwd :  D  partial sums of <G, W_sn>;   E  dW = (G - <G, W_sn> u v^T) / sigma        (u, v constants, as in torch)
//            a SECOND normalised instance of the same weight (the discriminator step normalises twice per optimiser step: real
//            pass, fake pass) goes through the same two launches: D forms both dot products, E adds both terms.
// Rounds 1-3 normalised in launches of their own (A2, B2: one workgroup per weight, 13.6 + 5 us of the 44 us of a set), round 3's
// variant with arrival counters (the last workgroup of a weight normalises) paid the same in agent-scope loads past the L2.
// All reductions run in a fixed order (no atomics): the training run is reproducible.
#include "tp_common.h"

namespace {
constexpr int kMaxW = TP_SN_MAX_WEIGHTS;
constexpr int kSlabRows = 64;

struct Batch {
  tp_sn_weight w[kMaxW];
  int n;
  int blk_a[kMaxW + 1];   // prefix sums of workgroups per weight for the kernel being launched
};

__device__ __forceinline__ int find_weight(const Batch& b, int blk, int& local) {
  int i = 0;
  while (i + 1 < b.n && blk >= b.blk_a[i + 1]) ++i;
  local = blk - b.blk_a[i];
  return i;
}

// sum over the 256 threads of the workgroup, the same value (same order of additions) in every thread: a butterfly inside each wavefront,
// then the four wavefront sums in wavefront order -- 2 barriers (rounds 1-4: an LDS tree with 9; the six sums at the top of every
// sn_scale_sets workgroup were a third of that launch)
__device__ __forceinline__ float block_sum(float v, float* red) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  const float r = ((red[0] + red[1]) + red[2]) + red[3];
  __syncthreads();
  return r;
}

// A: t_part[slab][c] = sum_{r in slab} W[r][c] u[r]
// `set` > 0 (tp_sn_fwd_sets): u is not read from the module's buffer but formed from the PREVIOUS set's s = W v exactly as kernel C forms
// it (same thread-to-row mapping, same reduction tree: the same bits) -- so the normalisation launches of all sets can wait until the end.
__global__ __launch_bounds__(256) void sn_wtu_kernel(Batch b, int set) {
  __shared__ float us[kSlabRows];
  __shared__ float red[256];
  int local;
  const tp_sn_weight& w = b.w[find_weight(b, blockIdx.x, local)];
  const int cblocks = (w.cols + 255) / 256;
  const int slab = local / cblocks, cb = local - slab * cblocks;
  const int r0 = slab * kSlabRows, nr = min(kSlabRows, w.rows - r0);
  if (set > 0) {
    const float* sp = w.work + (int64_t)TP_SN_MAX_SLABS * w.cols + (int64_t)(set - 1) * w.rows;
    const int r_a = threadIdx.x, r_b = threadIdx.x + 256;
    const float s_a = r_a < w.rows ? sp[r_a] : 0.0f, s_b = r_b < w.rows ? sp[r_b] : 0.0f;
    const float nrm = fmaxf(sqrtf(block_sum(fmaf(s_b, s_b, fmaf(s_a, s_a, 0.0f)), red)), 1e-12f);
    if (threadIdx.x < nr) us[threadIdx.x] = sp[r0 + threadIdx.x] / nrm;
  } else if (threadIdx.x < nr) {
    us[threadIdx.x] = w.u[r0 + threadIdx.x];
  }
  __syncthreads();
  const int c = cb * 256 + threadIdx.x;
  if (c >= w.cols) return;
  float acc = 0.0f;
#pragma unroll 32
  for (int r = 0; r < nr; ++r) acc = fmaf(w.weight[(int64_t)(r0 + r) * w.cols + c], us[r], acc);   // 32 loads in flight (8: 4 more round trips)
  w.work[(int64_t)slab * w.cols + c] = acc;
}

constexpr int kWvRows = 4;                     // rows of s per workgroup of kernel B (one per wave)
using f32x4 = __attribute__((ext_vector_type(4))) float;

// B: v = normalize(sum_slabs t_part) (training; else the module's v) in LDS, then s[r] = sum_c W[r][c] v[c] for 4 rows (a wave per row).
// 16-byte loads with 16-32 of them in flight per thread: a first version with 4-byte loads, eight in flight, paid one L2 latency per
// batch, 16 + 16 batches per workgroup of the 512 x 4096 weight (29-36 us per launch; now 2 + 2 batches).
__global__ __launch_bounds__(256) void sn_wv_kernel(Batch b, int training, int set) {
  extern __shared__ __attribute__((aligned(16))) float vs[];      // [cols]
  __shared__ float red[256];
  int local;
  const tp_sn_weight& w = b.w[find_weight(b, blockIdx.x, local)];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const bool vec = (w.cols & 3) == 0 && (((uintptr_t)w.work | (uintptr_t)w.weight | (uintptr_t)w.v) & 15) == 0;
  if (training) {
    const int slabs = (w.rows + kSlabRows - 1) / kSlabRows;
    float ss = 0.0f;
    if (vec) {
      const int c4n = w.cols >> 2;
#pragma unroll 4
      for (int c4 = t; c4 < c4n; c4 += 256) {
        f32x4 part[TP_SN_MAX_SLABS];
#pragma unroll
        for (int sl = 0; sl < TP_SN_MAX_SLABS; ++sl)
          part[sl] = sl < slabs ? reinterpret_cast<const f32x4*>(w.work + (int64_t)sl * w.cols)[c4] : f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 tt = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int sl = 0; sl < TP_SN_MAX_SLABS; ++sl) tt += part[sl];            // slab-ascending (absent slabs add 0)
        reinterpret_cast<f32x4*>(vs)[c4] = tt;
        ss = fmaf(tt[3], tt[3], fmaf(tt[2], tt[2], fmaf(tt[1], tt[1], fmaf(tt[0], tt[0], ss))));
      }
    } else {
      for (int c = t; c < w.cols; c += 256) {
        float tt = 0.0f;
        for (int sl = 0; sl < slabs; ++sl) tt += w.work[(int64_t)sl * w.cols + c];
        vs[c] = tt;
        ss = fmaf(tt, tt, ss);
      }
    }
    const float nrm = fmaxf(sqrtf(block_sum(ss, red)), 1e-12f);
    for (int c = t; c < w.cols; c += 256) {
      const float v = vs[c] / nrm;
      vs[c] = v;
      if (local == 0) {                        // (one workgroup per weight updates the module's buffer and this forward's copy)
        w.v[c] = v;
        if (w.v_out) w.v_out[c] = v;
      }
    }
  } else {
    for (int c = t; c < w.cols; c += 256) {
      const float v = w.v[c];
      vs[c] = v;
      if (local == 0 && w.v_out) w.v_out[c] = v;
    }
  }
  __syncthreads();
  float* s = w.work + (int64_t)TP_SN_MAX_SLABS * w.cols + (int64_t)set * w.rows;
  const int r = local * kWvRows + wave;
  if (r >= w.rows) return;
  const float* row = w.weight + (int64_t)r * w.cols;
  float acc = 0.0f;
  if (vec) {
    const f32x4* row4 = reinterpret_cast<const f32x4*>(row);
    const f32x4* v4 = reinterpret_cast<const f32x4*>(vs);
#pragma unroll 16
    for (int c4 = lane; c4 < (w.cols >> 2); c4 += 64) {                        // 16 x 16-byte loads in flight
      const f32x4 a = row4[c4], v = v4[c4];
      acc = fmaf(a[3], v[3], fmaf(a[2], v[2], fmaf(a[1], v[1], fmaf(a[0], v[0], acc))));
    }
  } else {
#pragma unroll 8
    for (int c = lane; c < w.cols; c += 64) acc = fmaf(row[c], vs[c], acc);
  }
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if (lane == 0) s[r] = acc;
}

// C: u = normalize(s) (training; else the module's u), sigma = u . s, W_sn = W / sigma   (rows <= 512: two values of s per thread)
constexpr int kScaleElems = 4096;              // elements of W per workgroup
__global__ __launch_bounds__(256) void sn_scale_kernel(Batch b, int training) {
  __shared__ float red[256];
  int local;
  const tp_sn_weight& w = b.w[find_weight(b, blockIdx.x, local)];
  const float* s = w.work + (int64_t)TP_SN_MAX_SLABS * w.cols;
  const int r_a = threadIdx.x, r_b = threadIdx.x + 256;
  const float s_a = r_a < w.rows ? s[r_a] : 0.0f, s_b = r_b < w.rows ? s[r_b] : 0.0f;
  float u_a, u_b;
  if (training) {
    const float ss = fmaf(s_b, s_b, fmaf(s_a, s_a, 0.0f));
    const float nrm = fmaxf(sqrtf(block_sum(ss, red)), 1e-12f);
    u_a = s_a / nrm; u_b = s_b / nrm;
  } else {
    u_a = r_a < w.rows ? w.u[r_a] : 0.0f; u_b = r_b < w.rows ? w.u[r_b] : 0.0f;
  }
  const float sg = block_sum(fmaf(u_b, s_b, fmaf(u_a, s_a, 0.0f)), red);
  if (local == 0) {                            // one workgroup per weight: the module's u, sigma, this forward's copy of u
    if (training) {
      if (r_a < w.rows) w.u[r_a] = u_a;
      if (r_b < w.rows) w.u[r_b] = u_b;
    }
    if (w.u_out) {
      if (r_a < w.rows) w.u_out[r_a] = u_a;
      if (r_b < w.rows) w.u_out[r_b] = u_b;
    }
    if (threadIdx.x == 0) *w.sigma = sg;
  }
  const int64_t n = (int64_t)w.rows * w.cols;
  for (int64_t i = (int64_t)local * kScaleElems + threadIdx.x; i < min(n, (int64_t)(local + 1) * kScaleElems); i += 256)
    w.weight_sn[i] = w.weight[i] / sg;
}

// C for SEVERAL sets in one launch (tp_sn_fwd_sets): u_k = normalize(s_k), sigma_k = u_k . s_k for every set k, the module's u from the
// last one, W_sn_k = W / sigma_k for every set from ONE read of W.
struct SetOuts { float* weight_sn[TP_SN_MAX_SETS][kMaxW]; float* sigma[TP_SN_MAX_SETS][kMaxW]; float* u_out[TP_SN_MAX_SETS][kMaxW]; };
__global__ __launch_bounds__(256) void sn_scale_sets_kernel(Batch b, SetOuts o, int n_sets) {
  __shared__ float red[256];
  int local;
  const int wi = find_weight(b, blockIdx.x, local);
  const tp_sn_weight& w = b.w[wi];
  const int r_a = threadIdx.x, r_b = threadIdx.x + 256;
  float sg[TP_SN_MAX_SETS];
#pragma unroll
  for (int k = 0; k < TP_SN_MAX_SETS; ++k) {
    sg[k] = 1.0f;
    if (k < n_sets) {
      const float* s = w.work + (int64_t)TP_SN_MAX_SLABS * w.cols + (int64_t)k * w.rows;
      const float s_a = r_a < w.rows ? s[r_a] : 0.0f, s_b = r_b < w.rows ? s[r_b] : 0.0f;
      const float nrm = fmaxf(sqrtf(block_sum(fmaf(s_b, s_b, fmaf(s_a, s_a, 0.0f)), red)), 1e-12f);
      const float u_a = s_a / nrm, u_b = s_b / nrm;
      sg[k] = block_sum(fmaf(u_b, s_b, fmaf(u_a, s_a, 0.0f)), red);
      if (local == 0) {
        if (k == n_sets - 1) {
          if (r_a < w.rows) w.u[r_a] = u_a;
          if (r_b < w.rows) w.u[r_b] = u_b;
        }
        if (o.u_out[k][wi]) {
          if (r_a < w.rows) o.u_out[k][wi][r_a] = u_a;
          if (r_b < w.rows) o.u_out[k][wi][r_b] = u_b;
        }
        if (threadIdx.x == 0) *o.sigma[k][wi] = sg[k];
      }
    }
  }
  const int64_t n = (int64_t)w.rows * w.cols;
  const int64_t i0 = (int64_t)local * kScaleElems, i1 = min(n, i0 + kScaleElems);
  bool vec = (i1 - i0) == kScaleElems && ((uintptr_t)w.weight & 15) == 0;
#pragma unroll
  for (int k = 0; k < TP_SN_MAX_SETS; ++k)
    if (k < n_sets) vec = vec && ((uintptr_t)o.weight_sn[k][wi] & 15) == 0;
  if (vec) {                                       // 16-byte accesses, the block's four loads in flight together
    f32x4 wv[kScaleElems / 1024];
#pragma unroll
    for (int j = 0; j < kScaleElems / 1024; ++j) wv[j] = reinterpret_cast<const f32x4*>(w.weight + i0)[j * 256 + threadIdx.x];
#pragma unroll
    for (int k = 0; k < TP_SN_MAX_SETS; ++k)
      if (k < n_sets)
#pragma unroll
        for (int j = 0; j < kScaleElems / 1024; ++j) {
          f32x4 q;
#pragma unroll
          for (int e = 0; e < 4; ++e) q[e] = wv[j][e] / sg[k];
          reinterpret_cast<f32x4*>(o.weight_sn[k][wi] + i0)[j * 256 + threadIdx.x] = q;
        }
    return;
  }
  for (int64_t i = i0 + threadIdx.x; i < i1; i += 256) {
    const float wv = w.weight[i];
#pragma unroll
    for (int k = 0; k < TP_SN_MAX_SETS; ++k)
      if (k < n_sets) o.weight_sn[k][wi][i] = wv / sg[k];
  }
}

// what tp_sn_bwd_step adds to the two launches (all zero for tp_sn_bwd)
struct StepTail {
  const float* t[4]; float w[4]; int n;            // D: loss total = sum_k t[k][0] * w[k] ...
  float* total; int* bad; int* snapshot; int n_bad, word_finite;    // ... its finiteness into the sticky word, the words' snapshot
};
struct RmsTail {
  float* p[kMaxW]; float* sq[kMaxW]; float* step[kMaxW];            // E: RMSprop on the element just formed (on == 1)
  const float* lr_dev; float lr_host, alpha, one_minus_alpha, eps;
  const int* gate; int n_gate, on, elems;
};

// D: per-workgroup partial of <G, W_sn> (and of the second instance's <G2, W_sn2>)
__global__ __launch_bounds__(256) void sn_dot_kernel(Batch b, StepTail st) {
  __shared__ float red[256];
  int local;
  if (blockIdx.x == 0 && threadIdx.x == 0 && st.n > 0) {          // (tp_weighted_sum_flags's arithmetic and order of side effects)
    float acc = 0.f;
    for (int k = 0; k < st.n; ++k) acc += st.t[k][0] * st.w[k];
    st.total[0] = acc;
    if (!(acc - acc == 0.f)) st.bad[st.word_finite] |= 1;
    for (int k = 0; k < st.n_bad; ++k) st.snapshot[k] = st.bad[k];
  }
  const tp_sn_weight& w = b.w[find_weight(b, blockIdx.x, local)];
  const int64_t n = (int64_t)w.rows * w.cols;
  const int parts = (int)((n + 4095) / 4096);
  float acc = 0.0f, acc2 = 0.0f;
#pragma unroll 8
  for (int64_t i = (int64_t)local * 4096 + threadIdx.x; i < min(n, (int64_t)(local + 1) * 4096); i += 256) {
    acc = fmaf(w.grad_sn[i], w.weight_sn[i], acc);
    if (w.grad_sn2) acc2 = fmaf(w.grad_sn2[i], w.weight_sn2[i], acc2);
  }
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) w.work[local] = acc;
  if (w.grad_sn2) {
    acc2 = block_sum(acc2, red);
    if (threadIdx.x == 0) w.work[parts + local] = acc2;
  }
}

// E: dW = (G - <G, W_sn> u v^T) / sigma  (+ the same of the second instance)
constexpr int kGradElems = 1024;      // (TP_SN_GRAD_ELEMS: 2048 the same, 4096 / 8192 slower -- D2b alone 209 / 215 us, profiles/r5)
__global__ __launch_bounds__(256) void sn_grad_kernel(Batch b, RmsTail rt) {
  __shared__ float red[256];
  int local;
  const int wi = find_weight(b, blockIdx.x, local);
  const tp_sn_weight& w = b.w[wi];
  bool apply = rt.on != 0;
  for (int k = 0; k < rt.n_gate; ++k) apply = apply && rt.gate[k] == 0;      // a flagged step: parameters and statistics stay as they are
  const float lr = !apply ? 0.f : rt.lr_dev != nullptr ? *rt.lr_dev : rt.lr_host;
  if (apply && local == 0 && threadIdx.x == 0 && rt.step[wi] != nullptr) rt.step[wi][0] += 1.0f;
  const int64_t n = (int64_t)w.rows * w.cols;
  const int parts = (int)((n + 4095) / 4096);
  float d = 0.0f, d2 = 0.0f;
  for (int i = threadIdx.x; i < parts; i += 256) { d += w.work[i]; if (w.grad_sn2) d2 += w.work[parts + i]; }
  d = block_sum(d, red);
  if (w.grad_sn2) d2 = block_sum(d2, red);
  const float sg = *w.sigma, sg2 = w.grad_sn2 ? *w.sigma2 : 1.0f;
  // four elements per thread and round: ALL their operands are requested before the first store (the pointers may alias as far as the
  // compiler knows, so a plain loop waits out a memory latency per element -- the Adam kernel's lesson, csrc/train_misc.hip)
  constexpr int kU = 4;
  const int64_t i_end = min(n, (int64_t)(local + 1) * rt.elems);
  for (int64_t i0 = (int64_t)local * rt.elems + threadIdx.x; i0 < i_end; i0 += kU * 256) {
    float g1[kU], g2[kU], uv1[kU], uv2[kU], old[kU], sq0[kU], p0[kU];
    bool live[kU];
#pragma unroll
    for (int q = 0; q < kU; ++q) {
      const int64_t i = min(i0 + q * 256, i_end - 1);
      live[q] = i0 + q * 256 < i_end;
      const int r = (int)(i / w.cols), c = (int)(i - (int64_t)r * w.cols);
      g1[q] = w.grad_sn[i];
      uv1[q] = d * w.u[r] * w.v[c];                 // ((d u) v, the order of the plain form)
      g2[q] = w.grad_sn2 ? w.grad_sn2[i] : 0.f;
      uv2[q] = w.grad_sn2 ? d2 * w.u2[r] * w.v2[c] : 0.f;
      old[q] = w.accumulate ? w.grad[i] : 0.f;
      sq0[q] = apply ? rt.sq[wi][i] : 0.f;
      p0[q] = apply ? rt.p[wi][i] : 0.f;
    }
#pragma unroll
    for (int q = 0; q < kU; ++q) {
      if (!live[q]) continue;
      const int64_t i = i0 + q * 256;
      float gv = (g1[q] - uv1[q]) / sg;
      if (w.grad_sn2) gv = gv + (g2[q] - uv2[q]) / sg2;
      gv = w.accumulate ? old[q] + gv : gv;
      w.grad[i] = gv;
      if (apply) {                                 // csrc/rmsprop.hip's arithmetic, torch's order
        float sq = sq0[q];
        sq = __fadd_rn(__fmul_rn(sq, rt.alpha), __fmul_rn(__fmul_rn(rt.one_minus_alpha, gv), gv));
        rt.sq[wi][i] = sq;
        const float avg = __fadd_rn(sqrtf(sq), rt.eps);
        rt.p[wi][i] = __fadd_rn(p0[q], __fmul_rn(-lr, __fdiv_rn(gv, avg)));
      }
    }
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
    if (bwd && w.grad_sn2 && (!w.weight_sn2 || !w.u2 || !w.v2 || !w.sigma2)) { tp::set_error("%s: incomplete second instance", what); return -1; }
  }
  return 0;
}
}  // namespace

extern "C" int64_t tp_sn_work_floats(int rows, int cols) {
  const int64_t fwd = (int64_t)TP_SN_MAX_SLABS * cols + (int64_t)TP_SN_MAX_SETS * rows, bwd = 2 * (((int64_t)rows * cols + 4095) / 4096);
  return fwd > bwd ? fwd : bwd;
}

extern "C" int tp_sn_fwd(const tp_sn_weight* ws, int n, int training, tp_stream_t stream) {
  if (int rc = check(ws, n, false, "tp_sn_fwd")) return rc;
  hipStream_t st = (hipStream_t)stream;
  Batch b;
  int max_cols = 0;
  for (int i = 0; i < n; ++i) max_cols = ws[i].cols > max_cols ? ws[i].cols : max_cols;
  TP_REQUIRE((size_t)max_cols * sizeof(float) <= 64 * 1024, "at most 16384 columns");
  if (training) {
    int g = fill(b, ws, n, [](const tp_sn_weight& w) { return ((w.cols + 255) / 256) * ((w.rows + kSlabRows - 1) / kSlabRows); });
    hipLaunchKernelGGL(sn_wtu_kernel, dim3(g), dim3(256), 0, st, b, 0);
  }
  int g = fill(b, ws, n, [](const tp_sn_weight& w) { return (w.rows + kWvRows - 1) / kWvRows; });
  static unsigned long long flags = 0;
  if ((size_t)max_cols * sizeof(float) > 32 * 1024 && tp::first_use_on_device(flags))
    TP_REQUIRE(hipFuncSetAttribute(reinterpret_cast<const void*>(sn_wv_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024) == hipSuccess,
               "cannot raise the LDS limit");
  hipLaunchKernelGGL(sn_wv_kernel, dim3(g), dim3(256), (size_t)max_cols * sizeof(float), st, b, training, 0);
  g = fill(b, ws, n, [](const tp_sn_weight& w) { return (int)(((int64_t)w.rows * w.cols + kScaleElems - 1) / kScaleElems); });
  hipLaunchKernelGGL(sn_scale_kernel, dim3(g), dim3(256), 0, st, b, training);
  return tp::check_launch("tp_sn_fwd");
}

// n_sets consecutive training-mode power iterations of the same n weights (what n_sets forwards in a row would run: the discriminator
// step's three per iteration), entry k * n + i = weight i of set k (its weight_sn / sigma / u_out / v_out; weight, u, v, work shared):
// 2 launches per set + ONE normalisation launch for all sets instead of 3 per set.  Bit-identical to n_sets tp_sn_fwd calls.
extern "C" int tp_sn_fwd_sets(const tp_sn_weight* ws, int n, int n_sets, tp_stream_t stream) {
  TP_REQUIRE(n_sets >= 1 && n_sets <= TP_SN_MAX_SETS, "1..TP_SN_MAX_SETS sets");
  for (int k = 0; k < n_sets; ++k)
    if (int rc = check(ws + (size_t)k * n, n, false, "tp_sn_fwd_sets")) return rc;
  hipStream_t st = (hipStream_t)stream;
  int max_cols = 0;
  for (int i = 0; i < n; ++i) {
    max_cols = ws[i].cols > max_cols ? ws[i].cols : max_cols;
    for (int k = 1; k < n_sets; ++k) {
      const tp_sn_weight& a = ws[i], & c = ws[(size_t)k * n + i];
      TP_REQUIRE(a.weight == c.weight && a.u == c.u && a.v == c.v && a.work == c.work && a.rows == c.rows && a.cols == c.cols,
                 "the sets of a weight share weight / u / v / work");
      TP_REQUIRE(a.weight_sn != c.weight_sn && a.sigma != c.sigma, "every set needs its own weight_sn / sigma");
    }
  }
  TP_REQUIRE((size_t)max_cols * sizeof(float) <= 64 * 1024, "at most 16384 columns");
  static unsigned long long flags = 0;
  if ((size_t)max_cols * sizeof(float) > 32 * 1024 && tp::first_use_on_device(flags))
    TP_REQUIRE(hipFuncSetAttribute(reinterpret_cast<const void*>(sn_wv_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024) == hipSuccess,
               "cannot raise the LDS limit");
  Batch b;
  SetOuts o{};
  for (int k = 0; k < n_sets; ++k) {
    const tp_sn_weight* wk = ws + (size_t)k * n;
    int g = fill(b, wk, n, [](const tp_sn_weight& w) { return ((w.cols + 255) / 256) * ((w.rows + kSlabRows - 1) / kSlabRows); });
    hipLaunchKernelGGL(sn_wtu_kernel, dim3(g), dim3(256), 0, st, b, k);
    g = fill(b, wk, n, [](const tp_sn_weight& w) { return (w.rows + kWvRows - 1) / kWvRows; });
    hipLaunchKernelGGL(sn_wv_kernel, dim3(g), dim3(256), (size_t)max_cols * sizeof(float), st, b, 1, k);
    for (int i = 0; i < n; ++i) { o.weight_sn[k][i] = wk[i].weight_sn; o.sigma[k][i] = wk[i].sigma; o.u_out[k][i] = wk[i].u_out; }
  }
  const int g = fill(b, ws, n, [](const tp_sn_weight& w) { return (int)(((int64_t)w.rows * w.cols + kScaleElems - 1) / kScaleElems); });
  hipLaunchKernelGGL(sn_scale_sets_kernel, dim3(g), dim3(256), 0, st, b, o, n_sets);
  return tp::check_launch("tp_sn_fwd_sets");
}

namespace {
int sn_bwd_launch(const tp_sn_weight* ws, int n, const StepTail& tl, const RmsTail& rt, hipStream_t st, const char* what) {
  Batch b;
  int g = fill(b, ws, n, [](const tp_sn_weight& w) { return (int)(((int64_t)w.rows * w.cols + 4095) / 4096); });
  hipLaunchKernelGGL(sn_dot_kernel, dim3(g), dim3(256), 0, st, b, tl);
  static const int elems = [] { const char* e = getenv("TP_SN_GRAD_ELEMS"); const int v = e ? atoi(e) : 0; return v >= 256 ? v : kGradElems; }();
  RmsTail r2 = rt;
  r2.elems = elems;
  b.n = n; b.blk_a[0] = 0;
  for (int i = 0; i < n; ++i) { b.w[i] = ws[i]; b.blk_a[i + 1] = b.blk_a[i] + (int)(((int64_t)ws[i].rows * ws[i].cols + elems - 1) / elems); }
  g = b.blk_a[n];
  hipLaunchKernelGGL(sn_grad_kernel, dim3(g), dim3(256), 0, st, b, r2);
  return tp::check_launch(what);
}
}  // namespace

extern "C" int tp_sn_bwd(const tp_sn_weight* ws, int n, tp_stream_t stream) {
  if (int rc = check(ws, n, true, "tp_sn_bwd")) return rc;
  return sn_bwd_launch(ws, n, StepTail{}, RmsTail{}, (hipStream_t)stream, "tp_sn_bwd");
}

extern "C" int tp_sn_bwd_step(const tp_sn_weight* ws, int n, const tp_sn_step_tail* a, tp_stream_t stream) {
  if (int rc = check(ws, n, true, "tp_sn_bwd_step")) return rc;
  TP_REQUIRE(a != nullptr && a->n_terms >= 1 && a->n_terms <= 4 && a->total && a->bad && a->snapshot && a->n_bad > 0 && a->n_bad <= 64
             && a->word_finite >= 0 && a->word_finite < a->n_bad, "tp_sn_bwd_step: bad loss-total / gate arguments");
  StepTail tl{};
  RmsTail rt{};
  for (int k = 0; k < a->n_terms; ++k) { TP_REQUIRE(a->terms[k] != nullptr, "tp_sn_bwd_step: null term"); tl.t[k] = a->terms[k]; tl.w[k] = a->weights[k]; }
  tl.n = a->n_terms; tl.total = a->total; tl.bad = a->bad; tl.snapshot = a->snapshot; tl.n_bad = a->n_bad; tl.word_finite = a->word_finite;
  for (int i = 0; i < n; ++i) {
    TP_REQUIRE(a->param[i] && a->square_avg[i] && !ws[i].accumulate, "tp_sn_bwd_step: parameter / square_avg of every weight, no accumulation");
    for (int j = 0; j < i; ++j) TP_REQUIRE(a->param[j] != a->param[i] && (a->step[j] == nullptr || a->step[j] != a->step[i]), "tp_sn_bwd_step: repeated tensor");
    rt.p[i] = a->param[i]; rt.sq[i] = a->square_avg[i]; rt.step[i] = a->step[i];
  }
  rt.lr_dev = a->lr_dev; rt.lr_host = a->lr_host; rt.alpha = a->alpha; rt.one_minus_alpha = a->one_minus_alpha; rt.eps = a->eps;
  rt.gate = a->snapshot; rt.n_gate = a->n_bad; rt.on = 1;
  return sn_bwd_launch(ws, n, tl, rt, (hipStream_t)stream, "tp_sn_bwd_step");
}
