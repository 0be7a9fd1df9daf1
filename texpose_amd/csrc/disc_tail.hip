// K17: the TAIL of the PatchGAN -- the full-map convolution that ends the ladder (K15: a [M, K] x [K, N] product, K = 8192, N = 64,
// a handful of rows) AND the scale-conditioned head (K14) -- as ONE launch per derivative order.
// reference layers/discriminator.py:30-40,110-115: z = conv(a, W0) over the whole 4x4 map;  t0 = lrelu([z, enc(s), s]);
// t1 = lrelu(W1 t0);  t2 = lrelu(W2 t1);  out = W3 t2.
//
// Why: in the replayed B=4 iteration the tail was 2 launches forward (skinny_fwd 11-14 us on 64 workgroups + head_fwd 15-20 us on
// ONE workgroup), 3 backward (head_bwd 9-40 us on one workgroup -- its weight-gradient loops re-read their operands from global
// memory --, a rocBLAS data gradient, skinny_wgrad) and 2 in the R1 double backward (skinny_fwd + head_bwd_bwd 30 us): 131 us of
// single-workgroup head kernels and 16 launches on the discriminator step's serial chain (profiles/r4).
//   F  (tp_disc_tail_fwd):  the K range of z is split over S workgroups per column; the LAST workgroup to arrive (one ticket,
//       the gfx950 hand-over of patch_conv.hip reduce_tiles) adds the partial sums in slice order and runs the head on the
//       weights every workgroup prefetched into LDS while its products were in flight.
//       MODE_R1 (tp_disc_tail_bwd_bwd): the same launch shape for the R1 penalty's second pass: z = c W0^T is the cotangent of the
//       first pass' gz, the last workgroup runs the head's double backward (formulas: csrc/disc_head.hip).
//   B  (tp_disc_tail_bwd):  every workgroup owns 64 columns of K, recomputes the head's backward (a few 10^4 MACs on weights in
//       LDS) and produces its slice of BOTH the data gradient c_a = gz W0 and the weight gradient gW0 = sum_rows gz (x) a; rows of
//       a second (cotangent, input) pair -- the R1 path's, texpose_amd/disc_step.py -- join the same sum.  Workgroup 0 also writes
//       gz / e1 / e2 and the head's weight gradients, from LDS.
// Plain fp32 FMAs, fixed summation orders, no float atomics.
#include "tp_common.h"

#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "the last-workgroup hand-over below relies on the gfx950 sc1 write-through contract (csrc/patch_conv.hip reduce_tiles)"
#endif

namespace {
constexpr int kT = 256;
constexpr int kMaxM = TP_DISC_TAIL_MAX_ROWS;       // rows of one pass
using f32x4 = __attribute__((ext_vector_type(4))) float;

struct TailP {
  // ladder side
  const float* a;        // F: [M,K] ladder output (MODE_R1: the cotangent c [M,K]);  B: [M,K] ladder output (weight gradient), may be NULL
  const float* W0;       // [N,K]
  // head
  const float* scale;    // [M]        F
  const float* W1; const float* W2; const float* W3;   // [H,Cin], [H,H], [H]
  const float* g;        // [M]        B, R1: cotangent of out
  float* t0; float* t1; float* t2;     // [M,Cin], [M,H], [M,H]   F: written;  B, R1: read
  float* e1; float* e2;                // [M,H]                   B: written (optional);  R1: read
  float* out;            // F: out [M];  R1: d/d g [M] (optional)
  float* gz;             // B: [M,N] (optional)
  float* c_a;            // B: [M,K] data gradient (optional)
  int c_a_store;         // B: c_a is written (0: only c_z is wanted; c_a != NULL still selects the data-gradient code)
  float* gW0;            // B: [N,K] (optional)
  const float* gy2;      // B: [M2,N] extra cotangent rows of the weight gradient (optional)
  const float* a2;       // B: [M2,K] their inputs
  float* gW1; float* gW2; float* gW3;  // B, R1: head weight gradients (optional in B)
  // B, optional: the InstanceNorm + LeakyReLU backward of the ladder's last stage applied to c_a before it leaves the workgroup
  // (K9 tp_inorm_lrelu_bwd: c_z = rstd P(c_a * s(xhat)) [+ addend]; an instance = in_P consecutive columns, in_P | 64)
  const float* in_xhat;  // [M,K]
  const float* in_rstd;  // [M * K / in_P]
  const float* in_addend;// [M,K] or NULL
  float* c_z;            // [M,K]
  int in_P;
  float* ws;             // F, R1: partial sums [N][S][M]
  unsigned* ticket;      // F, R1: one zero word
  int M, M2, K, N, C, L, H, Cin, S;
  float slope;
  int accumulate;        // B: gW1..3 += (the R1 pass wrote its share there first)
};

__device__ __forceinline__ int imax_dev(int a, int b) { return a > b ? a : b; }
__device__ __forceinline__ float lrelu(float v, float s) { return v > 0.f ? v : v * s; }
__device__ __forceinline__ float dl(float t, float s) { return t > 0.f ? 1.f : s; }

extern __shared__ __attribute__((aligned(16))) float smem[];

// diagnostic build (make tail_timing, tools/tail_bench.py --timing): 100 MHz timestamps of one workgroup's sections
#ifdef TP_TAIL_TIMING
__device__ unsigned long long g_tail_stamps[16];
#define TSTAMP(i, cond) do { __syncthreads(); if ((cond) && threadIdx.x == 0) g_tail_stamps[i] = wall_clock64(); } while (0)
#else
#define TSTAMP(i, cond) do { } while (0)
#endif

// W1 | W2 | W3 into LDS (37 KB at ndf = 64; L2 hits for all but the first workgroup): 16-byte loads, four in flight per thread.
// W2's rows are stored with stride H + 1: the forward reads W2[o][j] with lanes = o, and a row stride of H = 64 floats puts all 64
// lanes on one LDS bank (measured: the 64 x 64 product took 2.6 us against 1.2 us for the 64 x 73 one, whose stride is odd).
__device__ __forceinline__ void stage_head_weights(const TailP& p, float* w1, float* w2, float* w3) {
  const int n1 = p.H * p.Cin, n2 = p.H * p.H, t = threadIdx.x, ld2 = p.H + 1;
  const bool vec = (((uintptr_t)p.W1 | (uintptr_t)p.W2) & 15) == 0 && (n1 & 3) == 0 && (p.H & 3) == 0;
  if (vec) {
    const f32x4* s1 = reinterpret_cast<const f32x4*>(p.W1);
    const f32x4* s2 = reinterpret_cast<const f32x4*>(p.W2);
    f32x4* d1 = reinterpret_cast<f32x4*>(w1);
#pragma unroll 4
    for (int i = t; i < n1 / 4; i += kT) d1[i] = s1[i];
#pragma unroll 4
    for (int i = t; i < n2 / 4; i += kT) {
      const f32x4 v = s2[i];
      const int e = 4 * i, r = e / p.H, c = e - r * p.H;              // (H % 4 == 0: the four values share a row)
      float* d = w2 + r * ld2 + c;
      d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
    }
  } else {
    for (int i = t; i < n1; i += kT) w1[i] = p.W1[i];
    for (int i = t; i < n2; i += kT) w2[(i / p.H) * ld2 + i % p.H] = p.W2[i];
  }
  for (int i = t; i < p.H; i += kT) w3[i] = p.W3[i];
}

// out[m][o] = f( sum_j W[o * ldw + j] * x[m * ldx + j] ), j < n_in, o < n_out, m < M  (W row-major [n_out][ldw] in LDS, x [kMaxM][ldx] in LDS):
// wave w takes the quarter [w n_in / 4, ...) of j for every o (lanes = o: row stride ldw is odd or the rows are read along j -- no
// bank conflicts either way), MB >= M accumulators per thread (rows beyond M hold whatever the LDS rows hold: never consumed), the
// four partial sums meet in `ps` [4][kMaxM][n_out] and are added in wave order.  TRANS: W is read transposed (W[j * ldw + o]).
// Every j step issues its 1 + MB LDS reads before its MB FMAs, four steps unrolled: the loop runs at LDS issue rate, not at LDS
// latency per FMA (a first version with a run-time row count predicated every FMA: 6.5 us per 64 x 73 product; profiles/r4).
template <bool TRANS, int MB, class Fin>
__device__ __forceinline__ void matvec4(const float* W, int ldw, const float* x, int ldx, int n_in, int n_out, int M, float* ps, Fin fin) {
  const int t = threadIdx.x, w = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
  const int per = (n_in + 3) >> 2, j0 = min(n_in, w * per), j1 = min(n_in, j0 + per);
  for (int o0 = 0; o0 < n_out; o0 += 64) {
    const int o = min(o0 + lane, n_out - 1);
    float acc[MB];
#pragma unroll
    for (int m = 0; m < MB; ++m) acc[m] = 0.f;
#pragma unroll 4
    for (int j = j0; j < j1; ++j) {
      const float wv = TRANS ? W[j * ldw + o] : W[o * ldw + j];
      float xv[MB];
#pragma unroll
      for (int m = 0; m < MB; ++m) xv[m] = x[m * ldx + j];
#pragma unroll
      for (int m = 0; m < MB; ++m) acc[m] = fmaf(wv, xv[m], acc[m]);
    }
    if (o0 + lane < n_out) {
#pragma unroll
      for (int m = 0; m < MB; ++m) ps[(w * kMaxM + m) * n_out + o] = acc[m];
    }
  }
  __syncthreads();
  for (int e = t; e < M * n_out; e += kT) {
    const int m = e / n_out, o = e - m * n_out;
    fin(m, o, ((ps[(0 * kMaxM + m) * n_out + o] + ps[(1 * kMaxM + m) * n_out + o]) + ps[(2 * kMaxM + m) * n_out + o]) + ps[(3 * kMaxM + m) * n_out + o]);
  }
  __syncthreads();
}

// out[i][j] (+)= sum_m u[m * ldu + i] * v[m * ldv + j] for the elements e = i * cols + j in [e0, e1) (u, v in LDS; rows ascending;
// rows M .. MB-1 of u / v are ZERO or finite garbage times zero: the callers zero-fill u beyond M)
template <int MB>
__device__ __forceinline__ void outer_rows(float* out, const float* u, int ldu, const float* v, int ldv, int cols, int e0, int e1, bool add) {
  for (int e = e0 + (int)threadIdx.x; e < e1; e += kT) {
    const int i = e / cols, j = e - i * cols;
    const float old = add ? out[e] : 0.f;
    float uu[MB], vv[MB];
#pragma unroll
    for (int m = 0; m < MB; ++m) { uu[m] = u[m * ldu + i]; vv[m] = v[m * ldv + j]; }
    float acc = 0.f;
#pragma unroll
    for (int m = 0; m < MB; ++m) acc = fmaf(uu[m], vv[m], acc);
    out[e] = old + acc;
  }
}

// ---------------------------------------------------------------------------------------------------------------- F / R1
// MB: compile-time row bucket (4 / 8 / 16 >= M): the row loops are unrolled without predicates; rows M .. MB-1 of the LDS row arrays
// are zero (filled at kernel start) or never consumed.
// (pa, pb, na: two independent problems in one launch, workgroups [0, na) on the first -- see csrc/patch_conv.hip conv4s2_fwd_in_kernel)
template <bool R1, int MB>
__global__ __launch_bounds__(kT) void disc_tail_fwd_kernel(TailP pa, TailP pb, int na) {
  const bool second = na >= 0 && (int)blockIdx.x >= na;
  const TailP& p = second ? pb : pa;
  const int bid = (int)blockIdx.x - (second ? na : 0), vgrid = second ? (int)gridDim.x - na : (na >= 0 ? na : (int)gridDim.x);
  float* w1 = smem; float* w2 = w1 + p.H * p.Cin; float* w3 = w2 + p.H * (p.H + 1);
  float* a0 = w3 + p.H; float* a1 = a0 + kMaxM * p.Cin; float* a2 = a1 + kMaxM * p.H;
  float* ps = a2 + kMaxM * p.H;                    // [4][kMaxM][max(H, N)] partial sums; R1: then e1 | e2 [kMaxM][H] each, g [kMaxM]
  __shared__ int last;
  __shared__ float wsum[4][kMaxM];
  const int t = threadIdx.x, n = bid / p.S, s = bid % p.S, M = p.M;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
  TSTAMP(0, bid == 0);
  for (int e = t; e < kMaxM * (p.Cin + 2 * p.H); e += kT) a0[e] = 0.f;                     // a0 | a1 | a2: rows >= M stay zero
  if (R1)
    for (int e = t; e < 2 * kMaxM * p.H + kMaxM; e += kT) (ps + 4 * kMaxM * imax_dev(p.H, p.N))[e] = 0.f;      // e1 | e2 | g likewise
  stage_head_weights(p, w1, w2, w3);               // (every workgroup: whichever arrives last has them; issued first, consumed last)
  TSTAMP(1, bid == 0);
  // ---- this workgroup's slice of z[:, n]: k in [k0, k1), 16-byte loads, all rows' operands in flight together
  const int per = ((p.K / 4 + p.S - 1) / p.S) * 4, k0 = min(p.K, s * per), k1 = min(p.K, k0 + per);
  const float* wr = p.W0 + (size_t)n * p.K;
  float acc[MB];
#pragma unroll
  for (int r = 0; r < MB; ++r) acc[r] = 0.f;
#pragma unroll 2
  for (int k = k0 + 4 * t; k + 3 < k1; k += 4 * kT) {
    const f32x4 wv = *reinterpret_cast<const f32x4*>(wr + k);
    f32x4 xv[MB];
#pragma unroll
    for (int r = 0; r < MB; ++r) xv[r] = *reinterpret_cast<const f32x4*>(p.a + (size_t)min(r, M - 1) * p.K + k);    // (rows >= M: a valid row, unused)
#pragma unroll
    for (int r = 0; r < MB; ++r) acc[r] = fmaf(xv[r][3], wv[3], fmaf(xv[r][2], wv[2], fmaf(xv[r][1], wv[1], fmaf(xv[r][0], wv[0], acc[r]))));
  }
  TSTAMP(2, bid == 0);
  // wave butterfly (fixed order), then the four waves in wave order
#pragma unroll
  for (int r = 0; r < MB; ++r) {
    float v = acc[r];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if (lane == 0) wsum[wave][r] = v;
  }
  __syncthreads();
  // ---- hand-over (patch_conv.hip reduce_tiles: sc1 stores, the storing wave drained, ONE agent-scope add behind the barrier)
  if (t < M) {
    const float v = ((wsum[0][t] + wsum[1][t]) + wsum[2][t]) + wsum[3][t];
    __hip_atomic_store(p.ws + ((size_t)n * p.S + s) * kMaxM + t, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (t == 0) last = __hip_atomic_fetch_add(p.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(vgrid - 1);
  __syncthreads();
  TSTAMP(3, bid == 0);
  if (!last) return;
  TSTAMP(4, true);
  if (t == 0) __hip_atomic_store(p.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // ready for the next launch
  // z[m][nn] = sum of the S slices in slice order; the rest of a0
  float* e1s = ps + 4 * kMaxM * imax_dev(p.H, p.N); float* e2s = e1s + kMaxM * p.H; float* gs = e2s + kMaxM * p.H;
  if (R1) {                                        // (the first pass' intermediates: loads in flight beside the z sums)
    for (int e = t; e < M * p.H; e += kT) { e1s[e] = p.e1[e]; e2s[e] = p.e2[e]; }
    if (t < M) gs[t] = p.g[t];
  }
  for (int e = t; e < M * p.Cin; e += kT) {
    const int m = e / p.Cin, j = e - m * p.Cin;
    float v;
    if (j < p.N) {
      float z = 0.f;
      for (int s0 = 0; s0 < p.S; s0 += 8) {            // eight slices' loads in flight, added in slice order
        float part[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)
          part[u] = __hip_atomic_load(p.ws + ((size_t)j * p.S + min(s0 + u, p.S - 1)) * kMaxM + m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (s0 + u < p.S) z += part[u];
      }
      v = R1 ? dl(p.t0[e], p.slope) * z : lrelu(z, p.slope);                 // R1: a0 = d(t0) [c, 0]
    } else if (R1) {
      v = 0.f;
    } else {
      const float sc = p.scale[m];
      if (j < p.C + 2 * p.L) {
        const int l = (j - p.C) % p.L;
        const float arg = tp::mul_rn(sc, tp::mul_rn((float)(1 << l), 3.14159265358979323846f));     // s * (2^l pi rounded to fp32)
        v = tp::sincos_sel(arg, j - p.C >= p.L ? 1 : 0);
      } else v = sc;
      v = lrelu(v, p.slope);
    }
    a0[e] = v;
    if (!R1) p.t0[e] = v;
  }
  __syncthreads();
  const int kin = R1 ? p.C : p.Cin;                // (R1: the encoding / scale entries of a0 are zero)
  TSTAMP(5, true);
  matvec4<false, MB>(w1, p.Cin, a0, p.Cin, kin, p.H, M, ps, [&](int m, int o, float v) {
    const int e = m * p.H + o;
    if (!R1) { v = lrelu(v, p.slope); p.t1[e] = v; } else v *= dl(p.t1[e], p.slope);
    a1[e] = v;
  });
  TSTAMP(6, true);
  matvec4<false, MB>(w2, p.H + 1, a1, p.H, p.H, p.H, M, ps, [&](int m, int o, float v) {
    const int e = m * p.H + o;
    if (!R1) { v = lrelu(v, p.slope); p.t2[e] = v; } else v *= dl(p.t2[e], p.slope);
    a2[e] = v;
  });
  TSTAMP(7, true);
  if (p.out != nullptr)
    for (int m = wave; m < M; m += 4) {
      float v = 0.f;
      for (int j = lane; j < p.H; j += 64) v = fmaf(w3[j], a2[m * p.H + j], v);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
      if (lane == 0) p.out[m] = v;
    }
  TSTAMP(8, true);
  if (!R1) return;
  // ---- R1: weight gradients of the double backward, rows in ascending order: d/dW3 = sum g a2, d/dW2 = sum e2 (x) a1, d/dW1 = sum e1 (x) a0
  outer_rows<MB>(p.gW3, gs, 1, a2, p.H, p.H, 0, p.H, false);          // (i = 0: u[m * 1 + 0] = g[m])
  outer_rows<MB>(p.gW2, e2s, p.H, a1, p.H, p.H, 0, p.H * p.H, false);
  outer_rows<MB>(p.gW1, e1s, p.H, a0, p.Cin, p.Cin, 0, p.H * p.Cin, false);
  TSTAMP(9, true);
}

// ---------------------------------------------------------------------------------------------------------------- B
template <int MB>
__global__ __launch_bounds__(kT) void disc_tail_bwd_kernel(TailP pa, TailP pb, int na) {
  const bool second = na >= 0 && (int)blockIdx.x >= na;
  const TailP& p = second ? pb : pa;
  const int bid = (int)blockIdx.x - (second ? na : 0), vgrid = second ? (int)gridDim.x - na : (na >= 0 ? na : (int)gridDim.x);
  float* w1 = smem; float* w2 = w1 + p.H * p.Cin; float* w3 = w2 + p.H * (p.H + 1);
  float* s2 = w3 + p.H; float* s1 = s2 + kMaxM * p.H; float* gzs = s1 + kMaxM * p.H;      // e2, e1 [kMaxM,H]; gz rows [2 kMaxM][N]
  float* t0s = gzs + 2 * kMaxM * p.N; float* t1s = t0s + kMaxM * p.Cin; float* t2s = t1s + kMaxM * p.H;
  float* gsm = t2s + kMaxM * p.H;                  // g [kMaxM]
  float* ps = gsm + kMaxM;                         // [4][kMaxM][max(H, N, 64)]: matvec partial sums, then the data gradient's
  const int t = threadIdx.x, M = p.M, M2 = p.M2;
  TSTAMP(0, bid == 0);
  // rows >= M (>= M2 of the extra rows) of every row array are zero: disjoint from what is written below, no barrier in between
  for (int e = t + M * p.H; e < kMaxM * p.H; e += kT) { s2[e] = 0.f; s1[e] = 0.f; t1s[e] = 0.f; t2s[e] = 0.f; }
  for (int e = t + M * p.N; e < kMaxM * p.N; e += kT) gzs[e] = 0.f;
  for (int e = t + M2 * p.N; e < kMaxM * p.N; e += kT) gzs[kMaxM * p.N + e] = 0.f;
  for (int e = t + M * p.Cin; e < kMaxM * p.Cin; e += kT) t0s[e] = 0.f;
  if (t >= M && t < kMaxM) gsm[t] = 0.f;
  stage_head_weights(p, w1, w2, w3);
  for (int e = t; e < M * p.Cin; e += kT) t0s[e] = p.t0[e];
  for (int e = t; e < M * p.H; e += kT) { t1s[e] = p.t1[e]; t2s[e] = p.t2[e]; }
  for (int e = t; e < M2 * p.N; e += kT) gzs[kMaxM * p.N + e] = p.gy2[e];       // (the extra rows start at row kMaxM)
  if (t < M) gsm[t] = p.g[t];
  __syncthreads();
  TSTAMP(1, bid == 0);
  // ---- the head's backward (recomputed by every workgroup: ~M (H + H H + H C) MACs)
  for (int e = t; e < M * p.H; e += kT) {
    const int m = e / p.H, o = e - m * p.H;
    s2[e] = dl(t2s[e], p.slope) * (w3[o] * gsm[m]);
  }
  __syncthreads();
  matvec4<true, MB>(w2, p.H + 1, s2, p.H, p.H, p.H, M, ps, [&](int m, int j, float v) { s1[m * p.H + j] = v * dl(t1s[m * p.H + j], p.slope); });
  matvec4<true, MB>(w1, p.Cin, s1, p.H, p.H, p.N, M, ps,                                      // only the z part of e0 is anybody's gradient (N == C)
                    [&](int m, int j, float v) { gzs[m * p.N + j] = v * dl(t0s[m * p.Cin + j], p.slope); });
  TSTAMP(2, bid == 0);
  // ---- this workgroup's 64 columns of K: thread (kc, run) -- data gradient: run = quarter of the N rows of W0; weight gradient:
  // run = quarter of the N rows of gW0
  const int kc = t & 63, k = bid * 64 + kc, kk0 = min(k, p.K - 1);
  const int run = __builtin_amdgcn_readfirstlane(t >> 6);
  const int per = (p.N + 3) / 4, n0 = run * per, n1 = min(p.N, n0 + per);
  float xa[MB], xb[MB];
  if (p.gW0 != nullptr) {                          // (issued before the data gradient's loop: one latency for both)
#pragma unroll
    for (int m = 0; m < MB; ++m) {
      xa[m] = p.a[(size_t)min(m, M - 1) * p.K + kk0];
      xb[m] = M2 > 0 ? p.a2[(size_t)min(m, M2 - 1) * p.K + kk0] : 0.f;
    }
  }
  if (p.c_a != nullptr) {
    float acc[MB];
#pragma unroll
    for (int m = 0; m < MB; ++m) acc[m] = 0.f;
    const float* wc = p.W0 + kk0;
#pragma unroll 4
    for (int n = n0; n < n1; ++n) {
      const float wv = wc[(size_t)n * p.K];
      float gv[MB];
#pragma unroll
      for (int m = 0; m < MB; ++m) gv[m] = gzs[m * p.N + n];
#pragma unroll
      for (int m = 0; m < MB; ++m) acc[m] = fmaf(gv[m], wv, acc[m]);
    }
#pragma unroll
    for (int m = 0; m < MB; ++m) ps[(run * kMaxM + m) * 64 + kc] = acc[m];
    __syncthreads();
    for (int e = t; e < ((M * 64 + kT - 1) / kT) * kT; e += kT) {               // (whole waves: the lane groups below shuffle)
      const int m = min(e >> 6, M - 1), c = e & 63, kk = min(bid * 64 + c, p.K - 1);
      const bool live = (e >> 6) < M && bid * 64 + c < p.K;
      const float v = ((ps[(0 * kMaxM + m) * 64 + c] + ps[(1 * kMaxM + m) * 64 + c]) + ps[(2 * kMaxM + m) * 64 + c]) + ps[(3 * kMaxM + m) * 64 + c];
      const size_t idx = (size_t)m * p.K + kk;
      if (live && p.c_a_store) p.c_a[idx] = v;
      if (p.c_z != nullptr) {
        // K9's backward on the instance this lane group holds (in_P consecutive columns of row m): a = v s(xhat),
        // c_z = rstd (a - mean a - xhat mean(a xhat)) [+ addend]; xor-shuffles inside the group (fixed order)
        const float xh = p.in_xhat[idx];
        const float a = v * (xh > 0.f ? 1.0f : p.slope);
        float sa = a, sah = a * xh;
        for (int o = 1; o < p.in_P; o <<= 1) { sa += __shfl_xor(sa, o, 64); sah += __shfl_xor(sah, o, 64); }
        const float ma = sa / (float)p.in_P, mah = sah / (float)p.in_P;
        float z = p.in_rstd[idx / p.in_P] * (a - ma - xh * mah);
        if (p.in_addend != nullptr) z += p.in_addend[idx];
        if (live) p.c_z[idx] = z;
      }
    }
  }
  TSTAMP(3, bid == 0);
  if (p.gW0 != nullptr && k < p.K) {
#pragma unroll 2
    for (int n = n0; n < n1; ++n) {
      float ga[MB], gb[MB];
#pragma unroll
      for (int m = 0; m < MB; ++m) { ga[m] = gzs[m * p.N + n]; gb[m] = gzs[(kMaxM + m) * p.N + n]; }      // (rows >= M / M2: zero)
      float v = 0.f;
#pragma unroll
      for (int m = 0; m < MB; ++m) v = fmaf(ga[m], xa[m], v);
#pragma unroll
      for (int m = 0; m < MB; ++m) v = fmaf(gb[m], xb[m], v);
      p.gW0[(size_t)n * p.K + k] = v;
    }
  }
  TSTAMP(4, bid == 0);
  // ---- what the caller keeps of the head's backward (workgroup 0), and the head's weight gradients, spread over the workgroups
  if (bid == 0) {
    for (int e = t; e < M * p.H; e += kT) {
      if (p.e1 != nullptr) p.e1[e] = s1[e];
      if (p.e2 != nullptr) p.e2[e] = s2[e];
    }
    if (p.gz != nullptr)
      for (int e = t; e < M * p.N; e += kT) p.gz[e] = gzs[e];
  }
  if (p.gW1 == nullptr) return;
  const bool add = p.accumulate != 0;
  const int G = vgrid, b = bid;
  const int n1e = p.H * p.Cin, n2e = p.H * p.H;
  outer_rows<MB>(p.gW1, s1, p.H, t0s, p.Cin, p.Cin, (int)((int64_t)n1e * b / G), (int)((int64_t)n1e * (b + 1) / G), add);
  outer_rows<MB>(p.gW2, s2, p.H, t1s, p.H, p.H, (int)((int64_t)n2e * b / G), (int)((int64_t)n2e * (b + 1) / G), add);
  outer_rows<MB>(p.gW3, gsm, 1, t2s, p.H, p.H, (int)((int64_t)p.H * b / G), (int)((int64_t)p.H * (b + 1) / G), add);
  TSTAMP(5, bid == 0);
}

int imax3(int a, int b, int c) { return a > b ? (a > c ? a : c) : (b > c ? b : c); }
size_t lds_fwd(const TailP& p) {
  return sizeof(float) * ((size_t)p.H * p.Cin + (size_t)p.H * (p.H + 1) + p.H + 3 + (size_t)kMaxM * (p.Cin + 2 * p.H) +
                          (size_t)4 * kMaxM * imax3(p.H, p.N, 1) + (size_t)2 * kMaxM * p.H + kMaxM);
}
size_t lds_bwd(const TailP& p) {
  return sizeof(float) * ((size_t)p.H * p.Cin + (size_t)p.H * (p.H + 1) + p.H + 3 + (size_t)kMaxM * 2 * p.H + (size_t)2 * kMaxM * p.N +
                          (size_t)kMaxM * (p.Cin + 2 * p.H) + kMaxM + (size_t)4 * kMaxM * imax3(p.H, p.N, 64));
}

int fill(TailP* q, const tp_disc_tail_args* a, const char* what) {
  if (!a || a->M <= 0 || a->M > kMaxM || a->M2 < 0 || a->M2 > kMaxM || a->K <= 0 || a->K % 4 != 0 || a->N <= 0 || a->L < 0 || a->L > 24 || a->H <= 0) {
    tp::set_error("%s: bad sizes (at most %d rows, K a multiple of 4)", what, kMaxM);
    return -1;
  }
  q->M = a->M; q->M2 = a->M2; q->K = a->K; q->N = a->N; q->C = a->N; q->L = a->L; q->H = a->H; q->Cin = a->N + 2 * a->L + 1; q->slope = a->slope;
  q->a = a->a; q->W0 = a->W0; q->scale = a->scale; q->W1 = a->W1; q->W2 = a->W2; q->W3 = a->W3; q->g = a->g_out;
  q->t0 = a->t0; q->t1 = a->t1; q->t2 = a->t2; q->e1 = a->e1; q->e2 = a->e2; q->out = a->out; q->gz = a->gz; q->c_a = a->c_a;
  q->gW0 = a->gW0; q->gy2 = a->gy2; q->a2 = a->a2; q->gW1 = a->gW1; q->gW2 = a->gW2; q->gW3 = a->gW3;
  q->ws = (float*)a->workspace; q->ticket = a->ticket; q->accumulate = a->accumulate_gw;
  q->in_xhat = a->in_xhat; q->in_rstd = a->in_rstd; q->in_addend = a->in_addend; q->c_z = a->c_z; q->in_P = a->in_P; q->c_a_store = a->c_a != nullptr;
  if (q->c_z != nullptr) {
    if (!q->in_xhat || !q->in_rstd || q->in_P <= 0 || 64 % q->in_P != 0 || q->K % q->in_P != 0 || (q->K & 63) != 0) {
      tp::set_error("%s: the fused InstanceNorm backward needs xhat, rstd, in_P | 64 and K a multiple of 64", what);
      return -1;
    }
    if (q->c_a == nullptr) q->c_a = q->c_z;          // (selects the data-gradient code; c_a_store = 0: nothing is written through it)
  }
  if (!q->W0 || !q->W1 || !q->W2 || !q->W3 || !q->t0 || !q->t1 || !q->t2) { tp::set_error("%s: null pointer", what); return -1; }
  return 0;
}

// one problem (qb == nullptr) or two in one launch
template <class K>
int launch(K kernel, const TailP& q, int grid, size_t lds, tp_stream_t stream, const char* what, unsigned long long& flags,
           const TailP* qb = nullptr, int grid_b = 0, size_t lds_b = 0) {
  if (lds_b > lds) lds = lds_b;
  if (lds > 150 * 1024) { tp::set_error("%s: head too wide for one workgroup's LDS", what); return -1; }
  if (lds > 48 * 1024 && tp::first_use_on_device(flags) &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess) {
    tp::set_error("%s: cannot raise the LDS limit", what);
    return -1;
  }
  if (qb != nullptr) hipLaunchKernelGGL(kernel, dim3(grid + grid_b), dim3(kT), lds, (hipStream_t)stream, q, *qb, grid);
  else hipLaunchKernelGGL(kernel, dim3(grid), dim3(kT), lds, (hipStream_t)stream, q, q, -1);
  return tp::check_launch(what);
}

int splits_for(const TailP& q) {
  int S = 256 / q.N;                                // ~256 workgroups
  if (S < 1) S = 1;
  if (S > 64) S = 64;
  while (S > 1 && q.K / S < 4 * kT) S >>= 1;        // at least one 16-byte load per thread and slice
  return S;
}
}  // namespace

extern "C" {
#ifdef TP_TAIL_TIMING
int tp_disc_tail_stamps(unsigned long long* host16) {      // (diagnostic build only: not part of the C ABI)
  return (int)hipMemcpyFromSymbol(host16, HIP_SYMBOL(g_tail_stamps), sizeof(unsigned long long) * 16);
}
#endif
size_t tp_disc_tail_workspace_bytes(int N) { return (size_t)N * 64 * kMaxM * sizeof(float); }     // [N][S <= 64][kMaxM]

int tp_disc_tail_fwd(const tp_disc_tail_args* a, tp_stream_t stream) {
  static unsigned long long flags[3] = {0, 0, 0};
  TailP q{};
  if (int rc = fill(&q, a, "tp_disc_tail_fwd")) return rc;
  TP_REQUIRE(q.a && q.scale && q.out && q.ws && q.ticket, "operand missing");
  q.S = splits_for(q);
  if (q.M <= 4) return launch(disc_tail_fwd_kernel<false, 4>, q, q.N * q.S, lds_fwd(q), stream, "tp_disc_tail_fwd", flags[0]);
  if (q.M <= 8) return launch(disc_tail_fwd_kernel<false, 8>, q, q.N * q.S, lds_fwd(q), stream, "tp_disc_tail_fwd", flags[1]);
  return launch(disc_tail_fwd_kernel<false, 16>, q, q.N * q.S, lds_fwd(q), stream, "tp_disc_tail_fwd", flags[2]);
}
int tp_disc_tail_fwd_pair(const tp_disc_tail_args* a, const tp_disc_tail_args* b, tp_stream_t stream) {
  static unsigned long long flags[3] = {0, 0, 0};
  TailP qa{}, qb{};
  TP_REQUIRE(a && b, "null argument");
  if (int rc = fill(&qa, a, "tp_disc_tail_fwd_pair")) return rc;
  if (int rc = fill(&qb, b, "tp_disc_tail_fwd_pair")) return rc;
  TP_REQUIRE(qa.a && qa.scale && qa.out && qa.ws && qa.ticket && qb.a && qb.scale && qb.out && qb.ws && qb.ticket, "operand missing");
  TP_REQUIRE(qa.ws != qb.ws && qa.ticket != qb.ticket, "the two problems of a pair need their own workspace / ticket");
  qa.S = splits_for(qa); qb.S = splits_for(qb);
  const int rows = qa.M > qb.M ? qa.M : qb.M, ga = qa.N * qa.S, gb = qb.N * qb.S;
  if (rows <= 4) return launch(disc_tail_fwd_kernel<false, 4>, qa, ga, lds_fwd(qa), stream, "tp_disc_tail_fwd_pair", flags[0], &qb, gb, lds_fwd(qb));
  if (rows <= 8) return launch(disc_tail_fwd_kernel<false, 8>, qa, ga, lds_fwd(qa), stream, "tp_disc_tail_fwd_pair", flags[1], &qb, gb, lds_fwd(qb));
  return launch(disc_tail_fwd_kernel<false, 16>, qa, ga, lds_fwd(qa), stream, "tp_disc_tail_fwd_pair", flags[2], &qb, gb, lds_fwd(qb));
}
int tp_disc_tail_bwd_bwd(const tp_disc_tail_args* a, tp_stream_t stream) {
  static unsigned long long flags[3] = {0, 0, 0};
  TailP q{};
  if (int rc = fill(&q, a, "tp_disc_tail_bwd_bwd")) return rc;
  TP_REQUIRE(q.a && q.g && q.e1 && q.e2 && q.gW1 && q.gW2 && q.gW3 && q.ws && q.ticket, "operand missing");
  q.S = splits_for(q);
  if (q.M <= 4) return launch(disc_tail_fwd_kernel<true, 4>, q, q.N * q.S, lds_fwd(q), stream, "tp_disc_tail_bwd_bwd", flags[0]);
  if (q.M <= 8) return launch(disc_tail_fwd_kernel<true, 8>, q, q.N * q.S, lds_fwd(q), stream, "tp_disc_tail_bwd_bwd", flags[1]);
  return launch(disc_tail_fwd_kernel<true, 16>, q, q.N * q.S, lds_fwd(q), stream, "tp_disc_tail_bwd_bwd", flags[2]);
}
static int bwd_fill(TailP* q, const tp_disc_tail_args* a, const char* what) {
  if (int rc = fill(q, a, what)) return rc;
  TP_REQUIRE(q->g != nullptr, "g_out missing");
  TP_REQUIRE((q->gW1 && q->gW2 && q->gW3) || (!q->gW1 && !q->gW2 && !q->gW3), "gW1..3: all or none");
  TP_REQUIRE(q->gW0 == nullptr || q->a != nullptr, "gW0 needs the ladder output a");
  TP_REQUIRE(q->M2 == 0 || (q->gy2 && q->a2 && q->gW0), "the extra rows belong to the weight gradient gW0");
  return 0;
}
int tp_disc_tail_bwd_pair(const tp_disc_tail_args* a, const tp_disc_tail_args* b, tp_stream_t stream) {
  static unsigned long long flags[3] = {0, 0, 0};
  TailP qa{}, qb{};
  TP_REQUIRE(a && b, "null argument");
  if (int rc = bwd_fill(&qa, a, "tp_disc_tail_bwd_pair")) return rc;
  if (int rc = bwd_fill(&qb, b, "tp_disc_tail_bwd_pair")) return rc;
  const int ra = qa.M > qa.M2 ? qa.M : qa.M2, rb = qb.M > qb.M2 ? qb.M : qb.M2, rows = ra > rb ? ra : rb;
  const int ga = (qa.K + 63) / 64, gb = (qb.K + 63) / 64;
  if (rows <= 4) return launch(disc_tail_bwd_kernel<4>, qa, ga, lds_bwd(qa), stream, "tp_disc_tail_bwd_pair", flags[0], &qb, gb, lds_bwd(qb));
  if (rows <= 8) return launch(disc_tail_bwd_kernel<8>, qa, ga, lds_bwd(qa), stream, "tp_disc_tail_bwd_pair", flags[1], &qb, gb, lds_bwd(qb));
  return launch(disc_tail_bwd_kernel<16>, qa, ga, lds_bwd(qa), stream, "tp_disc_tail_bwd_pair", flags[2], &qb, gb, lds_bwd(qb));
}
int tp_disc_tail_bwd(const tp_disc_tail_args* a, tp_stream_t stream) {
  static unsigned long long flags[3] = {0, 0, 0};
  TailP q{};
  if (int rc = bwd_fill(&q, a, "tp_disc_tail_bwd")) return rc;
  const int rows = q.M > q.M2 ? q.M : q.M2, grid = (q.K + 63) / 64;
  if (rows <= 4) return launch(disc_tail_bwd_kernel<4>, q, grid, lds_bwd(q), stream, "tp_disc_tail_bwd", flags[0]);
  if (rows <= 8) return launch(disc_tail_bwd_kernel<8>, q, grid, lds_bwd(q), stream, "tp_disc_tail_bwd", flags[1]);
  return launch(disc_tail_bwd_kernel<16>, q, grid, lds_bwd(q), stream, "tp_disc_tail_bwd", flags[2]);
}
}
