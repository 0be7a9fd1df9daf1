// K4: per-ray alpha composite, forward and backward (gfx950).
//
// Replaces NeRF.composite (reference layers/nerf_static_transient_light.py:168-212): ~30
// element-wise launches + 3 cumsums, each a full HBM round trip, become one streaming pass.
// HBM-bound: algorithmic traffic = 40 B read + 8 B written per sample (+4 B with prob) and
// 12 B read + 56 B written per ray.
//
// Mapping: one 64-lane wavefront per ray, lanes stride over the N samples in chunks of 64, so every
// load/store of a chunk is a contiguous segment.  The three transmittances (joint, static-only,
// transient-only) are exclusive prefix sums of tau = sigma * delta; they are computed with a
// wave-level shuffle scan plus a scalar carry between chunks, and the 14 per-ray sums are reduced
// with a butterfly at the end.  No LDS, no atomics, no inter-wave communication.
#include "tp_common.h"

namespace {

constexpr int kWaves = 4;  // waves (= rays in flight) per workgroup

// Wave scans and sums on the vector ALU (tp_common.h: DPP).  Round 4 built them from __shfl (ds_bpermute_b32: 108 LDS instructions per
// 64 samples in the forward, 186 in the backward) and the LDS pipe, not HBM, set both kernels' time.
__device__ __forceinline__ float wave_incl_scan(float v, int) { return tp::wave_scan_dpp(v); }
__device__ __forceinline__ float wave_sum(float v) { return tp::wave_total_dpp(v); }

// Everything the forward derives for one sample.
struct Sample {
  float cs[3], ct[3], z, u, dist;
  float T, Ts, Tt;     // transmittances before this sample
  float es, et, e;     // exp(-tau_s), exp(-tau_t), exp(-tau)
  float as, at, a;     // alphas
};

struct Carry { float s, t, j; };

// The raw inputs of one sample: colours, densities, depth, uncertainty, and the distance to the next sample (already times |ray|).
struct Raw { float cs[3], ct[3], sig_s, sig_t, z, u, dist; };

// From global memory: sample i of ray q (all 64 lanes must call: the next depth comes from the neighbouring lane).
// SQ (the forward): q is wave-uniform and held in scalar registers -- the kernel reads its wavefront index with readfirstlane --, so the
// ray's base addresses are scalar arithmetic and every access is base + 32-bit lane offset instead of 64-bit vector address arithmetic
// (316 -> 300 vector instructions per chunk).  The backward keeps the vector form: there the scalar form spills scalar registers.
template <bool SQ>
__device__ __forceinline__ Raw fetch_global(const tp_composite_args& p, int64_t q, int i, int lane, float len) {
  Raw r;
  const bool ok = i < p.N;
  const int64_t e0 = SQ ? q * p.N : 0;
  const int64_t e = SQ ? (int64_t)(ok ? (unsigned)i : 0u) : q * p.N + (ok ? i : 0);
  r.sig_s = 0.f; r.sig_t = 0.f; r.z = 0.f; r.u = 0.f;
#pragma unroll
  for (int c = 0; c < 3; ++c) { r.cs[c] = 0.f; r.ct[c] = 0.f; }
  if (ok) {
    const float2* rp = reinterpret_cast<const float2*>(p.rgb + e0 * 6) + e * 3;
    const float2 r0 = rp[0], r1 = rp[1], r2 = rp[2];
    r.cs[0] = r0.x; r.ct[0] = r0.y; r.cs[1] = r1.x; r.ct[1] = r1.y; r.cs[2] = r2.x; r.ct[2] = r2.y;
    const float2 dn = reinterpret_cast<const float2*>(p.density + e0 * 2)[e];
    r.sig_s = dn.x; r.sig_t = dn.y;
    r.z = (p.depth + e0)[e];
    r.u = (p.uncert + e0)[e];
  }
  // the next sample's depth: the neighbouring lane's (lane 63: the next chunk's first, its own load)
  float zn = tp::wave_shl1(r.z);
  if (lane == 63 && i + 1 < p.N) zn = (p.depth + e0)[e + 1];
  const float dz = (i == p.N - 1) ? 1e10f : (zn - r.z);
  r.dist = ok ? dz * len : 0.f;
  return r;
}

// exp(x) for x <= 0 (optical depths are non-negative): 2^(x log2 e) on the hardware's exp2 (1 ulp) with the rounding of the product
// repaired to first order -- t = fl(x L), r = (x L - t) + x L_lo exactly enough, e^x = 2^t (1 + r ln 2).  Eight vector instructions; libm's
// expf spends eleven, four of them on the overflow / denormal range no transmittance reaches (x = -inf -> 0 and NaN -> NaN as in expf;
// results below 2^-126 flush to zero: they multiply colours in [0,1]).  Max error 1.5 ulp against expf's 1 (tests: G7 at 1e-4 / 1e-6).
__device__ __forceinline__ float exp_neg(float x) {
  const float L = 1.44269502162933349609375f, L_lo = 1.925963033500011e-8f;      // log2(e) = L + L_lo
  const float t = x * L;
  float r = __fmaf_rn(x, L, -t) + x * L_lo;
  r = x < -1e30f ? 0.f : r;                                // (t may have overflowed to -inf: 0 * inf otherwise; NaN stays NaN)
  const float e = __builtin_amdgcn_exp2f(t);
  return __fmaf_rn(e, r * 0.693147182464599609375f, e);
}

// The scan of the current chunk on top of the carry (all 64 lanes must call; `ok`: this lane holds a sample).
__device__ __forceinline__ Sample scan_sample(const Raw& r, bool ok, Carry& carry) {
  Sample s;
#pragma unroll
  for (int c = 0; c < 3; ++c) { s.cs[c] = r.cs[c]; s.ct[c] = r.ct[c]; }
  s.z = r.z; s.u = r.u; s.dist = r.dist;
  const float ts = r.sig_s * s.dist, tt = r.sig_t * s.dist, tj = ts + tt;
  float sc[3] = {ts, tt, tj};
  tp::wave_scan_dpp(sc);                                  // the three scans step by step together
  const float is = sc[0], it = sc[1], ij = sc[2];
  const float xs = tp::wave_shr1(is), xt = tp::wave_shr1(it), xj = tp::wave_shr1(ij);
  s.Ts = exp_neg(-(carry.s + xs)); s.Tt = exp_neg(-(carry.t + xt)); s.T = exp_neg(-(carry.j + xj));
  s.es = exp_neg(-ts); s.et = exp_neg(-tt); s.e = exp_neg(-tj);
  s.as = 1.f - s.es; s.at = 1.f - s.et; s.a = 1.f - s.e;
  carry.s += tp::lane_value(is, 63); carry.t += tp::lane_value(it, 63); carry.j += tp::lane_value(ij, 63);
  if (!ok) { s.T = s.Ts = s.Tt = 0.f; }   // padding lanes contribute nothing
  return s;
}

// Loads sample i of ray q and runs the scan for the current chunk (all 64 lanes must call).
template <bool SQ = false>
__device__ __forceinline__ Sample load_sample(const tp_composite_args& p, int64_t q, int i, int lane, float len,
                                              Carry& carry) {
  const Raw r = fetch_global<SQ>(p, q, i, lane, len);
  return scan_sample(r, i < p.N, carry);
}

__device__ __forceinline__ float ray_len(const float* ray, int64_t q) {
  const float a = ray[3 * q], b = ray[3 * q + 1], c = ray[3 * q + 2];
  return sqrtf(a * a + b * b + c * c);
}

__global__ __launch_bounds__(kWaves * 64) void composite_fwd_kernel(tp_composite_args p) {
  const int lane = threadIdx.x & 63;
  const int64_t wave0 = (int64_t)blockIdx.x * kWaves + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t stride = (int64_t)gridDim.x * kWaves;
  for (int64_t q = wave0; q < p.n; q += stride) {
    const float len = ray_len(p.ray, q);
    Carry carry = {0.f, 0.f, 0.f};
    float acc[14];
#pragma unroll
    for (int k = 0; k < 14; ++k) acc[k] = 0.f;
    for (int base = 0; base < p.N; base += 64) {
      const int i = base + lane;
      const Sample s = load_sample<true>(p, q, i, lane, len, carry);
      const float ws = s.T * s.as, wt = s.T * s.at, w = s.T * s.a;
      const float os = s.Ts * s.as, ot = s.Tt * s.at;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        acc[c] += s.cs[c] * ws + s.ct[c] * wt;
        acc[3 + c] += os * s.cs[c];
        acc[6 + c] += ot * s.ct[c];
      }
      acc[9] += s.z * os;
      acc[10] += w;
      acc[11] += os;
      acc[12] += ot;
      acc[13] += s.u * wt;
      if (i < p.N) {
        const int64_t e0 = q * p.N;
        if (p.alpha_static) (p.alpha_static + e0)[(unsigned)i] = s.as;
        if (p.alpha_transient) (p.alpha_transient + e0)[(unsigned)i] = s.at;
        if (p.prob) (p.prob + e0)[(unsigned)i] = w;
      }
    }
    tp::wave_totals14(acc);
    if (lane == 0) {
      acc[13] += p.min_uncert;
      float2* o = reinterpret_cast<float2*>(p.out_ray + q * 14);
#pragma unroll
      for (int k = 0; k < 7; ++k) o[k] = make_float2(acc[2 * k], acc[2 * k + 1]);
      if (p.rgb_ray) { p.rgb_ray[q * 3] = acc[0]; p.rgb_ray[q * 3 + 1] = acc[1]; p.rgb_ray[q * 3 + 2] = acc[2]; }
      if (p.uncert_ray) p.uncert_ray[q] = acc[13];
    }
  }
}

__device__ __forceinline__ float wave_rev_incl_scan(float v, int lane) { return tp::wave_rev_scan_dpp(v, lane); }

constexpr int kMaxChunks = 32;   // backward supports N <= 2048 samples per ray

// Backward.  With w_i = T_i a_i, T_i = exp(-sum_{j<i} tau_j):  d(sum_i w_i v_i)/d tau_k =
// T_k exp(-tau_k) v_k - sum_{i>k} w_i v_i, applied to the joint / static-only / transient-only
// families.  The suffix sums are built by walking the ray BACKWARDS with a reverse wave scan: forming
// them as (total - prefix) leaves a rounding residue that the 1e10-long last interval multiplies into
// O(1e3) garbage, whereas the reference's autograd gives the last sample an exactly-zero suffix.
// Pass 0 walks forward only to record the transmittance carry at each 64-sample chunk start.
__global__ __launch_bounds__(kWaves * 64) void composite_bwd_kernel(tp_composite_bwd_args b) {
  __shared__ float s_carry[kWaves][3][kMaxChunks];
  const tp_composite_args& p = b.fwd;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t wave0 = (int64_t)blockIdx.x * kWaves + wv;
  const int64_t stride = (int64_t)gridDim.x * kWaves;
  const int n_chunks = (p.N + 63) / 64;
  for (int64_t q = wave0; q < p.n; q += stride) {
    const float len = ray_len(p.ray, q);
    // the ray's 14 cotangents are wave-uniform: lane k < 14 loads (and sums) the k-th, then they are read lane by lane into scalar
    // registers -- one coalesced load per source instead of 14 same-address loads per lane
    float gl = (b.g_out_ray && lane < 14) ? b.g_out_ray[q * 14 + lane] : 0.f;
    if (lane < 3) {
      if (b.g_rgb_ray) gl += b.g_rgb_ray[q * 3 + lane];
      // cotangents of the two aliases of rgb_ray (each consumer of the colours back-propagates into its own: no add launches)
      if (b.g_rgb_ray2) gl += b.g_rgb_ray2[q * 3 + lane];
      if (b.g_rgb_ray3) gl += b.g_rgb_ray3[q * 3 + lane];
    }
    if (lane == 13 && b.g_uncert_ray) gl += b.g_uncert_ray[q];
    float g[14];
#pragma unroll
    for (int k = 0; k < 14; ++k) g[k] = tp::lane_value(gl, k);
    Carry carry = {0.f, 0.f, 0.f};
    for (int c = 0; c < n_chunks; ++c) {
      if (lane == 0) { s_carry[wv][0][c] = carry.s; s_carry[wv][1][c] = carry.t; s_carry[wv][2][c] = carry.j; }
      (void)load_sample(p, q, c * 64 + lane, lane, len, carry);
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the wave's own LDS writes are visible to its reads
    float run1 = 0.f, run2 = 0.f, run3 = 0.f;    // suffix sums carried in from the chunks behind
    for (int c = n_chunks - 1; c >= 0; --c) {
      const int i = c * 64 + lane;
      const bool ok = i < p.N;
      const int64_t e = q * p.N + (ok ? i : 0);
      Carry cc = {s_carry[wv][0][c], s_carry[wv][1][c], s_carry[wv][2][c]};
      const Sample s = load_sample(p, q, i, lane, len, cc);
      const float gp = (b.g_prob && ok) ? b.g_prob[e] : 0.f;
      const float A = g[0] * s.cs[0] + g[1] * s.cs[1] + g[2] * s.cs[2];
      const float Bv = g[0] * s.ct[0] + g[1] * s.ct[1] + g[2] * s.ct[2] + g[13] * s.u;
      const float C = g[10] + gp;
      const float D = g[3] * s.cs[0] + g[4] * s.cs[1] + g[5] * s.cs[2] + g[9] * s.z + g[11];
      const float E = g[6] * s.ct[0] + g[7] * s.ct[1] + g[8] * s.ct[2] + g[12];
      const float P1 = s.T * (s.as * A + s.at * Bv + s.a * C);     // padding lanes: T = 0 -> 0
      const float P2 = s.Ts * s.as * D;
      const float P3 = s.Tt * s.at * E;
      float rv[3] = {P1, P2, P3};
      tp::wave_rev_scan_dpp(rv, lane);
      const float r1 = rv[0], r2 = rv[1], r3 = rv[2];
      float suf1 = tp::wave_shl1(r1), suf2 = tp::wave_shl1(r2), suf3 = tp::wave_shl1(r3);
      suf1 += run1; suf2 += run2; suf3 += run3;
      run1 += tp::lane_value(r1, 0); run2 += tp::lane_value(r2, 0); run3 += tp::lane_value(r3, 0);
      if (!ok) continue;
      const float gas = b.g_alpha_static ? b.g_alpha_static[e] : 0.f;
      const float gat = b.g_alpha_transient ? b.g_alpha_transient[e] : 0.f;
      const float dts = s.T * (s.es * A + s.e * C) - suf1 + s.Ts * s.es * D - suf2 + gas * s.es;
      const float dtt = s.T * (s.et * Bv + s.e * C) - suf1 + s.Tt * s.et * E - suf3 + gat * s.et;
      float2 gd = make_float2(dts * s.dist, dtt * s.dist);
      if (b.g_density_add) {            // a second cotangent of the densities (the transient regulariser's) joins here
        const float2 ad = *reinterpret_cast<const float2*>(b.g_density_add + e * 2);
        gd.x += ad.x; gd.y += ad.y;
      }
      *reinterpret_cast<float2*>(b.g_density + e * 2) = gd;
      const float ws = s.T * s.as, wt = s.T * s.at, os = s.Ts * s.as, ot = s.Tt * s.at;
      float2* gr = reinterpret_cast<float2*>(b.g_rgb + e * 6);
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) gr[ch] = make_float2(ws * g[ch] + os * g[3 + ch], wt * g[ch] + ot * g[6 + ch]);
      b.g_uncert[e] = wt * g[13];
    }
  }
}

int grid_for(int64_t n) {
  int64_t blocks = (n + kWaves - 1) / kWaves;
  const int64_t cap = 256 * 8 * 4;  // 8 workgroups per CU, 4 rounds: plenty of bytes in flight
  return (int)(blocks < cap ? blocks : cap);
}

}  // namespace

extern "C" int tp_composite_fwd(const tp_composite_args* a, tp_stream_t stream) {
  TP_REQUIRE(a && a->ray && a->rgb && a->density && a->depth && a->uncert && a->out_ray, "null pointer");
  TP_REQUIRE(a->N > 0 && a->n >= 0, "bad sizes");
  if (a->n == 0) return 0;
  // (Measured and dropped, profiles/r5/03: nontemporal loads of the inputs -- no change; a variant that brings 128 samples in by fully
  // coalesced 16-byte loads, parks them in wavefront-private LDS and prefetches the next stage -- 4 % slower: with the scans on the
  // vector ALU the kernel sits between its instruction count, ~500 per ray and wavefront, and the memory system; a pure 16-byte read
  // stream reaches 5.4 TB/s on this chip, tools/ubench/read_bw.hip.)
  hipLaunchKernelGGL(composite_fwd_kernel, dim3(grid_for(a->n)), dim3(kWaves * 64), 0, (hipStream_t)stream, *a);
  return tp::check_launch("tp_composite_fwd");
}

extern "C" int tp_composite_bwd(const tp_composite_bwd_args* a, tp_stream_t stream) {
  TP_REQUIRE(a && a->fwd.ray && a->fwd.rgb && a->fwd.density && a->fwd.depth && a->fwd.uncert, "null forward input");
  // every cotangent is optional (the kernel reads a NULL one as zero): a step whose only consumers are the fan-out aliases
  // (g_rgb_ray2 / 3, g_density_add) is as valid as one with g_out_ray; only the three outputs are required
  TP_REQUIRE(a->g_rgb && a->g_density && a->g_uncert, "null gradient output pointer");
  TP_REQUIRE(a->fwd.N > 0 && a->fwd.n >= 0, "bad sizes");
  TP_REQUIRE(a->fwd.N <= 64 * kMaxChunks, "composite backward supports at most 2048 samples per ray");
  if (a->fwd.n == 0) return 0;
  hipLaunchKernelGGL(composite_bwd_kernel, dim3(grid_for(a->fwd.n)), dim3(kWaves * 64), 0, (hipStream_t)stream, *a);
  return tp::check_launch("tp_composite_bwd");
}
