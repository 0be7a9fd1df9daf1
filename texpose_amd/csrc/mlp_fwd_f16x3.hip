// K2 (fast path): fused posenc + static/transient/light MLP forward on the f16 matrix cores with
// fp32-grade accuracy ("f16x3").  Same interface, schedule and register-resident dataflow as
// mlp_fwd.hip; what changes is the arithmetic of each product:
//
//   gfx950 has no TF32/xf32, and v_mfma_f32_32x32x2_f32 runs at 1/16 of the f16 MFMA rate.  Here every
//   fp32 operand (weights offline, activations in the layer epilogue) is carried as an unevaluated sum
//   hi + lo of two fp16 numbers (11 + 11 significand bits), and   W x  ~=  Whi xhi + Whi xlo + Wlo xhi
//   is issued as three v_mfma_f32_32x32x16_f16 with fp32 accumulation.  Products of two fp16 values are
//   exact in fp32, the dropped lo*lo term is 2^-22 relative, so the result is as accurate as an fp32
//   FMA chain (measured vs an fp64 oracle: within 1.3x of torch fp32) at 16/3 = 5.3x
//   the fp32-MFMA throughput.  Weights are pre-scaled by 2^8 (exact) so that their lo parts stay in the
//   fp16 normal range; the epilogue folds the 2^-8 into the bias FMA.  Activations must stay below 6e4
//   (NeRF activations are O(1..100)); the kernel raises bit 0 of a status word otherwise.
//
// An accumulator tile is reused as the next layer's B operand exactly as in the fp32 kernel: registers
// 8s..8s+7 of a 32x32 tile, converted to fp16, ARE the B fragment of k-step s (rows 16s + 8(j>>2) + 4h + (j&3)),
// and the packed weights absorb that row permutation (mlp_layout.h, "f16x3 stream").
#include "mlp_mma.h"

namespace {
using namespace tp_layout;
using namespace tp_mma;

using half8 = __attribute__((ext_vector_type(8))) _Float16;
using half2v = __attribute__((ext_vector_type(2))) __fp16;

constexpr int kBiasPad = (kBiasFloats + 63) / 64 * 64;
constexpr int kStageHalves = 5 * 2 * kThreads * 8;             // 5 k-steps x (hi, lo) x 256 lanes x 8 halves = 40 KiB
constexpr int kBufs = 3;                                         // weight-chunk ring: two chunks (3072 cycles) ahead
constexpr int kLdsBytes = kBufs * kChunkFloats * 4 + kBiasPad * 4 + kStageHalves * 2;
constexpr float kInvScale = 1.0f / (float)(1 << kF16WeightShift);

using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
__device__ __forceinline__ half8 pack8(const half2v (&p)[4]) {
  const u32x4 w = {__builtin_bit_cast(unsigned int, p[0]), __builtin_bit_cast(unsigned int, p[1]),
                   __builtin_bit_cast(unsigned int, p[2]), __builtin_bit_cast(unsigned int, p[3])};
  return __builtin_bit_cast(half8, w);
}

__device__ __forceinline__ f32x16 mfma16(half8 a, half8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// B operands of one 256-feature activation: [source tile][k-step of 16 features]
struct XF {
  half8 hi[8][2];
  half8 lo[8][2];
};

// Three-slot ring instead of the fp32 kernel's double buffer: a chunk is only 48 MFMAs (1536 cycles, ~0.65 us)
// here, shorter than an L2->LDS DMA round trip, so the prefetch runs TWO chunks ahead.  The DMA of chunk c+2 stays
// in flight across the barrier: counted `s_waitcnt vmcnt(8)` (the 8 DMA instructions of the newest chunk may be
// outstanding, everything older -- chunk c+1 -- has landed) + raw s_barrier; __syncthreads() would drain vmcnt(0).

__device__ __forceinline__ void ring_begin(Pipe& p) {
  int nxt = p.chunk + 2;
  if (nxt >= kNumChunks) nxt -= kNumChunks;
  int slot = p.buf + 2;
  if (slot >= kBufs) slot -= kBufs;
  dma_chunk(p, nxt, slot);
}
__device__ __forceinline__ void ring_end(Pipe& p) {
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  p.chunk = (p.chunk + 1 == kNumChunks) ? 0 : p.chunk + 1;
  p.buf = (p.buf + 1 == kBufs) ? 0 : p.buf + 1;
}
__device__ __forceinline__ const _Float16* chunk_ptr16(const Pipe& p) {
  return reinterpret_cast<const _Float16*>(p.lds + p.buf * kChunkFloats) + p.lane * 8;
}

// (k-step, tile) pairs of a wide chunk; pair q = s*8 + t.  The hi/lo A fragments are fetched kDepth pairs
// (kDepth x 96 MFMA cycles) ahead of their use: with one wave per SIMD nothing else hides the LDS latency.
constexpr int kDepth = 4;
template <int KS, class BFn>
__device__ __forceinline__ void mma_wide16(f32x16 (&acc)[8], const _Float16* l, BFn b) {
  constexpr int NP = KS * 8;
  half8 fh[kDepth], fl[kDepth];
#pragma unroll
  for (int q = 0; q < kDepth && q < NP; ++q) {
    fh[q] = *reinterpret_cast<const half8*>(l + (q * 2 + 0) * 512);
    fl[q] = *reinterpret_cast<const half8*>(l + (q * 2 + 1) * 512);
  }
  __builtin_amdgcn_sched_group_barrier(0x100, 2 * (kDepth < NP ? kDepth : NP), 0);
  // two tiles at a time with their three products interleaved: consecutive MFMAs never share an accumulator
#pragma unroll
  for (int q = 0; q < NP; q += 2) {
    const int s = q >> 3, t = q & 7;
    const half8 wh0 = fh[q % kDepth], wl0 = fl[q % kDepth];
    const half8 wh1 = fh[(q + 1) % kDepth], wl1 = fl[(q + 1) % kDepth];
    half8 xh, xl;
    b(s, xh, xl);
    acc[t] = mfma16(wh0, xh, acc[t]);
    acc[t + 1] = mfma16(wh1, xh, acc[t + 1]);
    acc[t] = mfma16(wh0, xl, acc[t]);
    acc[t + 1] = mfma16(wh1, xl, acc[t + 1]);
    acc[t] = mfma16(wl0, xh, acc[t]);
    acc[t + 1] = mfma16(wl1, xh, acc[t + 1]);
#pragma unroll
    for (int d = 0; d < 2; ++d)
      if (q + d + kDepth < NP) {
        fh[(q + d) % kDepth] = *reinterpret_cast<const half8*>(l + ((q + d + kDepth) * 2 + 0) * 512);
        fl[(q + d) % kDepth] = *reinterpret_cast<const half8*>(l + ((q + d + kDepth) * 2 + 1) * 512);
      }
    // hipcc waits with lgkmcnt(0) before the first MFMA that consumes a fragment; issuing the next fetches right
    // AFTER that MFMA (not before it) makes the wait cover only fetches that are already >= 5 MFMAs old
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    if (q + kDepth < NP) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 5, 0);
  }
}

__device__ __forceinline__ void part_gen16(Pipe& p, f32x16 (&acc)[8], const XF& X) {
#pragma unroll
  for (int ts = 0; ts < 8; ++ts) {
    ring_begin(p);
    mma_wide16<2>(acc, chunk_ptr16(p), [&](int s, half8& xh, half8& xl) { xh = X.hi[ts][s]; xl = X.lo[ts][s]; });
    ring_end(p);
  }
}

// 1..5-row output layer: one chunk, 16 k-steps, one accumulator tile
__device__ __forceinline__ f32x16 part_head16(Pipe& p, const XF& X) {
  f32x16 acc = {0};
  ring_begin(p);
  const _Float16* l = chunk_ptr16(p);
  half8 wh = *reinterpret_cast<const half8*>(l);
  half8 wl = *reinterpret_cast<const half8*>(l + 512);
  __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
  for (int s16 = 0; s16 < 16; ++s16) {
    half8 nh = wh, nl = wl;
    if (s16 + 1 < 16) {
      nh = *reinterpret_cast<const half8*>(l + ((s16 + 1) * 2 + 0) * 512);
      nl = *reinterpret_cast<const half8*>(l + ((s16 + 1) * 2 + 1) * 512);
    }
    const half8 xh = X.hi[s16 >> 1][s16 & 1], xl = X.lo[s16 >> 1][s16 & 1];
    acc = mfma16(wh, xh, acc);
    acc = mfma16(wh, xl, acc);
    acc = mfma16(wl, xh, acc);
    if (s16 + 1 < 16) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
    wh = nh; wl = nl;
  }
  ring_end(p);
  return acc;
}

__device__ __forceinline__ float softplus(float x) { return x > 20.0f ? x : log1pf(expf(x)); }
__device__ __forceinline__ float sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

struct Params {
  const float* packed;
  const float* center; const float* ray; const float* depth;
  const float* points; const float* ray_unit;
  const float* lat_trans; const float* lat_light;
  int B, R, N;
  int64_t n_samples, n_tiles;
  float* rgb; float* density; float* uncert; float* workspace; int* status;
};

// stage one "extra input" value as hi/lo halves: slot = 16 ks + 8 h + j of this lane
__device__ __forceinline__ void stage(_Float16* st, int tid, int ks, int j, float v) {
  const _Float16 hi = (_Float16)v;
  st[((ks * 2 + 0) * kThreads + tid) * 8 + j] = hi;
  st[((ks * 2 + 1) * kThreads + tid) * 8 + j] = (_Float16)(v - (float)hi);
}

__global__ __launch_bounds__(kThreads, 1) void mlp_fwd_f16x3_kernel(Params P) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, hh = lane >> 5;
  float* bias_lds = lds + kBufs * kChunkFloats;
  _Float16* st = reinterpret_cast<_Float16*>(bias_lds + kBiasPad);

  Pipe p;
  p.stream = P.packed; p.lds = lds; p.chunk = 0; p.buf = 0; p.wave = wave; p.lane = lane;
  for (int i = tid; i < kBiasFloats; i += kThreads) bias_lds[i] = P.packed[(size_t)kNumChunks * kChunkFloats + i];
  dma_chunk(p, 0, 0);
  dma_chunk(p, 1, 1);
  __syncthreads();

  half8* ws = reinterpret_cast<half8*>(P.workspace + (size_t)blockIdx.x * (128 * 256)) + tid;
  const auto staged = [&](int s, half8& xh, half8& xl, int ks0) {
    xh = *reinterpret_cast<const half8*>(st + (((ks0 + s) * 2 + 0) * kThreads + tid) * 8);
    xl = *reinterpret_cast<const half8*>(st + (((ks0 + s) * 2 + 1) * kThreads + tid) * 8);
  };

  for (int64_t tile = blockIdx.x; tile < P.n_tiles; tile += gridDim.x) {
    const int64_t s_raw = tile * 128 + wave * 32 + j;
    const bool live = s_raw < P.n_samples;
    const int64_t s = live ? s_raw : P.n_samples - 1;
    const int64_t q = s / P.N;
    const int b = (int)(q / P.R);
    float x[3], vu[3];
    if (P.center != nullptr) {
      const float z = P.depth[s];
      float nrm = 0.f;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float d = P.ray[3 * q + c];
        x[c] = tp::add_rn(P.center[3 * q + c], tp::mul_rn(d, z));
        nrm = tp::add_rn(nrm, tp::mul_rn(d, d));
        vu[c] = d;
      }
      const float den = fmaxf(sqrtf(nrm), 1e-12f);
#pragma unroll
      for (int c = 0; c < 3; ++c) vu[c] = tp::div_rn(vu[c], den);
    } else {
#pragma unroll
      for (int c = 0; c < 3; ++c) { x[c] = P.points[3 * s + c]; vu[c] = P.ray_unit[3 * s + c]; }
    }

    XF X;
    f32x16 acc[8];
    float amax = 0.f;
    float sig_s = 0.f, sig_t = 0.f, unc = 0.f, rgb_t[3] = {0.f, 0.f, 0.f}, rgb_s[3] = {0.f, 0.f, 0.f};

#pragma nounroll
    for (int li = 0; li < kNumWide; ++li) {
      if (li == L7) {
        const f32x16 a = part_head16(p, X);
        sig_s = softplus(fmaf(a[0], kInvScale, bias_lds[kHeadBiasOff + 0]));
      }
      if (li == R0) {
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            X.hi[t][k] = ws[((t * 2 + k) * 2 + 0) * kThreads];
            X.lo[t][k] = ws[((t * 2 + k) * 2 + 1) * kThreads];
          }
      }
#pragma unroll
      for (int t = 0; t < 8; ++t) acc[t] = f32x16{0};

      if (li != L0) part_gen16(p, acc, X);

      if (li == L0 || li == L4) {
        // [PE(x) | x | pad] in natural column order; this lane stages slots 16 ks + 8 h + jj
#pragma nounroll
        for (int e = 0; e < 32; ++e) {
          const int ks = e >> 3, jj = e & 7, slot = 16 * ks + 8 * hh + jj;
          float v;
          if (slot < 60) {
            const int c = slot / 20, rem = slot - c * 20, sc = rem / 10, l = rem - sc * 10;
            const float xc = c == 0 ? x[0] : (c == 1 ? x[1] : x[2]);
            const float arg = tp::mul_rn(xc, ldexpf(3.14159274101257324f, l));
            v = tp::sincos_sel(arg, sc);
          } else {
            v = slot == 60 ? x[0] : (slot == 61 ? x[1] : (slot == 62 ? x[2] : 0.0f));
          }
          stage(st, tid, ks, jj, v);
        }
#pragma unroll
        for (int qd = 0; qd < 2; ++qd) {
          ring_begin(p);
          mma_wide16<2>(acc, chunk_ptr16(p), [&](int s_, half8& xh, half8& xl) { staged(s_, xh, xl, qd * 2); });
          ring_end(p);
        }
      } else if (li == T0) {
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) stage(st, tid, 0, jj, P.lat_trans[b * 16 + 8 * hh + jj]);
        ring_begin(p);
        mma_wide16<1>(acc, chunk_ptr16(p), [&](int s_, half8& xh, half8& xl) { staged(s_, xh, xl, 0); });
        ring_end(p);
      } else if (li == R0) {
        // [ray_unit | PE(ray_unit) | x | light] in natural column order, 78 of 80 slots
#pragma nounroll
        for (int e = 0; e < 40; ++e) {
          const int ks = e >> 3, jj = e & 7, slot = 16 * ks + 8 * hh + jj;
          float v;
          if (slot < 3) {
            v = slot == 0 ? vu[0] : (slot == 1 ? vu[1] : vu[2]);
          } else if (slot < 27) {
            const int qq = slot - 3, c = qq >> 3, sc = (qq >> 2) & 1, l = qq & 3;
            const float vc = c == 0 ? vu[0] : (c == 1 ? vu[1] : vu[2]);
            const float arg = tp::mul_rn(vc, ldexpf(3.14159274101257324f, l));
            v = tp::sincos_sel(arg, sc);
          } else if (slot < 30) {
            v = slot == 27 ? x[0] : (slot == 28 ? x[1] : x[2]);
          } else if (slot < 78) {
            v = P.lat_light[b * 48 + slot - 30];
          } else {
            v = 0.0f;
          }
          stage(st, tid, ks, jj, v);
        }
#pragma unroll
        for (int qd = 0; qd < 3; ++qd) {
          ring_begin(p);
          if (qd < 2) mma_wide16<2>(acc, chunk_ptr16(p), [&](int s_, half8& xh, half8& xl) { staged(s_, xh, xl, qd * 2); });
          else mma_wide16<1>(acc, chunk_ptr16(p), [&](int s_, half8& xh, half8& xl) { staged(s_, xh, xl, 4); });
          ring_end(p);
        }
      }

      // un-scale, bias, ReLU, split into hi + lo fp16: the next layer's B operands.  hi = v truncated to 11
      // significant bits (one AND; exactly representable in fp16 for |v| >= 2^-14, below that the fp16
      // subnormal grid costs < 6e-8 absolute), lo = v - hi is exact in fp32; both are packed with
      // v_cvt_pkrtz_f16_f32 (hi converts exactly, lo keeps 11 more bits).
      const float* bl = bias_lds + (li * 2 + hh) * 128;
#pragma unroll
      for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const f32x4 b0 = *reinterpret_cast<const f32x4*>(bl + t * 16 + k * 8);
          const f32x4 b1 = *reinterpret_cast<const f32x4*>(bl + t * 16 + k * 8 + 4);
          const float bv[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
          half2v hp[4], lp[4];
#pragma unroll
          for (int e = 0; e < 8; e += 2) {
            const float v0 = fmaxf(fmaf(acc[t][k * 8 + e], kInvScale, bv[e]), 0.0f);
            const float v1 = fmaxf(fmaf(acc[t][k * 8 + e + 1], kInvScale, bv[e + 1]), 0.0f);
            amax = fmaxf(amax, fmaxf(v0, v1));
            const float h0 = __uint_as_float(__float_as_uint(v0) & 0xFFFFE000u);
            const float h1 = __uint_as_float(__float_as_uint(v1) & 0xFFFFE000u);
            hp[e >> 1] = __builtin_amdgcn_cvt_pkrtz(h0, h1);
            lp[e >> 1] = __builtin_amdgcn_cvt_pkrtz(v0 - h0, v1 - h1);
          }
          X.hi[t][k] = pack8(hp);
          X.lo[t][k] = pack8(lp);
        }

      if (li == L7) {
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            ws[((t * 2 + k) * 2 + 0) * kThreads] = X.hi[t][k];
            ws[((t * 2 + k) * 2 + 1) * kThreads] = X.lo[t][k];
          }
      }
      if (li == T2) {
        const f32x16 a = part_head16(p, X);
        const float* hb = bias_lds + kHeadBiasOff + 1;
        rgb_t[0] = sigmoid(fmaf(a[0], kInvScale, hb[0]));
        rgb_t[1] = sigmoid(fmaf(a[1], kInvScale, hb[1]));
        rgb_t[2] = sigmoid(fmaf(a[2], kInvScale, hb[2]));
        sig_t = softplus(fmaf(a[3], kInvScale, hb[3]));
        unc = softplus(fmaf(a[0], kInvScale, hb[4]));      // row 4 = register 0 of the upper lane half
      }
      if (li == R2) {
        const f32x16 a = part_head16(p, X);
        const float* hb = bias_lds + kHeadBiasOff + 6;
        rgb_s[0] = sigmoid(fmaf(a[0], kInvScale, hb[0]));
        rgb_s[1] = sigmoid(fmaf(a[1], kInvScale, hb[1]));
        rgb_s[2] = sigmoid(fmaf(a[2], kInvScale, hb[2]));
      }
    }

    if (live) {
      if (hh == 0) {
        float2* o = reinterpret_cast<float2*>(P.rgb + s * 6);
        o[0] = make_float2(rgb_s[0], rgb_t[0]);
        o[1] = make_float2(rgb_s[1], rgb_t[1]);
        o[2] = make_float2(rgb_s[2], rgb_t[2]);
        *reinterpret_cast<float2*>(P.density + s * 2) = make_float2(sig_s, sig_t);
      } else {
        P.uncert[s] = unc;
      }
    }
    if (P.status != nullptr && !(amax < 6.0e4f)) atomicOr(P.status, 1);
  }
}

}  // namespace

// launched by tp_mlp_fwd (mlp_fwd.hip) when args->precision == TP_MLP_F16X3
int tp_launch_mlp_fwd_f16x3(const tp_mlp_fwd_args* a, int grid, hipStream_t stream) {
  Params P;
  P.packed = (const float*)a->packed;
  P.center = a->center; P.ray = a->ray; P.depth = a->depth; P.points = a->points; P.ray_unit = a->ray_unit;
  P.lat_trans = a->lat_trans; P.lat_light = a->lat_light;
  P.B = a->B; P.R = a->R; P.N = a->N;
  P.n_samples = (int64_t)a->B * a->R * a->N;
  P.n_tiles = (P.n_samples + 127) / 128;
  P.rgb = a->rgb; P.density = a->density; P.uncert = a->uncert; P.workspace = (float*)a->workspace; P.status = a->status;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)mlp_fwd_f16x3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
    if (e != hipSuccess) { tp::set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
    attr_set = true;
  }
  hipLaunchKernelGGL(mlp_fwd_f16x3_kernel, dim3(grid), dim3(kThreads), kLdsBytes, stream, P);
  return tp::check_launch("tp_mlp_fwd(f16x3)");
}
