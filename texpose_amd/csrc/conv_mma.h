// Shared machinery of the implicit-GEMM convolution kernels (csrc/patch_conv.hip: K11 / K12; csrc/feat_chain.hip: K18): the fp32 matrix
// instruction, the cross-wavefront / cross-workgroup reduction of accumulator tiles with a fixed summation order, the split of a k
// range over slices and wavefronts, and the host-side tiling plan.  Included into each translation unit's anonymous namespace user.
#pragma once
#include "tp_common.h"
#include <stdlib.h>

namespace {
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

__device__ __forceinline__ f32x16 mfma(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }

struct ConvP {
  const float* x;      // F, G: input [N,C,H,W];        D: unused
  const float* w;      // F, D: weight [Co,C,4,4];      G: unused
  const float* gy;     // D, G: output-side tensor [N,Co,OH,OW]
  float* out;          // F: y [N,Co,OH,OW]; D: gx [N,C,H,W]; G: gW [Co,C,4,4]
  float* ws;           // split-K partial sums
  unsigned* cnt;       // one arrival counter per tile (zero between launches)
  int N, C, H, W, Co, OH, OW;
  int lw, low;         // log2(W), log2(OW)
  int lp;              // log2(OH * OW)
  int S;               // workgroups per tile
  int tiles_n;         // column tiles
  // D, optional (in_gx != NULL): the InstanceNorm + LeakyReLU backward of the stage in front of this convolution on the data gradient
  // (8x8 maps), csrc/patch_conv.hip conv4s2_dgrad_kernel<true>
  const float* in_xhat; const float* in_rstd; const float* in_addend; float* in_gx; float in_slope; int skip_out;
  // F + InstanceNorm, optional (x_copy != NULL): the launch also leaves a copy of its input x there (the workgroups share the work) --
  // the discriminator step's private copy of the render's patch stacks without a launch of its own
  float* x_copy;
};

// Sum NT accumulator tiles over the 4 wavefronts of the workgroup and over the S workgroups of the tile.  True in the one
// workgroup that ends up with the totals: thread (w, lane) then holds registers 4w..4w+3 of each tile, i.e. tile rows
// 8w + 4(lane>>5) + 0..3 of column lane & 31.
#ifdef TP_REDUCE_LDS_8K                            // (A/B build `make reduce8k`: eight registers per round, 8 KB of LDS per workgroup)
constexpr int kReduceRegs = 8;
#else
constexpr int kReduceRegs = 16;
#endif
constexpr int kReduceLdsFloats = kReduceRegs * 4 * 64;       // the LDS floats reduce_tiles needs, whatever NT

template <int NT>
__device__ __forceinline__ bool reduce_tiles(const f32x16 (&acc)[NT], float (&out)[NT][4], float* lds, const ConvP& p, int tile, int s) {
  const int t = threadIdx.x, w = t >> 6, lane = t & 63;
  // The four wavefronts' partial tiles meet in LDS, kReduceRegs accumulator registers per round.  With 8 (TP_REDUCE_LDS_8K) a
  // workgroup needs 8 KB instead of 16 KB per tile and fits into the LDS the training MLP's backward kernels leave free on a CU (40 KB
  // beside the data gradient, 16 KB beside the weight gradient): measured on one box, the discriminator chain then runs BESIDE the
  // render's backward, slows it by 20 us and finishes no earlier (its tail kernels, 73-81 KB of LDS, still wait) -- 0.5 % slower per
  // iteration, so 16 stays (profiles/r5).  Same operands and order of additions either way: bit-identical results.
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int half = 0; half < 16 / kReduceRegs; ++half) {
      if (nt + half > 0) __syncthreads();                      // (the previous round's reads are done)
#pragma unroll
      for (int r = 0; r < kReduceRegs; ++r) lds[(r * 4 + w) * 64 + lane] = acc[nt][kReduceRegs * half + r];
      __syncthreads();
      if (kReduceRegs == 16 || (w >> 1) == half)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float* q = lds + (((4 * w + i) & (kReduceRegs - 1)) * 4) * 64 + lane;
          out[nt][i] = ((q[0] + q[64]) + q[128]) + q[192];
        }
    }
  if (p.S == 1) return true;
  // Cross-workgroup hand-over WITHOUT device-scope fences: on gfx950 a fence is buffer_wbl2 + buffer_inv of the XCD's whole
  // L2 (it threw the other workgroups' weight lines away; an 18-MFLOP convolution took 100 us).  Instead every access to
  // shared words is itself device-scope (sc1: partial sums are written through to memory and read past the L2, the counter is
  // a device-scope atomic), and a workgroup counts itself in only after all of its stores have been acknowledged (vmcnt 0).
  // This is NOT the HIP / LLVM memory model (relaxed accesses carry no release / acquire ordering there); it is the gfx950
  // hardware contract of MI355X_MICROARCH.md, "Workgroup dispatch ... inter-workgroup visibility", valid-forms table row 1:
  // every handed-off byte stored sc1 and every storing wave drained (s_waitcnt vmcnt(0)) BEFORE the workgroup barrier behind
  // which ONE lane adds to an agent-scope counter; the last arriver is told by the value its add returned; the other waves
  // load (sc1, to registers) only after a barrier that lane then joins.  The file refuses to build for any other target.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "reduce_tiles relies on the gfx950 sc1 write-through hand-over (see the comment above): re-derive for another target"
#endif
  float* mine = p.ws + ((size_t)tile * p.S + s) * (NT * 4 * 256);
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int i = 0; i < 4; ++i) __hip_atomic_store(mine + (nt * 4 + i) * 256 + t, out[nt][i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this thread's stores are complete
  __syncthreads();
  __shared__ int last;
  if (t == 0) last = (__hip_atomic_fetch_add(&p.cnt[tile], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(p.S - 1));
  __syncthreads();
  if (!last) return false;
  const float* all = p.ws + (size_t)tile * p.S * (NT * 4 * 256);
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int i = 0; i < 4; ++i) out[nt][i] = 0.f;
  // kFly slices' loads in flight, added in slice order
  constexpr int kFly = 4;                           // (8 / 16 in flight measured: no difference, profiles/r5/02)
  for (int k0 = 0; k0 < p.S; k0 += kFly) {
    float part[kFly][NT * 4];
#pragma unroll
    for (int u = 0; u < kFly; ++u)
#pragma unroll
      for (int e = 0; e < NT * 4; ++e)
        part[u][e] = __hip_atomic_load(all + ((size_t)min(k0 + u, p.S - 1) * NT * 4 + e) * 256 + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int u = 0; u < kFly; ++u)
      if (k0 + u < p.S)
#pragma unroll
        for (int e = 0; e < NT * 4; ++e) out[e >> 2][e & 3] += part[u][e];
  }
  if (t == 0) __hip_atomic_store(&p.cnt[tile], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
  return true;
}

// this wavefront's part [b, e) of the k units [0, n) of slice s of S, the slice again split over the 4 wavefronts
__device__ __forceinline__ void k_range(int n, int S, int s, int w, int& b, int& e) {
  const int per_s = (n + S - 1) / S;
  const int s0 = min(n, s * per_s), s1 = min(n, s0 + per_s);
  const int per_w = (s1 - s0 + 3) >> 2;
  b = min(s1, s0 + w * per_w);
  e = min(s1, b + per_w);
}

inline int ilog2(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return (1 << l) == v ? l : -1;
}

// workspace floats per (tile, slice): NT * 4 * 256
struct Plan { int tiles_m, tiles_n, S, nt; size_t ws_floats; };

// TP_CONV_TARGET_WGS (diagnostic, read once): overrides the number of workgroups a launch spreads its split-K slices over (default:
// one per CU).  These kernels run three chains side by side in the captured training step; fewer, fatter workgroups trade one more
// load round trip per wavefront against less crowding of every CU (tools/README).
inline int conv_target_wgs(int wanted) {
  static const int forced = [] { const char* e = getenv("TP_CONV_TARGET_WGS"); return e ? atoi(e) : 0; }();
  return forced > 0 ? forced : wanted;
}

inline Plan plan(int rows, int cols, int k_units, int nt, int target_wgs, int min_units_per_wave) {
  Plan q;
  target_wgs = conv_target_wgs(target_wgs);
  q.tiles_m = (rows + 31) / 32;
  q.tiles_n = (cols + 31) / 32;
  q.nt = nt;
  const int tiles = q.tiles_m * q.tiles_n;
  int S = tiles >= target_wgs ? 1 : (target_wgs + tiles - 1) / tiles;
  const int max_s = (k_units + 4 * min_units_per_wave - 1) / (4 * min_units_per_wave);
  if (S > max_s) S = max_s;
  if (S < 1) S = 1;
  q.S = S;
  q.ws_floats = S > 1 ? (size_t)tiles * S * nt * 4 * 256 : 0;
  return q;
}
}  // namespace
