// Shared helpers for the gfx950 kernels (host side error plumbing + small device utilities).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/texpose_amd.h"

namespace tp {

void set_error(const char* fmt, ...);

// true the first time `flags` (one bit per device, a function-local static of the caller) sees the current device:
// per-device kernel attributes (large dynamic LDS) must be set on every GPU a process drives, not once per process
inline bool first_use_on_device(unsigned long long& flags) {
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0 || d > 63) return true;
  const bool first = !((flags >> d) & 1ull);
  flags |= 1ull << d;
  return first;
}
inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}

#define TP_REQUIRE(cond, msg)                 \
  do {                                        \
    if (!(cond)) {                            \
      tp::set_error("%s: %s", __func__, msg); \
      return -1;                              \
    }                                         \
  } while (0)

constexpr int kWave = 64;

// round-to-nearest single ops that the compiler may not contract into FMAs: the positional
// encoding amplifies 1-ulp differences of the sample position by up to 2^9*pi, so the point /
// depth arithmetic follows the reference's op-by-op rounding ("numerics").
__device__ __forceinline__ float mul_rn(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float add_rn(float a, float b) { return __fadd_rn(a, b); }
__device__ __forceinline__ float sub_rn(float a, float b) { return __fsub_rn(a, b); }
__device__ __forceinline__ float div_rn(float a, float b) { return __fdiv_rn(a, b); }
__device__ __forceinline__ float fma_rn(float a, float b, float c) { return __fmaf_rn(a, b, c); }

// sin / cos of the fp32 value `a`, accurate to ~1 ulp (the positional encoding evaluates x * 2^l * pi with |x| of a
// few units, i.e. arguments up to ~1e4 rad).  Branch-free, no libm: the reduction by pi/2 runs in fp64 (two-term
// split, exact for |a| far beyond any encodable coordinate; Inf/NaN give NaN like sinf/cosf), then the Cephes
// single-precision minimax polynomials on [-pi/4, pi/4].  ocml's sinf/cosf cost ~4x more instructions and were 20 %
// of the f16x3 forward; their inlined slow paths also cost registers next to the 256-register accumulator sets.
struct Reduced { float r, sp, cp; int q; };
__device__ __forceinline__ Reduced reduce_pio2(float a) {
  const double ad = (double)a;
  const double kd = rint(ad * 0.63661977236758138);
  double rd = __fma_rn(-kd, 1.5707963267948966, ad);
  rd = __fma_rn(-kd, 6.123233995736766e-17, rd);
  Reduced o;
  o.r = (float)rd;
  o.q = (int)(long long)kd;            // only the two low bits matter
  const float r = o.r, z = r * r;
  o.sp = __fmaf_rn(__fmaf_rn(__fmaf_rn(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f) * z, r, r);
  o.cp = __fmaf_rn(__fmaf_rn(__fmaf_rn(2.443315711809948e-5f, z, -1.388731625493765e-3f), z,
                              4.166664568298827e-2f) * z, z, __fmaf_rn(-0.5f, z, 1.0f));
  return o;
}
// sin(a) (quad = 0) or cos(a) (quad = 1): cos is the same code one quadrant later
__device__ __forceinline__ float sincos_sel(float a, int quad) {
  const Reduced o = reduce_pio2(a);
  const int q = o.q + quad;
  const float v = (q & 1) ? o.cp : o.sp;
  return (q & 2) ? -v : v;
}
// both from one reduction: bit-identical to sincos_sel(a, 0) / sincos_sel(a, 1)
__device__ __forceinline__ void sincos_both(float a, float& s, float& c) {
  const Reduced o = reduce_pio2(a);
  const float vs = (o.q & 1) ? o.cp : o.sp;
  const float vc = (o.q & 1) ? o.sp : o.cp;
  s = (o.q & 2) ? -vs : vs;
  c = ((o.q + 1) & 2) ? -vc : vc;
}

// ---- sin / cos of a * 2^l for l = 0 .. L-1 from ONE range reduction (the positional encoding of the f16x3 forward).
// The reference encodes fl32(x * fl32(2^l * pi_f32)) (layers/nerf_static_transient_light.py:217-234: `x[..., None] * freq`), and a
// power of two scales exactly: that argument IS 2^l * a with a = fl32(x * pi_f32).  So a is reduced once in fp64 (a = k pi/2 + r,
// |r| <= pi/4, two-term pi/2), sin r / cos r come from the fdlibm double-precision kernels (error < 2^-57), the quadrant is
// applied, and every further octave is an angle doubling in fp64 -- z -> z^2 on the unit circle doubles the absolute error per
// step, 2^9 * 2e-16 = 1e-13 after nine doublings, where ten independent reductions of a_l each carry |a_l| * 2^-53 ~ 1e-12 in r.
// The fp32 roundings of the results are within 1 ulp of the correctly rounded sin / cos of the reference's fp32 argument whenever
// |value| > 2e-6 (absolute error 1e-13 below that).  ~3.5x fewer instructions than per-octave reductions (DESIGN section 4, K2').
struct SinCos64 { double s, c; };
__device__ __forceinline__ SinCos64 sincos_f64(float a) {
  const double ad = (double)a;
  const double kd = rint(ad * 0.63661977236758138);
  double r = __fma_rn(-kd, 1.5707963267948966, ad);
  r = __fma_rn(-kd, 6.123233995736766e-17, r);
  const double z = r * r;
  // fdlibm k_sin.c / k_cos.c minimax coefficients on [-pi/4, pi/4]
  double ps = __fma_rn(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
  ps = __fma_rn(z, ps, 2.75573137070700676789e-06);
  ps = __fma_rn(z, ps, -1.98412698298579493134e-04);
  ps = __fma_rn(z, ps, 8.33333333332248946124e-03);
  ps = __fma_rn(z, ps, -1.66666666666666324348e-01);
  const double sr = __fma_rn(r * z, ps, r);
  double pc = __fma_rn(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
  pc = __fma_rn(z, pc, -2.75573143513906633035e-07);
  pc = __fma_rn(z, pc, 2.48015872894767294178e-05);
  pc = __fma_rn(z, pc, -1.38888888888741095749e-03);
  pc = __fma_rn(z, pc, 4.16666666666666019037e-02);
  const double cr = __fma_rn(z * z, pc, __fma_rn(-0.5, z, 1.0));
  const int q = (int)kd;                 // only the two low bits matter (|a| < 2^30 for any encodable coordinate)
  SinCos64 o;
  const double v_s = (q & 1) ? cr : sr, v_c = (q & 1) ? sr : cr;
  o.s = (q & 2) ? -v_s : v_s;
  o.c = ((q + 1) & 2) ? -v_c : v_c;
  return o;
}
__device__ __forceinline__ void sincos_double(SinCos64& v) {
  const double t = v.s + v.s;
  const double s2 = t * v.c;
  v.c = __fma_rn(-t, v.s, 1.0);
  v.s = s2;
}

// ---- cross-lane data movement on the vector ALU (DPP) instead of the LDS crossbar.  `__shfl*` compiles to ds_bpermute_b32: one LDS
// instruction per step, and a wave scan is six of them -- the composite kernels issued 108 per 64 samples and were bound by the LDS
// pipe, not by HBM (profiles/r5).  gfx9 DPP controls: row_shr:n 0x110+n, row_shl:n 0x100+n (within rows of 16 lanes), wave_shl:1 0x130,
// wave_shr:1 0x138 (whole wavefront), row_bcast:15 0x142 / row_bcast:31 0x143 (lane 15 of a row -> the next row; lane 31 -> rows 2, 3).
// Lanes without a source (row / wave ends, rows outside ROW_MASK) receive `old`.
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp(float old, float src) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float lane_value(float v, int lane) {          // (a wave-uniform value: lands in a scalar register)
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
// inclusive prefix sum over the 64 lanes: Hillis-Steele inside each row of 16, then the row totals ripple up
__device__ __forceinline__ float wave_scan_dpp(float v) {
  v += dpp<0x111>(0.f, v);
  v += dpp<0x112>(0.f, v);
  v += dpp<0x114>(0.f, v);
  v += dpp<0x118>(0.f, v);
  v += dpp<0x142, 0xA>(0.f, v);
  v += dpp<0x143, 0xC>(0.f, v);
  return v;
}
// inclusive SUFFIX sum (lane l: sum of lanes l .. 63): the same inside the rows with left shifts; DPP has no downward broadcast, so the
// totals of the rows above come through scalar registers
__device__ __forceinline__ float wave_rev_scan_dpp(float v, int lane) {
  v += dpp<0x101>(0.f, v);
  v += dpp<0x102>(0.f, v);
  v += dpp<0x104>(0.f, v);
  v += dpp<0x108>(0.f, v);
  const float t1 = lane_value(v, 16), t2 = lane_value(v, 32), t3 = lane_value(v, 48);
  const int row = lane >> 4;
  const float above = ((row < 3 ? t3 : 0.f) + (row < 2 ? t2 : 0.f)) + (row < 1 ? t1 : 0.f);
  return v + above;
}
// the same scan / total for K independent values at once, step-major: consecutive DPP instructions then never depend on each other
// (a DPP read of a register a vector instruction has just written costs two wait states: value-major code was one s_nop per step)
template <int CTRL, int ROW_MASK, int K>
__device__ __forceinline__ void dpp_add_step(float (&v)[K]) {
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] += dpp<CTRL, ROW_MASK>(0.f, v[k]);
}
template <int K>
__device__ __forceinline__ void wave_scan_dpp(float (&v)[K]) {
  dpp_add_step<0x111, 0xf>(v);
  dpp_add_step<0x112, 0xf>(v);
  dpp_add_step<0x114, 0xf>(v);
  dpp_add_step<0x118, 0xf>(v);
  dpp_add_step<0x142, 0xA>(v);
  dpp_add_step<0x143, 0xC>(v);
}
template <int K>
__device__ __forceinline__ void wave_rev_scan_dpp(float (&v)[K], int lane) {
  dpp_add_step<0x101, 0xf>(v);
  dpp_add_step<0x102, 0xf>(v);
  dpp_add_step<0x104, 0xf>(v);
  dpp_add_step<0x108, 0xf>(v);
  const int row = lane >> 4;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const float t1 = lane_value(v[k], 16), t2 = lane_value(v[k], 32), t3 = lane_value(v[k], 48);
    v[k] += ((row < 3 ? t3 : 0.f) + (row < 2 ? t2 : 0.f)) + (row < 1 ? t1 : 0.f);
  }
}
// ---- the wave totals of FOURTEEN values at once (the composite's per-ray sums) as a butterfly that halves the number of live
// registers while it halves the lane span: gfx950's v_permlane32_swap / v_permlane16_swap exchange the upper half (the odd rows) of
// one register with the lower half (the even rows) of another, so one add folds TWO values by 32 (16) lanes.  7 + 4 swap-adds, then
// four registers x four in-row steps: 38 vector instructions instead of 14 x 6 steps with their masked broadcasts (~140).
// out[k] = sum over the 64 lanes of v[k], the same in every lane (wave-uniform).
// (Inline assembly, not __builtin_amdgcn_permlane{32,16}_swap: ROCm 7.2's hipcc folds `r[0] + r[1]` of the builtin's result pair into
// `r[0] + r[0]` -- "v_permlane32_swap v1, v2; v_add_f32 v1, v1, v1" -- so sums came out doubled per step.  The s_nops cover the
// instruction's read-after-VALU-write wait states, which the compiler's hazard recogniser cannot see inside the asm.)
struct SwapPair { float a, b; };
__device__ __forceinline__ SwapPair swap32(float a, float b) {     // a' = [a.lo | b.lo], b' = [a.hi | b.hi]
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  return {a, b};
}
__device__ __forceinline__ SwapPair swap16(float a, float b) {     // rows: a' = [a0, b0, a2, b2], b' = [a1, b1, a3, b3]
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  return {a, b};
}
__device__ __forceinline__ void wave_totals14(float (&v)[14]) {
  float r[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) {                 // lanes < 32: v[k] folded by 32; lanes >= 32: v[k + 7] folded by 32
    const SwapPair p = swap32(v[k], v[k + 7]);
    r[k] = p.a + p.b;
  }
  float q[4];
#pragma unroll
  for (int k = 0; k < 3; ++k) {                 // rows: v[2k], v[2k+1], v[2k+7], v[2k+8], each folded to 16 lanes
    const SwapPair p = swap16(r[2 * k], r[2 * k + 1]);
    q[k] = p.a + p.b;
  }
  {
    const SwapPair p = swap16(r[6], r[6]);      // rows: v[6], v[6], v[13], v[13]
    q[3] = p.a + p.b;
  }
  dpp_add_step<0x111, 0xf>(q);                  // lane 15 of every row: the row's total
  dpp_add_step<0x112, 0xf>(q);
  dpp_add_step<0x114, 0xf>(q);
  dpp_add_step<0x118, 0xf>(q);
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    v[2 * k] = lane_value(q[k], 15); v[2 * k + 1] = lane_value(q[k], 31);
    v[2 * k + 7] = lane_value(q[k], 47); v[2 * k + 8] = lane_value(q[k], 63);
  }
  v[6] = lane_value(q[3], 15);
  v[13] = lane_value(q[3], 47);
}
__device__ __forceinline__ float wave_shr1(float v) { return dpp<0x138>(0.f, v); }     // lane l <- lane l-1, lane 0 <- 0
__device__ __forceinline__ float wave_shl1(float v) { return dpp<0x130>(0.f, v); }     // lane l <- lane l+1, lane 63 <- 0
__device__ __forceinline__ float wave_total_dpp(float v) { return lane_value(wave_scan_dpp(v), 63); }

// Philox4x32-10 (Salmon et al., SC'11); the stream layouts of its users are documented in oracle.philox_uniform (ray-gen jitter)
// and at tp_patch_coords (patch draws).
__device__ __forceinline__ uint4 philox4x32_10(uint4 c, uint2 k) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
    c = make_uint4(hi1 ^ c.y ^ k.x, lo1, hi0 ^ c.w ^ k.y, lo0);
    k.x += 0x9E3779B9u; k.y += 0xBB67AE85u;
  }
  return c;
}
__device__ __forceinline__ float u01(uint32_t w) { return (float)(w >> 8) * 5.9604644775390625e-08f; }

}  // namespace tp
