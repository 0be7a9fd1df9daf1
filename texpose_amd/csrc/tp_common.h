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
