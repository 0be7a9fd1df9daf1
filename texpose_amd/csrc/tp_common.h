// Shared helpers for the gfx950 kernels (host side error plumbing + small device utilities).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/texpose_amd.h"

namespace tp {

void set_error(const char* fmt, ...);

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

// sin(a) (quad = 0) or cos(a) (quad = 1) of the fp32 value `a`, accurate to ~1 ulp for |a| < 2^17 (the positional
// encoding evaluates x * 2^l * pi with |x| of a few units, i.e. arguments up to ~1e4 rad).  One branch-free path
// for both functions: Cody-Waite reduction by pi/2 with a 3-term split and FMAs (the first product is exact inside
// the FMA), then the Cephes single-precision minimax polynomials on [-pi/4, pi/4]; cos is the same code one
// quadrant later, so a wave whose lanes mix sin and cos entries does not evaluate both.  ocml's sinf/cosf cost
// ~4x more instructions and were 20 % of the f16x3 forward.
__device__ __forceinline__ float sincos_sel(float a, int quad) {
  if (__builtin_expect(!(fabsf(a) < 131072.0f), 0)) return quad ? cosf(a) : sinf(a);
  const float kf = rintf(a * 0.6366197466850281f);
  float r = __fmaf_rn(-kf, 1.5707963705062866f, a);
  r = __fmaf_rn(-kf, -4.371138828673793e-08f, r);
  r = __fmaf_rn(-kf, -1.7151245100058819e-15f, r);
  const int q = (int)kf + quad;
  const float z = r * r;
  const float sp = __fmaf_rn(__fmaf_rn(__fmaf_rn(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f) * z, r, r);
  const float cp = __fmaf_rn(__fmaf_rn(__fmaf_rn(2.443315711809948e-5f, z, -1.388731625493765e-3f), z,
                                       4.166664568298827e-2f) * z, z, __fmaf_rn(-0.5f, z, 1.0f));
  const float v = (q & 1) ? cp : sp;
  return (q & 2) ? -v : v;
}

}  // namespace tp
