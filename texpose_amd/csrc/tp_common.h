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

}  // namespace tp
