// Microbenchmark (GPU box): cycles per v_mfma_f32_32x32x16_f16 for one wave per SIMD, alone and with the
// ds_read_b128 traffic pattern of the f16x3 MLP kernel.  hipcc --offload-arch=gfx950 -O3 mfma_rate.hip -o mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
using half8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
template <int MODE>
__global__ __launch_bounds__(256, 1) void k(float* out, long long* cyc, int iters) {
  extern __shared__ __attribute__((aligned(16))) _Float16 lds[];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 16384 * 2; i += 256) lds[i] = (_Float16)(0.001f * (i & 63));
  __syncthreads();
  f32x16 acc[8];
  for (int t = 0; t < 8; ++t) acc[t] = f32x16{0};
  half8 xh, xl;
  for (int j = 0; j < 8; ++j) { xh[j] = (_Float16)(0.01f * lane + j); xl[j] = (_Float16)(0.001f * j); }
  const _Float16* l = lds + lane * 8;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      half8 wh, wl;
      if (MODE == 0) { wh = xh; wl = xl; }
      else { wh = *(const half8*)(l + (q * 2 + 0) * 512); wl = *(const half8*)(l + (q * 2 + 1) * 512); }
      const int t = q & 7;
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh, acc[t], 0, 0, 0);
    }
    if (MODE == 2) __syncthreads();
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int t = 0; t < 8; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE> void run(const char* name, int grid) {
  float* out; long long* cyc; hipMalloc(&out, grid * 256 * 4); hipMalloc(&cyc, grid * 8);
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<grid, 256, 65536>>>(out, cyc, 10);
  hipEventRecord(e0);
  k<MODE><<<grid, 256, 65536>>>(out, cyc, iters);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double nm = 48.0 * iters;
  printf("%-28s grid %4d: %.2f ms, %.1f memtime-ticks/MFMA, wall-ns/MFMA %.2f -> %.1f cycles at 2.4GHz, TFLOP/s %.0f\n", name, grid, ms,
         c / nm, ms * 1e6 / nm, ms * 1e6 / nm * 2.4, (double)grid * 4 * nm * 32768 / (ms * 1e-3) / 1e12);
}
int main() {
  hipFuncSetAttribute((const void*)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipFuncSetAttribute((const void*)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipFuncSetAttribute((const void*)k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  for (int grid : {1, 256}) {
    run<0>("mfma only (regs)", grid);
    run<1>("mfma + 2 ds_read_b128 / 3", grid);
    run<2>("same + barrier / 48 mfma", grid);
  }
  return 0;
}
