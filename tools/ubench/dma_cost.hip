// Microbenchmark (GPU box): cost of a weight-prefetch piece to a wave that owns its SIMD, in the instruction mix of the
// f16x3 MLP forward (see gen_dma_cost.py).  Build + run:
//   python3 gen_dma_cost.py > dma_cost.inc.h && hipcc --offload-arch=gfx950 -O3 dma_cost.hip -o dma_cost && ./dma_cost
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "dma_cost.inc.h"
using half8 = __attribute__((ext_vector_type(8))) _Float16;

#define CLOBBERS                                                                                                        \
  "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167", "v168", "v169", "v170", "v171", "v172", "v173", "v174", \
      "v175", "v176", "v177", "v178", "v179", "v180", "v181", "v182", "v183", "v184", "v185", "v186", "v187", "v188",   \
      "v189", "v190", "v191", "v192", "v193", "v194", "v195", "v196", "v197", "v198", "v199", "v200", "v201", "v202",   \
      "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", "v216",   \
      "v217", "v218", "v219", "v220", "v221", "v222", "v223", "a0", "a127", "memory"

template <int MODE>
__global__ __launch_bounds__(256, 1) void k(const float* weights, long long* cyc, int iters, int n_chunks) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int i = threadIdx.x; i < 24576; i += 256) lds[i] = 0.001f * (i & 127);
  __syncthreads();
  const unsigned rd = lane * 16, off = lane * 16;
  const unsigned wr = 65536 + wave * 8192 + lane * 16;          // VGPR-staged variant writes into the third slot
  asm volatile("s_mov_b32 m0, %0" ::"s"(65536u + wave * 8192u));
  long long t0 = __builtin_amdgcn_s_memtime();
  int chunk = (blockIdx.x * 7) % n_chunks;
  for (int it = 0; it < iters; ++it) {
    const float* base = weights + (size_t)chunk * 8192 + wave * 2048;
    if (MODE == 0) asm volatile(CHUNK_NONE ::[rd] "v"(rd), [off] "v"(off), [base] "s"(base), [wr] "v"(wr) : CLOBBERS);
    if (MODE == 1) asm volatile(CHUNK_DMA ::[rd] "v"(rd), [off] "v"(off), [base] "s"(base), [wr] "v"(wr) : CLOBBERS);
    if (MODE == 2) asm volatile(CHUNK_VGPR ::[rd] "v"(rd), [off] "v"(off), [base] "s"(base), [wr] "v"(wr) : CLOBBERS);
    if (MODE == 3) asm volatile(CHUNK_DMA2 ::[rd] "v"(rd), [off] "v"(off), [base] "s"(base), [wr] "v"(wr) : CLOBBERS);
    chunk = chunk + 1 == n_chunks ? 0 : chunk + 1;
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)");
  long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE> void run(const char* name, int grid, const float* w, int n_chunks) {
  long long* cyc; hipMalloc(&cyc, grid * 8);
  const int iters = 20000;
  hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<grid, 256, 98304>>>(w, cyc, 100, n_chunks);
  hipEventRecord(e0);
  k<MODE><<<grid, 256, 98304>>>(w, cyc, iters, n_chunks);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long c; hipMemcpy(&c, cyc + grid / 2, 8, hipMemcpyDeviceToHost);
  printf("%-34s grid %4d: %8.2f ms  %7.1f cycles/chunk (48 MFMAs: 1536 ideal)  %.2f GHz\n", name, grid, ms, (double)c / iters,
         (double)c / (ms * 1e6));
  hipFree(cyc);
}

int main() {
  const int n_chunks = 115;
  float* w; hipMalloc(&w, (size_t)n_chunks * 32768 + 65536);
  hipMemset(w, 0x3c, (size_t)n_chunks * 32768 + 65536);
  for (int grid : {1, 256}) {
    run<0>("no prefetch", grid, w, n_chunks);
    run<1>("LDS-DMA, 1 piece / group", grid, w, n_chunks);
    run<2>("global_load -> VGPR -> ds_write_b128", grid, w, n_chunks);
    run<3>("LDS-DMA, 2 pieces / 2 groups", grid, w, n_chunks);
  }
  return 0;
}
