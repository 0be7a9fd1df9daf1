// Microbenchmark (GPU box): what a record store costs a wave that owns its SIMD, inside the instruction mix of the f16x3 MLP
// blocks (gen_store_cost.py).  Build + run:
//   python3 gen_store_cost.py > store_cost.inc.h && hipcc --offload-arch=gfx950 -O3 store_cost.hip -o store_cost && ./store_cost
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include "store_cost.inc.h"

#define CLOBBERS                                                                                                        \
  "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167", "v168", "v169", "v170", "v171", "v172", "v173", "v174", \
      "v175", "v176", "v177", "v178", "v179", "v180", "v181", "v182", "v183", "v184", "v185", "v186", "v187", "v188",   \
      "v189", "v190", "v191", "v192", "v193", "v194", "v195", "v196", "v197", "v198", "v199", "v200", "v201", "v202",   \
      "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "a0", "a255", "memory"

using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;

template <int MODE>
__global__ __launch_bounds__(256, 1) void k(const float* weights, float* sink, long long* cyc, int iters, int n_chunks) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int i = threadIdx.x; i < 24576; i += 256) lds[i] = 0.001f * (i & 127);
  __syncthreads();
  const unsigned rd = lane * 16, off = lane * 16;
  asm volatile("s_mov_b32 m0, %0" ::"s"(65536u + wave * 8192u));
  long long t0 = __builtin_amdgcn_s_memtime();
  int chunk = (blockIdx.x * 7) % n_chunks;
  // every wave streams through its own 4 MiB window of the sink (4 KiB per chunk)
  float* win = sink + ((size_t)blockIdx.x * 4 + wave) * (1 << 20);
  const unsigned soff = lane * 16;
  for (int it = 0; it < iters; ++it) {
    const float* base = weights + (size_t)chunk * 8192 + wave * 2048;
    float* sb = win + (size_t)(it & 1023) * 1024;
    const uint64_t a = (uint64_t)(uintptr_t)sb;
    u32x4 rsrc = {(unsigned)a, (unsigned)(a >> 32) & 0xFFFFu, 0xFFFFFFFFu, 0x00020000u};
    float* vaddr = sb + lane * 4;
#define RUN(TXT)                                                                                                       \
  asm volatile(TXT ::[rd] "v"(rd), [off] "v"(off), [base] "s"(base), [soff] "v"(soff), [sbase] "s"(sb), [rsrc] "s"(rsrc), \
               [vaddr] "v"(vaddr)                                                                                      \
               : CLOBBERS)
    if (MODE == 0) RUN(CHUNK_NONE);
    if (MODE == 1) RUN(CHUNK_V_NT_1);
    if (MODE == 2) RUN(CHUNK_V_NT_2);
    if (MODE == 3) RUN(CHUNK_V_NT_4);
    if (MODE == 4) RUN(CHUNK_V_NT_8);
    if (MODE == 5) RUN(CHUNK_A_NT_4);
    if (MODE == 6) RUN(CHUNK_V_PLAIN_4);
    if (MODE == 7) RUN(CHUNK_V_X2_4);
    if (MODE == 8) RUN(CHUNK_V_X2_8);
    if (MODE == 9) RUN(CHUNK_BUF_NT_4);
    if (MODE == 10) RUN(CHUNK_VADDR_4);
    chunk = chunk + 1 == n_chunks ? 0 : chunk + 1;
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)");
  long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE> void run(const char* name, int grid, const float* w, float* sink, int n_chunks) {
  long long* cyc; hipMalloc(&cyc, grid * 8);
  const int iters = 20000;
  hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<grid, 256, 98304>>>(w, sink, cyc, 100, n_chunks);
  hipEventRecord(e0);
  k<MODE><<<grid, 256, 98304>>>(w, sink, cyc, iters, n_chunks);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long c; hipMemcpy(&c, cyc + grid / 2, 8, hipMemcpyDeviceToHost);
  printf("%-44s grid %4d: %8.2f ms  %7.1f cycles/chunk (48 MFMAs: 1536 ideal)  %.2f GHz\n", name, grid, ms, (double)c / iters,
         (double)c / (ms * 1e6));
  hipFree(cyc);
}

int main() {
  const int n_chunks = 115;
  float* w; hipMalloc(&w, (size_t)n_chunks * 32768 + 65536);
  hipMemset(w, 0x3c, (size_t)n_chunks * 32768 + 65536);
  float* sink; hipMalloc(&sink, (size_t)256 * 4 * (1 << 20) * 4);     // 4 GiB: 4 MiB per wave
  for (int grid : {1, 256}) {
    run<0>("DMA only", grid, w, sink, n_chunks);
    run<1>("+ 1 store  dwordx4 nt (VGPR data)", grid, w, sink, n_chunks);
    run<2>("+ 2 stores dwordx4 nt (VGPR data)", grid, w, sink, n_chunks);
    run<3>("+ 4 stores dwordx4 nt (VGPR data)", grid, w, sink, n_chunks);
    run<4>("+ 8 stores dwordx4 nt (VGPR data)", grid, w, sink, n_chunks);
    run<5>("+ 4 stores dwordx4 nt (AGPR data)", grid, w, sink, n_chunks);
    run<6>("+ 4 stores dwordx4 plain", grid, w, sink, n_chunks);
    run<7>("+ 4 stores dwordx2 nt", grid, w, sink, n_chunks);
    run<8>("+ 8 stores dwordx2 nt", grid, w, sink, n_chunks);
    run<9>("+ 4 buffer_store_dwordx4 nt (offen)", grid, w, sink, n_chunks);
    run<10>("+ 4 stores dwordx4 nt (64-bit vaddr)", grid, w, sink, n_chunks);
  }
  return 0;
}
