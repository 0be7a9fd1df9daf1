// microbenchmark: what a pure streaming READ reaches on this chip (the composite kernels read 5x what they write): float4 loads, U in
// flight per lane, plain or nontemporal, by blocks per CU.   hipcc --offload-arch=gfx950 -O3 read_bw.hip -o read_bw && ./read_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int U, bool NT>
__global__ __launch_bounds__(256) void read_kernel(const f32x4* __restrict__ x, size_t n, float* out) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + (U - 1) * stride < n; i += U * stride) {
    f32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(x + i + u * stride) : x[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u];
  }
  for (; i < n; i += stride) acc += x[i];
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = acc[0];
}

template <int U, bool NT>
float run(const f32x4* x, size_t n, float* out, int blocks) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  read_kernel<U, NT><<<blocks, 256>>>(x, n, out);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 10; ++r) read_kernel<U, NT><<<blocks, 256>>>(x, n, out);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / 10;
}

int main() {
  const size_t bytes = 1908326400ull / 16 * 16;     // the composite forward's algorithmic bytes at 480x640x128
  f32x4* x; float* out;
  hipMalloc(&x, bytes); hipMalloc(&out, 16);
  hipMemset(x, 0, bytes);
  const size_t n = bytes / 16;
  for (int bpc : {4, 8, 16, 32}) {
    const int blocks = 256 * bpc;
    printf("blocks/CU %2d:  U=4 %.0f  U=8 %.0f  U=8 nt %.0f  U=16 %.0f  U=16 nt %.0f GB/s\n", bpc,
           bytes / run<4, false>(x, n, out, blocks) / 1e6, bytes / run<8, false>(x, n, out, blocks) / 1e6, bytes / run<8, true>(x, n, out, blocks) / 1e6,
           bytes / run<16, false>(x, n, out, blocks) / 1e6, bytes / run<16, true>(x, n, out, blocks) / 1e6);
  }
  return 0;
}
