// Device helpers shared by the fused MLP forward (mlp_fwd.hip) and the dgrad kernel (mlp_bwd.hip):
// the double-buffered LDS-DMA weight-chunk pipeline and the software-pipelined MFMA k-step loop.
#pragma once
#include "tp_common.h"
#include "mlp_layout.h"

namespace tp_mma {
using namespace tp_layout;

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int kThreads = 256;

#define AS1(p) ((const __attribute__((address_space(1))) void*)(p))
#define AS3(p) ((__attribute__((address_space(3))) void*)(p))

__device__ __forceinline__ f32x16 mfma(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

struct Pipe {
  const float* stream;  // packed chunks (global)
  float* lds;           // two chunk buffers
  int chunk;            // chunk resident in buffer `buf`
  int buf;
  int wave, lane;
#ifdef TP_TRACE
  long long tr[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // diagnostic build only: cycles per kernel section
#endif
};

__device__ __forceinline__ void dma_chunk(const Pipe& p, int chunk, int buf) {
  // wave-uniform base (SGPR pair) + 32-bit lane offset: the saddr form of global_load_lds, no 64-bit VALU address math;
  // the LDS destination is wave-uniform as well (p.wave must be a readfirstlane value) so M0 is set by SALU
  const float* base = p.stream + (size_t)chunk * kChunkFloats + p.wave * 2048;
  float* dst = p.lds + buf * kChunkFloats + p.wave * 2048;
  const unsigned lane_off = (unsigned)p.lane * 16u;
  // the instruction's immediate offset advances BOTH addresses: one address pair + one M0 per four 1 KiB pieces
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const char* src = reinterpret_cast<const char*>(base + g * 1024) + lane_off;
    __builtin_amdgcn_global_load_lds(AS1(src), AS3(dst + g * 1024), 16, 0, 0);
    __builtin_amdgcn_global_load_lds(AS1(src), AS3(dst + g * 1024), 16, 1024, 0);
    __builtin_amdgcn_global_load_lds(AS1(src), AS3(dst + g * 1024), 16, 2048, 0);
    __builtin_amdgcn_global_load_lds(AS1(src), AS3(dst + g * 1024), 16, 3072, 0);
  }
}

// prefetch the next chunk of the stream (wrapping to the next tile's first chunk)
__device__ __forceinline__ void chunk_begin(Pipe& p, int n_chunks) {
  int nxt = p.chunk + 1;
  if (nxt == n_chunks) nxt = 0;
  dma_chunk(p, nxt, p.buf ^ 1);
}
// all waves are done with the current buffer and the prefetch has landed (syncthreads drains vmcnt)
__device__ __forceinline__ void chunk_end(Pipe& p, int n_chunks) {
  __syncthreads();
  p.chunk = (p.chunk + 1 == n_chunks) ? 0 : p.chunk + 1;
  p.buf ^= 1;
}
__device__ __forceinline__ const float* chunk_ptr(const Pipe& p) { return p.lds + p.buf * kChunkFloats + p.lane * 4; }

// KS k-steps of an 8-tile (256-output) layer; B operand of k-step s is b(s).
// The A fragments (and an LDS-resident B operand) of k-step s+1 are fetched before the 8 MFMAs of
// k-step s; sched_group_barrier pins that order (hipcc otherwise sinks each ds_read to just before
// its first use and stalls one wave per SIMD on lgkmcnt(0) every 4 MFMAs).
template <int KS, int NDS, class BFn>
__device__ __forceinline__ void mma_wide(f32x16 (&acc)[8], const float* l, BFn b) {
  f32x4 a0 = *reinterpret_cast<const f32x4*>(l);
  f32x4 a1 = *reinterpret_cast<const f32x4*>(l + 256);
  float bv = b(0);
  __builtin_amdgcn_sched_group_barrier(0x100, NDS, 0);
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    f32x4 n0 = a0, n1 = a1;
    float nb = bv;
    if (s + 1 < KS) {
      n0 = *reinterpret_cast<const f32x4*>(l + (s * 2 + 2) * 256);
      n1 = *reinterpret_cast<const f32x4*>(l + (s * 2 + 3) * 256);
      nb = b(s + 1);
    }
    acc[0] = mfma(a0.x, bv, acc[0]);
    acc[1] = mfma(a0.y, bv, acc[1]);
    acc[2] = mfma(a0.z, bv, acc[2]);
    acc[3] = mfma(a0.w, bv, acc[3]);
    acc[4] = mfma(a1.x, bv, acc[4]);
    acc[5] = mfma(a1.y, bv, acc[5]);
    acc[6] = mfma(a1.z, bv, acc[6]);
    acc[7] = mfma(a1.w, bv, acc[7]);
    if (s + 1 < KS) __builtin_amdgcn_sched_group_barrier(0x100, NDS, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
    a0 = n0; a1 = n1; bv = nb;
  }
}


// Offsets of this lane's 16 accumulator registers inside a [256 feature][32 sample] block (mlp_layout.h
// blk_off): the swizzle term (f>>1)&7 does not depend on the tile index, so tile t just adds t*1024.
__device__ __forceinline__ void lane_block_offsets(int j, int hh, int (&o16)[16]) {
#pragma unroll
  for (int r = 0; r < 16; ++r) o16[r] = blk_off(feat_of(0, r, hh), j);
}
// ---- 16-byte record stores -------------------------------------------------------------------------------------------
// A lane owns ONE sample and, per accumulator tile, four groups of four consecutive FEATURES (registers 4g..4g+3 <->
// features 8g + 4hh + 0..3); the record keeps a feature's 32 samples contiguous, so storing register by register is a
// 4-byte store per lane and instruction -- and the store path is ISSUE-bound (~140 cycles per store instruction measured in
// the recording forward: 125 k of 382 k cycles per tile).  Transposing each 4x4 (feature, sample) block inside its quad of
// lanes first (16 DPP / select instructions) turns four 4-byte stores into one 16-byte store of four consecutive samples
// of one feature: lane (j & 3) = i ends up with feature i of samples 4q..4q+3.  ALL lanes of the wave must take part.
__device__ __forceinline__ float quad_dpp(float x, bool swap2) {
  const int v = __float_as_int(x);
  return __int_as_float(swap2 ? __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true)      // quad_perm [2,3,0,1]
                              : __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
}
__device__ __forceinline__ void quad_transpose(float (&v)[4], bool odd, bool hi) {
  const float x1 = quad_dpp(v[1], false), y0 = quad_dpp(v[0], false), x3 = quad_dpp(v[3], false), y2 = quad_dpp(v[2], false);
  const float c0 = odd ? x1 : v[0], c1 = odd ? v[1] : y0, c2 = odd ? x3 : v[2], c3 = odd ? v[3] : y2;
  const float z0 = quad_dpp(c0, true), z1 = quad_dpp(c1, true), z2 = quad_dpp(c2, true), z3 = quad_dpp(c3, true);
  v[0] = hi ? z2 : c0; v[1] = hi ? z3 : c1; v[2] = hi ? c2 : z0; v[3] = hi ? c3 : z1;
}
// float offsets (inside a tile's 1024 floats of a [256][32] block) of the four 16-byte stores of this lane: group g holds
// feature 8g + 4hh + (j & 3), samples (j & ~3) .. +3
__device__ __forceinline__ void lane_quad_offsets(int j, int hh, int (&o4)[4]) {
#pragma unroll
  for (int g = 0; g < 4; ++g) o4[g] = blk_off(8 * g + 4 * hh + (j & 3), j & ~3);
}
// one accumulator tile (16 values of this lane's sample) -> its 1024 floats of the block; all 64 lanes must call it
__device__ __forceinline__ void store_tile_quads(float* tile_base, const float (&h)[16], int j, const int (&o4)[4]) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    float v[4] = {h[4 * g + 0], h[4 * g + 1], h[4 * g + 2], h[4 * g + 3]};
    quad_transpose(v, (j & 1) != 0, (j & 2) != 0);
    __builtin_nontemporal_store(f32x4{v[0], v[1], v[2], v[3]}, reinterpret_cast<f32x4*>(tile_base + o4[g]));   // (read once, by other kernels)
  }
}

__device__ __forceinline__ void store_block(float* blk, const f32x16 (&h)[8], const int (&o16)[16]) {
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    float* bt = blk + t * 1024;
#pragma unroll
    for (int r = 0; r < 16; ++r) bt[o16[r]] = h[t][r];
  }
}

// one generic 256->256 part: 8 chunks, chunk ts contracts the 32 features of tile ts of `h`
__device__ __forceinline__ void part_gen(Pipe& p, f32x16 (&acc)[8], const f32x16 (&h)[8], int n_chunks) {
#pragma unroll
  for (int ts = 0; ts < 8; ++ts) {
    chunk_begin(p, n_chunks);
    const f32x16 hv = h[ts];
    mma_wide<16, 2>(acc, chunk_ptr(p), [&](int s) { return hv[s]; });
    chunk_end(p, n_chunks);
  }
}

}  // namespace tp_mma
