// K3: backward of the two trainable heads (mlp_rgb, mlp_trans) and the per-image latents (gfx950).
//
// Replaces torch autograd through layers/nerf_static_transient_light.py:102-137.  The trunk is
// frozen and evaluated under no_grad in the reference (:34,87-100,236-239), so nothing flows into
// mlp_feat; gradients are produced for mlp_rgb.{0..3}, mlp_trans.{0..3} and the latent rows only.
//
// Three kernels on one stream:
//  1. mlp_dgrad_kernel -- same register-resident structure as the forward (samples on MFMA lanes,
//     transposed weights streamed L2->LDS as the A operand): dz3 from the output activations'
//     derivatives, then dz_l = (W_{l+1}^T dz_{l+1}) * [h_l > 0] for l = 2,1,0 of each head, written
//     as [256 feature][32 sample] blocks (XOR-swizzled sample quads).
//  2. the weight gradient -- dW_l = sum_s dz_l[:,s] in_l[:,s]^T as MFMA GEMMs whose k axis is the SAMPLE axis: both
//     operands are lane-linear LDS copies (LDS-DMA) of the recorded blocks, fragments are conflict-free
//     ds_read_b128 thanks to the swizzle; split-K over workgroups.  Two extra column tiles ride along: a one-hot
//     "image id" tile that yields per-image sums of dz (bias gradients, and through linearity everything that
//     multiplies a per-image latent) and, for mlp_rgb.0, the recorded [view encoding, x] columns.
//       mlp_wgrad_kernel        fp32 MFMA, records of the fp32 forward: 128 output rows x 10 column tiles per workgroup;
//       mlp_wgrad_f16x3_kernel  split-fp16 (hi hi + hi lo + lo hi), records of the range-checked f16x3 forward: one
//                               workgroup per whole 256 x 256 GEMM of a sample slice, software-pipelined (its header).
//  3. mlp_wgrad_finalize -- fixed-order reduction of the split-K partials (deterministic: no float atomics): weights
//     in the reference parameter layouts and per-image dz sums; mlp_wgrad_finalize2 then forms the biases, the latent
//     columns of the two first-layer weights and the latent-row gradients
//     dlat[b] = W0[:,latent cols]^T (sum_{s in b} dz0[:,s]).
#include "mlp_mma.h"

namespace {
using namespace tp_layout;
using namespace tp_mma;

// ------------------------------------------------------------------------------------------------
struct WPtrs { const float* w[16]; };

__global__ void packT_kernel(WPtrs w, float* __restrict__ out) {
  const int64_t n = (int64_t)kNumChunksT * kChunkFloats;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    int mat, o, f;
    chunkT_src((int)(e / kChunkFloats), (int)(e % kChunkFloats), mat, o, f);
    out[e] = o < 0 ? 0.0f : w.w[mat][(int64_t)o * 256 + f];
  }
}

// ------------------------------------------------------------------------------------------------
struct DgParams {
  const float* packed_t; const float* saved;
  const float* rgb; const float* density; const float* uncert;
  const float* g_rgb; const float* g_density; const float* g_uncert;
  int64_t n_samples, n_tiles;
  float* dz;
  unsigned int* dz_max;    // bits of max |dz| over the call (atomicMax; non-negative floats order like their bits)
};

__global__ __launch_bounds__(kThreads, 1) void mlp_dgrad_kernel(DgParams P) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, hh = lane >> 5;
  Pipe p;
  p.stream = P.packed_t; p.lds = lds; p.chunk = 0; p.buf = 0; p.wave = __builtin_amdgcn_readfirstlane(wave); p.lane = lane;
  dma_chunk(p, 0, 0);
  __syncthreads();

  for (int64_t tile = blockIdx.x; tile < P.n_tiles; tile += gridDim.x) {
    const int64_t s_raw = tile * 128 + wave * 32 + j;
    const bool live = s_raw < P.n_samples;
    const int64_t s = live ? s_raw : P.n_samples - 1;
    const int64_t gidx = tile * 4 + wave;
    const float* sv = P.saved + gidx * (int64_t)kSavedGroupFloats;
    float* dzg = P.dz + gidx * (int64_t)kDzGroupFloats;
    f32x16 h[8], acc[8];
    int o16[16];
    lane_block_offsets(j, hh, o16);
    float dzm = 0.0f;

#pragma nounroll
    for (int st = 0; st < 6; ++st) {
      const int head = st / 3, k = st - head * 3;
      // ReLU sign bits of the activation this step's result is masked with (recorded by the forward)
      const int slot = (head == 0 ? SV_T2 : SV_R2) - k;
      const uint32_t* mk = reinterpret_cast<const uint32_t*>(sv + kMaskOff) + (slot - 1) * 256 + lane;
      uint32_t mask[4];
#pragma unroll
      for (int w = 0; w < 4; ++w) mask[w] = mk[w * 64];
#pragma unroll
      for (int t = 0; t < 8; ++t) acc[t] = f32x16{0};
      if (k == 0) {
        // derivative of the output non-linearities (sigmoid: y(1-y); softplus: 1-exp(-y))
        float d[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (live) {
          const int sel = head == 0 ? 1 : 0;          // head 0 = transient (last dim 1), head 1 = static rgb
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const float y = P.rgb[s * 6 + c * 2 + sel];
            d[c] = P.g_rgb[s * 6 + c * 2 + sel] * y * (1.0f - y);
          }
          if (head == 0) {
            d[3] = P.g_density[s * 2 + 1] * (1.0f - expf(-P.density[s * 2 + 1]));
            d[4] = P.g_uncert[s] * (1.0f - expf(-P.uncert[s]));
          }
        }
        float* nb = dzg + (head == 0 ? kDzT3Off : kDzR3Off);
#pragma unroll
        for (int r = 0; r < 16; ++r) nb[blk_off(r + 16 * hh, j)] = (hh == 0 && r < 6) ? d[r < 6 ? r : 0] : 0.0f;
        chunk_begin(p, kNumChunksT);
        mma_wide<3, 2>(acc, chunk_ptr(p), [&](int s_) { return hh ? d[2 * s_ + 1] : d[2 * s_]; });
        chunk_end(p, kNumChunksT);
      } else {
        part_gen(p, acc, h, kNumChunksT);
      }
      // ReLU mask from the recorded activation; the result is both the next B operand and the wgrad A operand
#pragma unroll
      for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const bool on = (mask[t >> 1] >> ((t & 1) * 16 + r)) & 1u;
          h[t][r] = (live && on) ? acc[t][r] : 0.0f;
          dzm = fmaxf(dzm, fabsf(h[t][r]));
        }
      store_block(dzg + st * kBlockFloats, h, o16);
    }
    // range of the gradient record for the split-fp16 weight-gradient GEMM (one atomic per wave and tile)
    for (int off = 32; off >= 1; off >>= 1) dzm = fmaxf(dzm, __shfl_xor(dzm, off, 64));
    if (lane == 0 && P.dz_max != nullptr && dzm == dzm && dzm < 3.0e38f) atomicMax(P.dz_max, __float_as_uint(dzm));
  }
}

// ------------------------------------------------------------------------------------------------
constexpr int kWgItems = 14;     // 6 wide GEMMs x 2 row halves + 2 narrow (output-layer) GEMMs
constexpr int kWgTiles = 10;     // 8 input-feature tiles + image one-hot tile + [view enc, x] tile
constexpr int kWgBufFloats = 4096 + 8192 + 1024;

using half8w = __attribute__((ext_vector_type(8))) _Float16;
using u32x4w = __attribute__((ext_vector_type(4))) unsigned int;

__device__ __forceinline__ f32x16 mfma16w(half8w a, half8w b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

struct WgParams {
  const float* saved; const float* dz;
  int64_t n_samples, n_groups, rn;   // rn = samples per image (R*N)
  int n_slices, groups_per_slice;
  float* partial;
};

__device__ __forceinline__ void wg_dma(const WgParams& P, int64_t g, float* buf, int a_off, int a_pieces, int b_slot,
                                       bool has_ex, int wave, int lane) {
  const float* dzg = P.dz + g * (int64_t)kDzGroupFloats + a_off;
  const float* svg = P.saved + g * (int64_t)kSavedGroupFloats;
  for (int pc = wave; pc < a_pieces; pc += 4)
    __builtin_amdgcn_global_load_lds(AS1(dzg + pc * 256 + lane * 4), AS3(buf + pc * 256), 16, 0, 0);
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int pc = wave * 8 + k;
    __builtin_amdgcn_global_load_lds(AS1(svg + b_slot * kBlockFloats + pc * 256 + lane * 4),
                                     AS3(buf + 4096 + pc * 256), 16, 0, 0);
  }
  if (has_ex)
    __builtin_amdgcn_global_load_lds(AS1(svg + SV_EX * kBlockFloats + wave * 256 + lane * 4),
                                     AS3(buf + 4096 + 8192 + wave * 256), 16, 0, 0);
}

// fp32-MFMA weight gradient (records of the fp32 forward): 128 output rows x 10 column tiles per workgroup.
__global__ __launch_bounds__(kThreads, 1) void mlp_wgrad_kernel(WgParams P) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 31, hh = lane >> 5;
  // Workgroup -> (item, slice).  The two row halves of a wide GEMM read the same activation blocks; consecutive
  // workgroup ids go round-robin over the 8 XCDs, so the halves are placed 8 ids apart (same XCD, same L2, same time):
  // the second read of every block is then an L2 hit instead of a second trip over the fabric.
  const int xcd = blockIdx.x & 7, kk = blockIdx.x >> 3;
  const int pair = (kk >> 1) * 8 + xcd, quad_sel = kk & 1;
  if (pair >= (kWgItems / 2) * P.n_slices) return;          // padding workgroups of the last round
  const int gp = pair % (kWgItems / 2), slice = pair / (kWgItems / 2);
  const int item = gp < 6 ? 2 * gp + quad_sel : 12 + quad_sel;
  const bool big = item < 12;
  const int gemm = item >> 1, quad = item & 1;
  const int a_off = big ? gemm * kBlockFloats + quad * 4096 : (item == 12 ? kDzT3Off : kDzR3Off);
  const int a_pieces = big ? 16 : 4;
  int b_slot;
  if (big) {
    b_slot = gemm == 0 ? SV_T1 : gemm == 1 ? SV_T0 : gemm == 2 ? SV_FEAT : gemm == 3 ? SV_R1 : gemm == 4 ? SV_R0 : SV_FEAT;
  } else {
    b_slot = item == 12 ? SV_T2 : SV_R2;
  }
  const bool has_ex = big && gemm == 5;
  const int64_t g0 = (int64_t)slice * P.groups_per_slice;
  const int64_t g1 = g0 + P.groups_per_slice < P.n_groups ? g0 + P.groups_per_slice : P.n_groups;

  // narrow items reuse the wide code path with a 128-row A block whose rows 32.. are zero
  if (!big)
    for (int e = tid; e < 2 * 3072; e += kThreads) lds[(e / 3072) * kWgBufFloats + 1024 + (e % 3072)] = 0.0f;

  f32x16 acc[kWgTiles];
#pragma unroll
  for (int t = 0; t < kWgTiles; ++t) acc[t] = f32x16{0};

  // A rows of this wave: feature index (for the swizzle) and LDS row
  const int fa = (big ? quad * 128 : 0) + wave * 32 + i;
  const int a_row = wave * 32 + i;
  const int64_t lo = (int64_t)i * P.rn;                       // sample range of image `i` (one-hot tile column)
  const int64_t hi = lo + P.rn < P.n_samples ? lo + P.rn : P.n_samples;

  int buf = 0;
  if (g0 < g1) wg_dma(P, g0, lds, a_off, a_pieces, b_slot, has_ex, wave, lane);
  __syncthreads();
  for (int64_t g = g0; g < g1; ++g) {
    if (g + 1 < g1) wg_dma(P, g + 1, lds + (buf ^ 1) * kWgBufFloats, a_off, a_pieces, b_slot, has_ex, wave, lane);
    const float* A = lds + buf * kWgBufFloats;
    const float* Bm = A + 4096;
    const float* Ex = Bm + 8192;
    const int64_t sbase = g * 32 + 4 * hh;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int sq = 2 * q + hh;                               // sample quad of this lane half
      const f32x4 a4 = *reinterpret_cast<const f32x4*>(A + a_row * 32 + ((sq ^ ((fa >> 1) & 7)) << 2));
      f32x4 b4[8];
#pragma unroll
      for (int ft = 0; ft < 8; ++ft) {
        const int f = ft * 32 + i;
        b4[ft] = *reinterpret_cast<const f32x4*>(Bm + f * 32 + ((sq ^ ((f >> 1) & 7)) << 2));
      }
      f32x4 e4 = {0.f, 0.f, 0.f, 0.f};
      if (has_ex) e4 = *reinterpret_cast<const f32x4*>(Ex + i * 32 + ((sq ^ ((i >> 1) & 7)) << 2));
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int64_t smp = sbase + 8 * q + m;
        const float onehot = (smp >= lo && smp < hi) ? 1.0f : 0.0f;
#pragma unroll
        for (int ft = 0; ft < 8; ++ft) acc[ft] = mfma(a4[m], b4[ft][m], acc[ft]);
        acc[8] = mfma(a4[m], onehot, acc[8]);
        if (has_ex) acc[9] = mfma(a4[m], e4[m], acc[9]);
      }
    }
    __syncthreads();
    buf ^= 1;
  }
  float* out = P.partial + (((int64_t)item * P.n_slices + slice) * 4 + wave) * (kWgTiles * 1024);
#pragma unroll
  for (int t = 0; t < kWgTiles; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) out[(t * 16 + r) * 64 + lane] = acc[t][r];
}

// ------------------------------------------------------------------------------------------------
// Split-fp16 weight gradient (records of the range-checked f16x3 forward).  The products run on the f16 matrix cores as
// hi*hi + hi*lo + lo*hi with both operands split on the fly (the gradient record is first scaled by a power of two so
// that its largest entry sits at 2^13: exact, undone in the epilogue); 16 samples per MFMA, fp32-grade accuracy.
// One workgroup owns a WHOLE 256 x 256 GEMM of one sample slice: the four waves form a 2 x 2 grid of 128 x 128 quadrants
// (16 accumulator tiles each), so every recorded block is fetched by exactly one workgroup, an operand fragment is split
// once per 4 tiles instead of once per tile, and LDS fragment reads per MFMA fall to a third of the 32-row-per-wave
// layout above.  The one-hot (bias / latent) and [view enc, x] column tiles of a row block ride with one of its two waves.
// The two output-layer GEMMs (3 and 5 rows) are narrow workgroups: 32 rows, a wave per 64 columns, a fetch stream with
// almost no work; they get fewer, longer slices.  Partials keep the layout of the kernel above.  DESIGN.md section 4 K3
// has the measurements behind each choice (fetch in two parts, pinned pipeline, counted waits).
constexpr int kWg2BufFloats = 8192 + 8192 + 1024;

// The 16 quadrant tiles fill the 256 AGPRs (the compiler's MFMA form for this kernel); the extra column tiles must not
// compete for them, so their MFMAs are issued in the VGPR form by hand.  Hazards the compiler cannot see through the asm:
// a VALU result needs two wait states before an MFMA reads it (the s_nop; without it the first of these MFMAs read a
// half-written one-hot operand); dependent MFMAs on one accumulator are interlocked by the hardware; compiled code reads
// these tiles only in the epilogue, behind wg2_mfma_drain().
__device__ __forceinline__ void mfma16w_vgpr(f32x16& c, const half8w& a, const half8w& b) {
  asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ void wg2_mfma_drain() { asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); }

struct Wg2Params {
  const float* saved; const float* dz;
  const unsigned int* dz_max;
  int64_t n_samples, n_groups, rn;   // rn = samples per image (R*N)
  int n_w, n_n, gps_w, gps_n;        // slices / groups per slice of the wide and the narrow GEMMs
  float* partial;
};

// LDS-DMA of one group's operands, in two parts (A block; B block and view/x rows).  A wave keeps at most ~9 of these
// 1 KB pieces in flight: with all 17 issued at once the same bytes take 1.8x as long to arrive (measured, DMA only).
template <int PART>
__device__ __forceinline__ void wg2_dma(const Wg2Params& P, int64_t g, float* buf, int a_off, int b_slot,
                                        bool has_ex, int wave, int lane) {
  if (PART == 0) {
    const float* dzg = P.dz + g * (int64_t)kDzGroupFloats + a_off;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int pc = wave * 8 + k;
      __builtin_amdgcn_global_load_lds(AS1(dzg + pc * 256 + lane * 4), AS3(buf + pc * 256), 16, 0, 0);
    }
  } else {
    const float* svg = P.saved + g * (int64_t)kSavedGroupFloats;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int pc = wave * 8 + k;
      __builtin_amdgcn_global_load_lds(AS1(svg + b_slot * kBlockFloats + pc * 256 + lane * 4),
                                       AS3(buf + 8192 + pc * 256), 16, 0, 0);
    }
    if (has_ex)
      __builtin_amdgcn_global_load_lds(AS1(svg + SV_EX * kBlockFloats + wave * 256 + lane * 4),
                                       AS3(buf + 16384 + wave * 256), 16, 0, 0);
  }
}

// power-of-two scaling of the gradient record from the largest |dz| of the call
__device__ __forceinline__ void wg2_scales(const Wg2Params& P, float& dz_scale, float& out_scale) {
  dz_scale = 1.0f; out_scale = 1.0f;
  const float mx = __uint_as_float(*P.dz_max);
  if (mx > 1.0e-30f && mx < 1.0e30f) {
    int e;
    (void)frexpf(mx, &e);                       // mx = m * 2^e, m in [0.5, 1)
    dz_scale = ldexpf(1.0f, 14 - e);            // largest |dz| lands in [2^13, 2^14)
    out_scale = ldexpf(1.0f, e - 14);
  }
}

// one-hot "image id" B fragment of 8 consecutive samples: column i is image i, whose samples are [lo, hi).  `rel` is the
// first sample's offset from lo clamped into int range, `len` = hi - lo likewise: sample m is inside iff
// (unsigned)(rel + m) < (unsigned)len
__device__ __forceinline__ int wg2_clamp_rel(int64_t d) {
  return d < -64 ? -64 : (d > (int64_t)1 << 30 ? 1 << 30 : (int)d);
}
__device__ __forceinline__ half8w wg2_onehot(int rel, int len) {
  half8w oh;
#pragma unroll
  for (int m = 0; m < 8; ++m) oh[m] = (unsigned)(rel + m) < (unsigned)len ? (_Float16)1.0f : (_Float16)0.0f;
  return oh;
}

// pipeline pieces of the wide GEMM: a fragment's two raw sample quads as read from LDS, and its hi / lo fp16 operands
// built two elements at a time (wg2_split2 step k = elements 2k, 2k+1)
struct Raw { f32x4 q[2]; };
struct Frag {
  u32x4w h, l;
  __device__ __forceinline__ half8w hi() const { return __builtin_bit_cast(half8w, h); }
  __device__ __forceinline__ half8w lo() const { return __builtin_bit_cast(half8w, l); }
};
__device__ __forceinline__ void wg2_raw(const float* blk, int row, int sq0, Raw& r) {
  const int s = (row >> 1) & 7;
  r.q[0] = *reinterpret_cast<const f32x4*>(blk + row * 32 + ((sq0 ^ s) << 2));
  r.q[1] = *reinterpret_cast<const f32x4*>(blk + row * 32 + (((sq0 + 1) ^ s) << 2));
}
// 8 fp32 values (two sample quads) * scale -> hi + lo fp16 operands, two elements per step:
// hi = round-toward-zero conversion (= the value truncated to 11 significant bits: exact in fp16, one instruction for two
// elements), lo = value - hi (exact in fp32; v_fma_mix_f32 reads hi as fp16), rounded
// toward zero
__device__ __forceinline__ void wg2_split2(const Raw& r, int k, float scale, Frag& f) {
  const float v0 = r.q[k >> 1][(2 * k) & 3] * scale, v1 = r.q[k >> 1][(2 * k + 1) & 3] * scale;
  const unsigned int h = __builtin_bit_cast(unsigned int, __builtin_amdgcn_cvt_pkrtz(v0, v1));
  float l0, l1;
  asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l0) : "v"(h), "v"(v0));
  asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(l1) : "v"(h), "v"(v1));
  f.h[k] = h;
  f.l[k] = __builtin_bit_cast(unsigned int, __builtin_amdgcn_cvt_pkrtz(l0, l1));
}

// A wide (256 x 256) GEMM of one sample slice; HAS_EX is compile-time so that each variant is a straight-line loop
// whose accumulator tiles never change register class.
template <bool HAS_EX>
__device__ __forceinline__ void wg2_run_wide(const Wg2Params& P, float* lds, int gemm, int slice, int wave, int lane) {
  const int i = lane & 31, hh = lane >> 5;
  const int wr = wave >> 1, wc = wave & 1;
  const int a_off = gemm * kBlockFloats;
  const int b_slot = gemm == 0 ? SV_T1 : gemm == 1 ? SV_T0 : gemm == 2 ? SV_FEAT : gemm == 3 ? SV_R1 : gemm == 4 ? SV_R0 : SV_FEAT;
  const int64_t g0 = (int64_t)slice * P.gps_w;
  const int64_t g1 = g0 + P.gps_w < P.n_groups ? g0 + P.gps_w : P.n_groups;

  f32x16 acc[16], hot[2], ex[2];
#pragma unroll
  for (int t = 0; t < 16; ++t) acc[t] = f32x16{0};
  hot[0] = hot[1] = ex[0] = ex[1] = f32x16{0};

  const int64_t lo = (int64_t)i * P.rn;                       // sample range of image `i` (one-hot tile column)
  const int64_t hi = lo + P.rn < P.n_samples ? lo + P.rn : P.n_samples;
  const int len = wg2_clamp_rel(hi - lo);
  float dz_scale, out_scale;
  wg2_scales(P, dz_scale, out_scale);
  // rows of this wave's extra column tiles: wave column 0 takes row tiles 0, 1 of its row block, wave column 1 tiles 2, 3
  const int xrow = wr * 128 + wc * 64 + i;
  int buf = 0;
  if (g0 < g1) {
    wg2_dma<0>(P, g0, lds, a_off, b_slot, HAS_EX, wave, lane);
    wg2_dma<1>(P, g0, lds, a_off, b_slot, HAS_EX, wave, lane);
  }
  __syncthreads();
  // Fetch schedule from here on: the A block of group g + 2 is issued at stage 4 of group g (into the A half of the
  // current buffer, which nobody reads after stage 3), the B block of group g + 1 at stage 0 of group g; each has a whole
  // group to arrive, and a wave has 16-17 pieces in flight only briefly.  Waits are counted (pieces retire in order).
  if (g0 < g1) wg2_dma<0>(P, g0 + 1 < g1 ? g0 + 1 : g0, lds + kWg2BufFloats, a_off, b_slot, HAS_EX, wave, lane);
  Frag fa[4], fan[4], fb[2], fx[2], fe;          // A (this half / the other half), B (ping-pong), extra-tile A, view/x
  if (g0 < g1) {
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {                // (later groups get these during the previous group's second half)
      Raw r0;
      wg2_raw(lds, wr * 128 + rt * 32 + i, 2 * hh, r0);
#pragma unroll
      for (int k = 0; k < 4; ++k) wg2_split2(r0, k, dz_scale, fa[rt]);
    }
  }
  for (int64_t g = g0; g < g1; ++g) {
    float* nxt = lds + (buf ^ 1) * kWg2BufFloats;
    const int64_t gn = g + 1 < g1 ? g + 1 : g, gnn = g + 2 < g1 ? g + 2 : g1 - 1;   // (the last steps fetch again: no branch)
    float* A = lds + buf * kWg2BufFloats;
    const float* Bm = A + 8192;
    const float* Ex = Bm + 8192;
    // ---- software pipeline over the 8 (sample half Q, column tile ct) stages of the group ----
    // Stage s issues 12 MFMAs (4 row tiles x {hi hi, hi lo, lo hi}) for B fragment s; between them, in program order
    // pinned by sched_barrier, run the LDS reads of stage s + 2's B fragment and the split steps (2 elements each) of
    // stage s + 1's B fragment, of row tile s of the second half's A fragments (stages 0-3) or of the NEXT group's
    // first-half A fragments (stages 4-7: its A block was fetched during stages 0-3; the barrier before stage 4 says
    // it has landed for every wave), and of the extra tiles' fragments.
    Raw ra[2], rb[2], rx[2], re;
    wg2_raw(Bm, wc * 128 + i, 2 * hh, rb[0]);
    wg2_raw(Bm, wc * 128 + 32 + i, 2 * hh, rb[1]);
    wg2_raw(A, wr * 128 + i, 4 + 2 * hh, ra[0]);
#pragma unroll
    for (int k = 0; k < 4; ++k) wg2_split2(rb[0], k, 1.0f, fb[0]);
    const int rel = wg2_clamp_rel(g * 32 + 8 * hh - lo);
    half8w oh = wg2_onehot(rel, len);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const int Q = s >> 2, ct = s & 3, cur = s & 1;
      const int sq0 = 4 * Q + 2 * hh;
      // next group's operands, half per sample half (the last step fetches its own group again: no branch in the body)
      if (s == 0) wg2_dma<1>(P, gn, nxt, a_off, b_slot, HAS_EX, wave, lane);
      if (s == 4) {
        // A(g + 1) has landed once only this wave's B(g + 1) pieces are outstanding; every wave is past its last read
        // of A(g) (the extra tiles' second-half rows were read in stage 3)
        if (HAS_EX) asm volatile("s_waitcnt vmcnt(9) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        wg2_dma<0>(P, gnn, A, a_off, b_slot, HAS_EX, wave, lane);
      }
      if (s + 2 < 8) wg2_raw(Bm, (wc * 4 + ((s + 2) & 3)) * 32 + i, 4 * ((s + 2) >> 2) + 2 * hh, rb[cur]);
      if (s < 3) wg2_raw(A, wr * 128 + (s + 1) * 32 + i, 4 + 2 * hh, ra[(s + 1) & 1]);
      if (s == 4) wg2_raw(nxt, wr * 128 + i, 2 * hh, ra[0]);
      if (s >= 4 && s < 7) wg2_raw(nxt, wr * 128 + (s - 3) * 32 + i, 2 * hh, ra[(s + 1) & 1]);
      if (s == 0) { wg2_raw(A, xrow, 2 * hh, rx[0]); wg2_raw(A, xrow + 32, 2 * hh, rx[1]); }
      if (s == 3) { wg2_raw(A, xrow, 4 + 2 * hh, rx[0]); wg2_raw(A, xrow + 32, 4 + 2 * hh, rx[1]); }
      if (HAS_EX && ct == 2) wg2_raw(Ex, i, sq0, re);
      if (s == 4) oh = wg2_onehot(rel + 16, len);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k = 0; k < 12; ++k) {
        const int rt = k & 3, term = k >> 2;
        const half8w av = Q == 0 ? (term < 2 ? fa[rt].hi() : fa[rt].lo()) : (term < 2 ? fan[rt].hi() : fan[rt].lo());
        const half8w bv = term == 1 ? fb[cur].lo() : fb[cur].hi();
        acc[rt * 4 + ct] = mfma16w(av, bv, acc[rt * 4 + ct]);
        if (k < 4) {
          if (s < 7) wg2_split2(rb[cur ^ 1], k, 1.0f, fb[cur ^ 1]);
          else if (HAS_EX) wg2_split2(re, k, 1.0f, fe);
        } else if (k < 8) {
          if (s < 4) wg2_split2(ra[s & 1], k - 4, dz_scale, fan[s]);
          else wg2_split2(ra[s & 1], k - 4, dz_scale, fa[s - 4]);
        } else {
          if (ct == 1 || ct == 2) wg2_split2(rx[ct - 1], k - 8, dz_scale, fx[ct - 1]);
          if (HAS_EX && s == 3) wg2_split2(re, k - 8, 1.0f, fe);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (ct == 3) {                                           // the extra column tiles of this sample half
        mfma16w_vgpr(hot[0], fx[0].hi(), oh);
        mfma16w_vgpr(hot[1], fx[1].hi(), oh);
        mfma16w_vgpr(hot[0], fx[0].lo(), oh);
        mfma16w_vgpr(hot[1], fx[1].lo(), oh);
        if constexpr (HAS_EX) {
          mfma16w_vgpr(ex[0], fx[0].hi(), fe.hi());
          mfma16w_vgpr(ex[1], fx[1].hi(), fe.hi());
          mfma16w_vgpr(ex[0], fx[0].hi(), fe.lo());
          mfma16w_vgpr(ex[1], fx[1].hi(), fe.lo());
          mfma16w_vgpr(ex[0], fx[0].lo(), fe.hi());
          mfma16w_vgpr(ex[1], fx[1].lo(), fe.hi());
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");     // B(g + 1) landed; A(g + 2) may still be in flight
    __builtin_amdgcn_s_barrier();
    buf ^= 1;
  }
  wg2_mfma_drain();
  // partials in the layout of the 128-row kernel: chunk (item = 2 gemm + row half, slice), then
  // [row tile within the half][column tile 0..7, 8 = one-hot, 9 = view/x][16 registers][64 lanes]
  float* out = P.partial + ((int64_t)(2 * gemm + wr) * P.n_w + slice) * ((int64_t)4 * kWgTiles * 1024);
#pragma unroll
  for (int rt = 0; rt < 4; ++rt)
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        out[((rt * kWgTiles + wc * 4 + ct) * 16 + r) * 64 + lane] = acc[rt * 4 + ct][r] * out_scale;
#pragma unroll
  for (int k = 0; k < 2; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      out[(((2 * wc + k) * kWgTiles + 8) * 16 + r) * 64 + lane] = hot[k][r] * out_scale;
      if constexpr (HAS_EX) out[(((2 * wc + k) * kWgTiles + 9) * 16 + r) * 64 + lane] = ex[k][r] * out_scale;
    }
}

// A narrow GEMM (output layer: 32 A rows of which 5 / 3 are real) of one sample slice: wave w owns column tiles 2w, 2w+1,
// wave 0 also the one-hot tile.  Almost no work per byte, so it is a fetch stream: a ring of four group slots with three
// groups in flight (a wave's 9 pieces per group retire in order, so the wait for the oldest is counted).
constexpr int kNarrowSlots = 4;
constexpr int kNarrowGroupFloats = 1024 + 8192;          // A [32][32] + B [256][32] of one group
constexpr int kWg2NarrowBufFloats = kNarrowSlots * kNarrowGroupFloats / 2;
constexpr int kWg2LdsBytes = 2 * (kWg2BufFloats > kWg2NarrowBufFloats ? kWg2BufFloats : kWg2NarrowBufFloats) * (int)sizeof(float);

// group g of a narrow GEMM into a slot (9 pieces per wave)
__device__ __forceinline__ void wg2_dma_narrow(const Wg2Params& P, int64_t g, float* dst, int a_off, int b_slot,
                                               int wave, int lane) {
  const float* dzg = P.dz + g * (int64_t)kDzGroupFloats + a_off;
  const float* svg = P.saved + g * (int64_t)kSavedGroupFloats + b_slot * kBlockFloats;
  __builtin_amdgcn_global_load_lds(AS1(dzg + wave * 256 + lane * 4), AS3(dst + wave * 256), 16, 0, 0);
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int pc = wave * 8 + k;
    __builtin_amdgcn_global_load_lds(AS1(svg + pc * 256 + lane * 4), AS3(dst + 1024 + pc * 256), 16, 0, 0);
  }
}

__device__ __forceinline__ void wg2_run_narrow(const Wg2Params& P, float* lds, int gemm, int slice, int wave, int lane) {
  const int i = lane & 31, hh = lane >> 5;
  const int a_off = gemm == 6 ? kDzT3Off : kDzR3Off;
  const int b_slot = gemm == 6 ? SV_T2 : SV_R2;
  const int64_t g0 = (int64_t)slice * P.gps_n;
  const int64_t g1 = g0 + P.gps_n < P.n_groups ? g0 + P.gps_n : P.n_groups;
  f32x16 acc[2], hot;
  acc[0] = acc[1] = hot = f32x16{0};
  const int64_t lo = (int64_t)i * P.rn;
  const int64_t hi = lo + P.rn < P.n_samples ? lo + P.rn : P.n_samples;
  const int len = wg2_clamp_rel(hi - lo);
  float dz_scale, out_scale;
  wg2_scales(P, dz_scale, out_scale);
  if (g0 < g1) {
#pragma unroll
    for (int j = 0; j < kNarrowSlots - 1; ++j)                // (groups past the end are fetched again: constant counts)
      wg2_dma_narrow(P, g0 + j < g1 ? g0 + j : g1 - 1, lds + j * kNarrowGroupFloats, a_off, b_slot, wave, lane);
  }
  for (int64_t g = g0; g < g1; ++g) {
    asm volatile("s_waitcnt vmcnt(18) lgkmcnt(0)" ::: "memory");       // group g landed; g + 1, g + 2 may be in flight
    __builtin_amdgcn_s_barrier();                                      // ... for every wave, and slot g - 1 is read out
    const int64_t gf = g + kNarrowSlots - 1 < g1 ? g + kNarrowSlots - 1 : g1 - 1;
    wg2_dma_narrow(P, gf, lds + (int)((g - g0 + kNarrowSlots - 1) % kNarrowSlots) * kNarrowGroupFloats, a_off, b_slot, wave,
                   lane);
    const float* A = lds + (int)((g - g0) % kNarrowSlots) * kNarrowGroupFloats;
    const float* Bm = A + 1024;
    Raw ra[2], rb[2][2];                                               // all six fragments' reads in flight at once
#pragma unroll
    for (int Q = 0; Q < 2; ++Q) {
      wg2_raw(A, i, 4 * Q + 2 * hh, ra[Q]);
#pragma unroll
      for (int c = 0; c < 2; ++c) wg2_raw(Bm, (wave * 2 + c) * 32 + i, 4 * Q + 2 * hh, rb[Q][c]);
    }
    const int rel = wg2_clamp_rel(g * 32 + 8 * hh - lo);
#pragma unroll
    for (int Q = 0; Q < 2; ++Q) {
      Frag fa, fb;
#pragma unroll
      for (int k = 0; k < 4; ++k) wg2_split2(ra[Q], k, dz_scale, fa);
#pragma unroll
      for (int c = 0; c < 2; ++c) {
#pragma unroll
        for (int k = 0; k < 4; ++k) wg2_split2(rb[Q][c], k, 1.0f, fb);
        acc[c] = mfma16w(fa.hi(), fb.hi(), acc[c]);
        acc[c] = mfma16w(fa.hi(), fb.lo(), acc[c]);
        acc[c] = mfma16w(fa.lo(), fb.hi(), acc[c]);
      }
      if (wave == 0) {
        const half8w oh = wg2_onehot(rel + 16 * Q, len);
        mfma16w_vgpr(hot, fa.hi(), oh);
        mfma16w_vgpr(hot, fa.lo(), oh);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  wg2_mfma_drain();
  float* out = P.partial + ((int64_t)12 * P.n_w + (int64_t)(gemm - 6) * P.n_n + slice) * ((int64_t)4 * kWgTiles * 1024);
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) out[((wave * 2 + c) * 16 + r) * 64 + lane] = acc[c][r] * out_scale;
  if (wave == 0)
#pragma unroll
    for (int r = 0; r < 16; ++r) out[(8 * 16 + r) * 64 + lane] = hot[r] * out_scale;
}

__global__ __launch_bounds__(kThreads, 1) void mlp_wgrad_f16x3_kernel(Wg2Params P) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n_wide = 6 * P.n_w;
#if defined(WG2_ONLY_WIDE)                      // measurement builds: one kind of workgroup alone
  if ((int)blockIdx.x >= n_wide) return;
#elif defined(WG2_ONLY_NARROW)
  if ((int)blockIdx.x < n_wide) return;
#endif
  if ((int)blockIdx.x < n_wide) {
    const int gemm = (int)blockIdx.x % 6, slice = (int)blockIdx.x / 6;       // 0..5: mlp_trans.{2,1,0}, mlp_rgb.{2,1,0}
    if (gemm == 5) wg2_run_wide<true>(P, lds, gemm, slice, wave, lane);
    else wg2_run_wide<false>(P, lds, gemm, slice, wave, lane);
  } else {
    const int k = (int)blockIdx.x - n_wide;                                   // 6 = mlp_trans.3, 7 = mlp_rgb.3
    wg2_run_narrow(P, lds, 6 + k % 2, k / 2, wave, lane);
  }
}

// ------------------------------------------------------------------------------------------------
// Phase 1 of the fixed-order reduction: the split-K slices of every partial tile are summed in ascending slice order (one fixed order:
// run-to-run deterministic) and the sums go to their places -- weight matrices get their recorded-feature columns (and mlp_rgb.0 its
// [view enc, x] columns); the one-hot tiles become per-image sums of dz, dzsum[gemm][b][o], from which phase 2 forms everything that
// multiplies a per-image constant.
// One workgroup per 4-KiB partial tile (item, row tile, feature tile); thread t owns floats 4 t .. 4 t + 3 of the tile, i.e. ONE 16-byte
// load per slice and thread, a whole tile read contiguously per slice.  (Round 5's form -- one thread per OUTPUT element, four bytes per
// slice and thread, a wavefront touching two 128-byte pieces 4 KiB apart -- read the 57 MB of partials at 2 TB/s: 28 us at the end of
// the B=4 iteration's critical path.)
struct FinTiles {
  float* w_out[8]; int w_ld[8]; int w_rows[8];   // per gemm id: 0..2 = mlp_trans.{2,1,0}, 3..5 = mlp_rgb.{2,1,0}, 6 = mlp_trans.3, 7 = mlp_rgb.3
  float* dzsum;                                   // [8][32][256]
  const float* partial; int n_w, n_n; int B;
};

__global__ __launch_bounds__(256) void mlp_wgrad_finalize(FinTiles P) {
  // blockIdx.x = (item * 4 + rt) * kWgTiles + ft;  items 0..11 = (gemm 0..5, row half), 12 / 13 = the narrow GEMMs 6 / 7 (rt 0 only)
  const int ft = (int)blockIdx.x % kWgTiles, rt = ((int)blockIdx.x / kWgTiles) & 3, item = (int)blockIdx.x / (4 * kWgTiles);
  const bool wide = item < 12;
  const int gemm = wide ? item >> 1 : item - 6;
  if ((!wide && rt != 0) || (ft == 9 && gemm != 5)) return;            // tiles nobody wrote
  const int ns = wide ? P.n_w : P.n_n;
  const int64_t chunk0 = wide ? (int64_t)item * P.n_w : (int64_t)12 * P.n_w + (int64_t)(gemm - 6) * P.n_n;
  const float4* p = reinterpret_cast<const float4*>(P.partial + ((chunk0 * 4 + rt) * kWgTiles + ft) * 1024) + threadIdx.x;
  const int64_t stride4 = (int64_t)kWgTiles * 1024;                    // float4 units between the slices of a tile (4 kWgTiles 1024 floats)
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int sl0 = 0; sl0 < ns; sl0 += 32) {
    float4 v[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) v[k] = sl0 + k < ns ? p[(int64_t)(sl0 + k) * stride4] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < 32; ++k)
      if (sl0 + k < ns) { s.x += v[k].x; s.y += v[k].y; s.z += v[k].z; s.w += v[k].w; }
  }
  // where the four sums go: tile float q = r * 64 + h * 32 + col holds output row ii = (r & 3) | (h << 2) | ((r >> 2) << 3) of the row tile
  const int q = (int)threadIdx.x * 4, r = q >> 6, h = (q >> 5) & 1, col = q & 31;
  const int ii = (r & 3) | (h << 2) | ((r >> 2) << 3);
  const int o = wide ? (item & 1) * 128 + rt * 32 + ii : ii;
  if (o >= P.w_rows[gemm]) return;
  const float v4[4] = {s.x, s.y, s.z, s.w};
  if (ft < 8) {
    float* dst = P.w_out[gemm] + (int64_t)o * P.w_ld[gemm] + ft * 32 + col;
#pragma unroll
    for (int k = 0; k < 4; ++k) dst[k] = v4[k];
  } else if (ft == 8) {
    float* dst = P.dzsum + (int64_t)gemm * 32 * 256 + o;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (col + k < P.B) dst[(col + k) * 256] = v4[k];
  } else {                                                             // mlp_rgb.0's [view enc, x] columns 256 .. 285
    float* dst = P.w_out[5] + (int64_t)o * P.w_ld[5] + 256 + col;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (col + k < 30) dst[k] = v4[k];
  }
}

// Phase 2: biases (sum over images of dzsum), the latent columns of the two first-layer weights
// (dW0[o][col0 + c] = sum_b dzsum[b][o] lat[b][c]) and the latent rows (dlat[b][c] = sum_o W0[o][col0 + c] dzsum[b][o]:
// one wave per element, lanes over o, butterfly reduction).  Images are added in ascending order everywhere.
struct Fin2Params {
  const float* dzsum;                 // [8][32][256]
  float* bias[8]; int bias_rows[8];   // gemm order of dzsum
  float* g_w_t0; float* g_w_r0;       // mlp_trans.0.weight [256,272], mlp_rgb.0.weight [256,334] gradients
  const float* lat_trans; const float* lat_light;   // [B,16], [B,48]
  const float* w_t0; const float* w_r0;
  float* g_lat_trans; float* g_lat_light;
  int B, n_elem_blocks;
  unsigned int* dz_max;               // cleared here (the call's last kernel; its readers have finished): the next call with this
                                      // workspace and size finds it zero (tp_mlp_bwd_args.dz_max_is_clear)
};

__global__ void mlp_wgrad_finalize2(Fin2Params P) {
  if (blockIdx.x == 0 && threadIdx.x == 0 && P.dz_max != nullptr) *P.dz_max = 0u;
  if ((int)blockIdx.x < P.n_elem_blocks) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < 8 * 256) {
      const int gemm = e >> 8, o = e & 255;
      if (o >= P.bias_rows[gemm]) return;
      const float* dz = P.dzsum + (int64_t)gemm * 32 * 256 + o;
      float v = 0.0f;
#pragma unroll 8
      for (int b = 0; b < P.B; ++b) v += dz[b * 256];
      P.bias[gemm][o] = v;
    } else if (e < 8 * 256 + 256 * 64) {
      const int k = e - 8 * 256, o = k >> 6, c = k & 63;
      const bool tr = c < 16;
      const float* dz = P.dzsum + (int64_t)(tr ? 2 : 5) * 32 * 256 + o;
      float v = 0.0f;
      if (tr) {
#pragma unroll 8
        for (int b = 0; b < P.B; ++b) v += dz[b * 256] * P.lat_trans[b * 16 + c];
        P.g_w_t0[o * 272 + 256 + c] = v;
      } else {
#pragma unroll 8
        for (int b = 0; b < P.B; ++b) v += dz[b * 256] * P.lat_light[b * 48 + (c - 16)];
        P.g_w_r0[o * 334 + 286 + (c - 16)] = v;
      }
    }
    return;
  }
  const int lane = threadIdx.x & 63;
  const int e = ((int)blockIdx.x - P.n_elem_blocks) * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (e >= P.B * 64) return;
  const int b = e / 64, c = e % 64;
  const bool tr = c < 16;
  const float* dz = P.dzsum + (int64_t)(tr ? 2 : 5) * 32 * 256 + b * 256;
  const float* w = tr ? P.w_t0 + 256 + c : P.w_r0 + 286 + (c - 16);
  const int ld = tr ? 272 : 334;
  float v = 0.0f;
#pragma unroll
  for (int k = 0; k < 4; ++k) { const int o = lane + 64 * k; v += w[o * ld] * dz[o]; }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  if (lane == 0) { if (tr) P.g_lat_trans[b * 16 + c] = v; else P.g_lat_light[b * 48 + (c - 16)] = v; }
}

int num_cus() {
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
    cus = 256;
  return cus;
}

int64_t n_groups_of(int64_t n_samples) { return ((n_samples + 127) / 128) * 4; }
int slices_for(int64_t n_groups) {
  int s = num_cus() / kWgItems;
  if (s < 1) s = 1;
  return (int)(n_groups < s ? n_groups : s);
}
size_t align256(size_t x) { return (x + 255) / 256 * 256; }

}  // namespace

int tp_launch_mlp_dgrad_f16x3(const tp_mlp_bwd_args* a, float* dz, unsigned int* dz_max, int grid, hipStream_t stream);  // mlp_fwd_f16x3.hip

extern "C" size_t tp_mlp_packed_t_bytes(void) { return (size_t)kPackedTFloats * sizeof(float); }

extern "C" size_t tp_mlp_bwd_workspace_bytes(int64_t n_samples) {
  const int64_t ng = n_groups_of(n_samples);
  // dz record + split-K partials (sized for the largest slice count any device could ask for: 64)
  return align256((size_t)ng * kDzGroupFloats * sizeof(float)) +
         align256((size_t)kWgItems * 64 * 4 * kWgTiles * 1024 * sizeof(float)) + align256(8 * 32 * 256 * sizeof(float)) + 256;
}

extern "C" int tp_mlp_bwd(const tp_mlp_bwd_args* a, tp_stream_t stream_) {
  TP_REQUIRE(a && a->packed_t && a->saved && a->rgb && a->density && a->uncert && a->g_rgb && a->g_density &&
                 a->g_uncert && a->lat_trans && a->lat_light && a->workspace, "null pointer");
  TP_REQUIRE(a->B > 0 && a->B <= 32 && a->R > 0 && a->N > 0, "bad sizes (at most 32 images per call)");
  for (int i = 0; i < 4; ++i)
    TP_REQUIRE(a->weights.rgb_w[i] && a->weights.trans_w[i] && a->g_rgb_w[i] && a->g_rgb_b[i] && a->g_trans_w[i] &&
                   a->g_trans_b[i], "null weight / gradient pointer");
  TP_REQUIRE(a->g_lat_trans && a->g_lat_light, "null latent gradient pointer");
  TP_REQUIRE(a->wgrad_precision == TP_MLP_FP32 || a->wgrad_precision == TP_MLP_F16X3, "unknown wgrad_precision");
  TP_REQUIRE(a->wgrad_cus >= 0, "negative wgrad_cus");
  hipStream_t stream = (hipStream_t)stream_;
  const int64_t S = (int64_t)a->B * a->R * a->N;
  const int64_t n_tiles = (S + 127) / 128, ng = n_tiles * 4;
  float* dz = (float*)a->workspace;
  float* partial = (float*)((char*)a->workspace + align256((size_t)ng * kDzGroupFloats * sizeof(float)));
  float* dzsum = (float*)((char*)partial + align256((size_t)kWgItems * 64 * 4 * kWgTiles * 1024 * sizeof(float)));
  unsigned int* dz_max = (unsigned int*)((char*)dzsum + align256(8 * 32 * 256 * sizeof(float)));

  const bool f16 = a->wgrad_precision == TP_MLP_F16X3;
  static unsigned long long attr_devices = 0;
  if (tp::first_use_on_device(attr_devices)) {
    hipError_t e = hipFuncSetAttribute((const void*)mlp_dgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       2 * kChunkFloats * (int)sizeof(float));
    if (e == hipSuccess)
      e = hipFuncSetAttribute((const void*)mlp_wgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                              2 * kWgBufFloats * (int)sizeof(float));
    if (e == hipSuccess)
      e = hipFuncSetAttribute((const void*)mlp_wgrad_f16x3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                              kWg2LdsBytes);
    if (e != hipSuccess) { tp::set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
  }
  const int cus = num_cus();
  const int dg_grid = (int)(n_tiles < cus ? n_tiles : cus);
  if (f16) {
    if (!a->dz_max_is_clear) {
      hipError_t e = hipMemsetAsync(dz_max, 0, sizeof(unsigned int), stream);
      if (e != hipSuccess) { tp::set_error("hipMemsetAsync: %s", hipGetErrorString(e)); return (int)e; }
    }
    if (int rc = tp_launch_mlp_dgrad_f16x3(a, dz, dz_max, dg_grid, stream)) return rc;
  } else {
    if (a->repack) {
      WPtrs w;
      for (int i = 0; i < 16; ++i) w.w[i] = nullptr;
      for (int i = 0; i < 4; ++i) { w.w[W_RGB0 + i] = a->weights.rgb_w[i]; w.w[W_TRANS0 + i] = a->weights.trans_w[i]; }
      hipLaunchKernelGGL(packT_kernel, dim3(512), dim3(256), 0, stream, w, (float*)a->packed_t);
    }
    DgParams D;
    D.packed_t = (const float*)a->packed_t; D.saved = a->saved; D.rgb = a->rgb; D.density = a->density;
    D.uncert = a->uncert; D.g_rgb = a->g_rgb; D.g_density = a->g_density; D.g_uncert = a->g_uncert;
    D.n_samples = S; D.n_tiles = n_tiles; D.dz = dz; D.dz_max = nullptr;
    hipLaunchKernelGGL(mlp_dgrad_kernel, dim3((unsigned)dg_grid), dim3(kThreads), 2 * kChunkFloats * sizeof(float), stream, D);
  }

  int n_w, n_n;                                  // split-K slices of the wide / narrow GEMMs (partial layout, finalize)
  int cus_w = cus;                               // CUs the weight-gradient launch fills (one workgroup each)
  if (f16) {
    // one workgroup per CU: 6 wide GEMMs x n_w slices + 2 narrow x n_n; 38 + 14 slices level the two kinds on 256 CUs
    // (measured: 39 + 11 makes the narrow stream the tail, 36 + 20 the wide GEMMs)
    if (a->wgrad_cus > 0 && a->wgrad_cus < cus) cus_w = a->wgrad_cus < 16 ? 16 : a->wgrad_cus;
    n_w = cus_w * 19 / 128;
    if (n_w > 64) n_w = 64;
    if (n_w < 1) n_w = 1;
    n_n = (cus_w - 6 * n_w) / 2;
    if (n_n > 64) n_n = 64;
    if (n_n < 1) n_n = 1;
    if (ng < n_w) n_w = (int)ng;
    if (ng < n_n) n_n = (int)ng;
    Wg2Params Wg;
    Wg.saved = a->saved; Wg.dz = dz; Wg.dz_max = dz_max; Wg.n_samples = S; Wg.n_groups = ng; Wg.rn = (int64_t)a->R * a->N;
    Wg.n_w = n_w; Wg.n_n = n_n;
    Wg.gps_w = (int)((ng + n_w - 1) / n_w); Wg.gps_n = (int)((ng + n_n - 1) / n_n);
    Wg.partial = partial;
    hipLaunchKernelGGL(mlp_wgrad_f16x3_kernel, dim3((unsigned)(6 * n_w + 2 * n_n)), dim3(kThreads),
                       kWg2LdsBytes, stream, Wg);
  } else {
    WgParams Wg;
    Wg.saved = a->saved; Wg.dz = dz; Wg.n_samples = S; Wg.n_groups = ng; Wg.rn = (int64_t)a->R * a->N;
    Wg.n_slices = slices_for(ng);
    Wg.groups_per_slice = (int)((ng + Wg.n_slices - 1) / Wg.n_slices);
    Wg.partial = partial;
    n_w = n_n = Wg.n_slices;
    const unsigned wg_grid = (unsigned)((kWgItems * Wg.n_slices + 15) / 16 * 16);   // whole rounds of 2 x 8 XCDs
    hipLaunchKernelGGL(mlp_wgrad_kernel, dim3(wg_grid), dim3(kThreads), 2 * kWgBufFloats * sizeof(float), stream, Wg);
  }

  FinTiles F;
  // gemm ids: 0..2 = mlp_trans.{2,1,0}, 3..5 = mlp_rgb.{2,1,0}, 6 = mlp_trans.3, 7 = mlp_rgb.3
  float* const w_out[8] = {a->g_trans_w[2], a->g_trans_w[1], a->g_trans_w[0], a->g_rgb_w[2], a->g_rgb_w[1], a->g_rgb_w[0],
                           a->g_trans_w[3], a->g_rgb_w[3]};
  const int w_ld[8] = {256, 256, 272, 256, 256, 334, 256, 256};       // (mlp_trans.0 / mlp_rgb.0: the latent columns are phase 2's)
  const int w_rows[8] = {256, 256, 256, 256, 256, 256, 5, 3};
  for (int g = 0; g < 8; ++g) { F.w_out[g] = w_out[g]; F.w_ld[g] = w_ld[g]; F.w_rows[g] = w_rows[g]; }
  F.dzsum = dzsum; F.partial = partial; F.n_w = n_w; F.n_n = n_n; F.B = a->B;
  hipLaunchKernelGGL(mlp_wgrad_finalize, dim3((unsigned)(kWgItems * 4 * kWgTiles)), dim3(256), 0, stream, F);

  Fin2Params G;
  G.dzsum = dzsum;
  float* const biases[8] = {a->g_trans_b[2], a->g_trans_b[1], a->g_trans_b[0], a->g_rgb_b[2], a->g_rgb_b[1], a->g_rgb_b[0],
                            a->g_trans_b[3], a->g_rgb_b[3]};
  for (int g = 0; g < 8; ++g) { G.bias[g] = biases[g]; G.bias_rows[g] = g < 6 ? 256 : (g == 6 ? 5 : 3); }
  G.g_w_t0 = a->g_trans_w[0]; G.g_w_r0 = a->g_rgb_w[0];
  G.lat_trans = a->lat_trans; G.lat_light = a->lat_light;
  G.w_t0 = a->weights.trans_w[0]; G.w_r0 = a->weights.rgb_w[0];
  G.g_lat_trans = a->g_lat_trans; G.g_lat_light = a->g_lat_light;
  G.B = a->B; G.n_elem_blocks = (8 * 256 + 256 * 64) / 256;
  G.dz_max = f16 ? dz_max : nullptr;
  hipLaunchKernelGGL(mlp_wgrad_finalize2, dim3((unsigned)(G.n_elem_blocks + (a->B * 64 + 3) / 4)), dim3(256), 0, stream, G);
  return tp::check_launch("tp_mlp_bwd");
}
