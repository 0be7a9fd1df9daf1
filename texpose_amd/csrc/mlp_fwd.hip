// K2: fused positional encoding + static / transient / light MLP forward for gfx950 (CDNA4).
//
// Replaces NeRF.forward / forward_samples (reference layers/nerf_static_transient_light.py:76-166,
// camera.py:317-322): 16 Linear layers + concats + activations whose [S,256..334] intermediates
// the reference streams through memory.  Here a sample never leaves the register file between
// its 3 input coordinates and its 9 outputs.
//
// Design (MI355X-first, not a translation):
//  * exact-fp32 matrix cores: v_mfma_f32_32x32x2_f32 (bit-identical to an fmaf chain), needed for
//    the 1e-4 parity bar through 12 chained layers; roofline = 157 TFLOP/s.
//  * SAMPLES sit on the MFMA column/lane axis, FEATURES on the accumulator-register axis:
//    out[f][s] = sum_k W[f][k] h[k][s].  The weights are the A operand, and a layer's 8x(32x32)
//    accumulator tiles ARE the next layer's B operands (k-step (ts,r) contracts the two features
//    held by register r of tile ts in the two lane halves) -- no LDS round trip, no shuffles.
//  * one wave owns 32 samples x 256 features = 128 accumulator VGPRs + 128 for the previous
//    layer; 4 waves (one per SIMD, ~340 of the 512 unified VGPR/AGPRs) = 128 samples / workgroup.
//  * weights are pre-packed in exactly the order the kernel consumes them (mlp_layout.h) and are
//    streamed L2 -> LDS in 32 KiB chunks with global_load_lds_dwordx4 (LDS-DMA), double buffered:
//    the DMA for chunk c+1 is issued before the 128 MFMAs (8192 cycles) of chunk c, and one
//    __syncthreads (which drains vmcnt) per chunk hands the buffer over.  A fragments are read
//    with ds_read_b128 (4 fragments per read, conflict-free: lane-linear).
//  * 3.7 MB of packed weights stay L2 / Infinity-Cache resident; HBM sees 12..32 B in and 36 B
//    out per sample against 1.82 MFLOP of work.
//  * persistent workgroups (grid <= CUs) walk sample tiles; the trunk feature needed by both
//    heads is parked in a per-workgroup scratch slab (each lane re-reads only what it wrote).
#include <cstdlib>
#include "mlp_mma.h"

namespace {
using namespace tp_layout;

using namespace tp_mma;

constexpr int kTileSamples = 128;
constexpr int kBiasPad = (kBiasFloats + 63) / 64 * 64;
constexpr int kEncFloats = 32 * kThreads;                     // per-lane [x, PE(x)] B operands
constexpr int kExFloats = 40 * kThreads;                      // per-lane head extras (latents, view encoding)
// 2 x 32 KiB weight buffers + biases + the per-lane "extra input" B operands = 150 KiB of the 160 KiB LDS
constexpr int kLdsFloats = 2 * kChunkFloats + kBiasPad + kEncFloats + kExFloats;

// 1..5-row head: one chunk, 128 k-steps over all of `h`, single accumulator tile
__device__ __forceinline__ f32x16 part_head(Pipe& p, const f32x16 (&h)[8]) {
  f32x16 acc = {0};
  chunk_begin(p, kNumChunks);
  const float* l = chunk_ptr(p);
  f32x4 a = *reinterpret_cast<const f32x4*>(l);
  __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#pragma unroll
  for (int s4 = 0; s4 < 32; ++s4) {
    f32x4 n = a;
    if (s4 + 1 < 32) n = *reinterpret_cast<const f32x4*>(l + (s4 + 1) * 256);
    const f32x16 hv = h[s4 >> 2];
    acc = mfma(a.x, hv[(s4 & 3) * 4 + 0], acc);
    acc = mfma(a.y, hv[(s4 & 3) * 4 + 1], acc);
    acc = mfma(a.z, hv[(s4 & 3) * 4 + 2], acc);
    acc = mfma(a.w, hv[(s4 & 3) * 4 + 3], acc);
    if (s4 + 1 < 32) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
    a = n;
  }
  chunk_end(p, kNumChunks);
  return acc;
}

__device__ __forceinline__ float softplus(float x) { return x > 20.0f ? x : log1pf(expf(x)); }
__device__ __forceinline__ float sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

struct Params {
  const float* packed;
  const float* center; const float* ray; const float* depth;
  const float* points; const float* ray_unit;
  const float* lat_trans; const float* lat_light;
  int B, R, N;
  int64_t n_samples, n_tiles;
  float* rgb; float* density; float* uncert; float* saved; float* workspace;
  const float* density_noise;      // optional: added to the static density's pre-activation (nerf.density_noise_reg, train mode)
};

// this lane's sample of `tile` (lanes j and j + 32 of a wave hold sample wave * 32 + j), clamped into the call
__device__ __forceinline__ float density_noise_of(const Params& P, int64_t tile, int wave, int j) {
  if (P.density_noise == nullptr) return 0.0f;
  int64_t s = tile * 128 + wave * 32 + j;
  if (s >= P.n_samples) s = P.n_samples - 1;
  return P.density_noise[s];
}

template <bool SAVE>
__global__ __launch_bounds__(kThreads, 1) void mlp_fwd_kernel(Params P) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, hh = lane >> 5;
  float* bias_lds = lds + 2 * kChunkFloats;
  // Inputs that do not come from a previous layer are staged per lane in LDS ([k-step][thread]):
  // each lane reads back only what it wrote, so no barrier is involved, and the trigonometry runs
  // as a rolled loop instead of 42 inlined sinf/cosf expansions holding registers.
  float* enc_lds = bias_lds + kBiasPad + tid;
  float* ex_lds = bias_lds + kBiasPad + kEncFloats + tid;

  Pipe p;
  p.stream = P.packed; p.lds = lds; p.chunk = 0; p.buf = 0; p.wave = __builtin_amdgcn_readfirstlane(wave); p.lane = lane;
  // biases -> LDS once per workgroup; first weight chunk -> buffer 0
  for (int i = tid; i < kBiasFloats; i += kThreads) bias_lds[i] = P.packed[(size_t)kNumChunks * kChunkFloats + i];
  dma_chunk(p, 0, 0);
  __syncthreads();

  float* ws = P.workspace + (size_t)blockIdx.x * (kTileSamples * 256);

  for (int64_t tile = blockIdx.x; tile < P.n_tiles; tile += gridDim.x) {
    // ------------------------------------------------------------------ per-sample inputs
    const int64_t s_raw = tile * kTileSamples + wave * 32 + j;
    const bool live = s_raw < P.n_samples;
    const int64_t s = live ? s_raw : P.n_samples - 1;
    const int64_t q = s / P.N;              // ray
    const int b = (int)(q / P.R);           // image
    float x[3], vu[3];
    if (P.center != nullptr) {
      const float z = P.depth[s];
      float nrm = 0.f;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float d = P.ray[3 * q + c];
        x[c] = tp::add_rn(P.center[3 * q + c], tp::mul_rn(d, z));   // camera.py:321
        nrm = tp::add_rn(nrm, tp::mul_rn(d, d));
        vu[c] = d;
      }
      const float den = fmaxf(sqrtf(nrm), 1e-12f);                   // F.normalize (layers/...light.py:156)
#pragma unroll
      for (int c = 0; c < 3; ++c) vu[c] = tp::div_rn(vu[c], den);
    } else {
#pragma unroll
      for (int c = 0; c < 3; ++c) { x[c] = P.points[3 * s + c]; vu[c] = P.ray_unit[3 * s + c]; }
    }
    // positional encoding of x: this lane half evaluates sin (h=0) or cos (h=1) of x_c * 2^l * pi
#pragma nounroll
    for (int r = 0; r < 30; ++r) {
      const int c = r / 10;
      const float xc = c == 0 ? x[0] : (c == 1 ? x[1] : x[2]);
      const float arg = tp::mul_rn(xc, ldexpf(3.14159274101257324f, r - c * 10));
      enc_lds[r * kThreads] = tp::sincos_sel(arg, hh);
    }
    enc_lds[30 * kThreads] = hh ? x[1] : x[0];
    enc_lds[31 * kThreads] = hh ? 0.0f : x[2];

    f32x16 h[8], acc[8];
    float sig_s = 0.f, sig_t = 0.f, unc = 0.f, rgb_t[3] = {0.f, 0.f, 0.f}, rgb_s[3] = {0.f, 0.f, 0.f};

#pragma nounroll
    for (int li = 0; li < kNumWide; ++li) {
      if (li == L7) {   // static density = softplus(row 0 of mlp_feat.7) (layers/...light.py:94-98)
        const f32x16 a = part_head(p, h);
        sig_s = softplus(a[0] + bias_lds[kHeadBiasOff + 0] + density_noise_of(P, tile, wave, j));
      }
      if (li == R0) {   // bring the trunk feature back for the rgb head
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(ws + ((t * 4 + g) * kThreads + tid) * 4));
            h[t][g * 4 + 0] = v.x; h[t][g * 4 + 1] = v.y; h[t][g * 4 + 2] = v.z; h[t][g * 4 + 3] = v.w;
          }
      }
#pragma unroll
      for (int t = 0; t < 8; ++t) acc[t] = f32x16{0};

      if (li != L0) part_gen(p, acc, h, kNumChunks);

      if (li == L0 || li == L4) {          // [x, PE(x)] columns (layers/...light.py:81-82,90-91)
#pragma unroll
        for (int qd = 0; qd < 2; ++qd) {
          chunk_begin(p, kNumChunks);
          mma_wide<16, 3>(acc, chunk_ptr(p), [&](int s_) { return enc_lds[(qd * 16 + s_) * kThreads]; });
          chunk_end(p, kNumChunks);
        }
      } else if (li == T0) {               // transient latent (layers/...light.py:126-128)
#pragma unroll
        for (int r = 0; r < 8; ++r) ex_lds[r * kThreads] = P.lat_trans[b * 16 + r + 8 * hh];
        chunk_begin(p, kNumChunks);
        mma_wide<8, 3>(acc, chunk_ptr(p), [&](int s_) { return ex_lds[s_ * kThreads]; });
        chunk_end(p, kNumChunks);
      } else if (li == R0) {               // [ray_unit, PE(ray_unit), x, light] (layers/...light.py:104-117)
#pragma nounroll
        for (int r = 0; r < 12; ++r) {
          const int c = r >> 2;
          const float vc = c == 0 ? vu[0] : (c == 1 ? vu[1] : vu[2]);
          const float arg = tp::mul_rn(vc, ldexpf(3.14159274101257324f, r & 3));
          ex_lds[r * kThreads] = tp::sincos_sel(arg, hh);
        }
        ex_lds[12 * kThreads] = hh ? vu[1] : vu[0];
        ex_lds[13 * kThreads] = hh ? x[0] : vu[2];
        ex_lds[14 * kThreads] = hh ? x[2] : x[1];
        if (SAVE && live) {   // mlp_rgb.0 input columns 256..285 for the weight gradient
          float* sx = P.saved + (tile * 4 + wave) * (int64_t)kSavedGroupFloats + SV_EX * kBlockFloats;
#pragma unroll
          for (int r = 0; r < 15; ++r) sx[blk_off(x40_col(r, hh), j)] = ex_lds[r * kThreads];
        }
#pragma unroll
        for (int r = 15; r < 39; ++r) ex_lds[r * kThreads] = P.lat_light[b * 48 + (r - 15) + 24 * hh];
        ex_lds[39 * kThreads] = 0.0f;
#pragma unroll
        for (int qd = 0; qd < 3; ++qd) {
          chunk_begin(p, kNumChunks);
          if (qd < 2) mma_wide<16, 3>(acc, chunk_ptr(p), [&](int s_) { return ex_lds[(qd * 16 + s_) * kThreads]; });
          else mma_wide<8, 3>(acc, chunk_ptr(p), [&](int s_) { return ex_lds[(32 + s_) * kThreads]; });
          chunk_end(p, kNumChunks);
        }
      }

      // bias + ReLU; the result is the next layer's B operand
      const float* bl = bias_lds + (li * 2 + hh) * 128;
#pragma unroll
      for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 bv = *reinterpret_cast<const f32x4*>(bl + t * 16 + g * 4);
          h[t][g * 4 + 0] = fmaxf(acc[t][g * 4 + 0] + bv.x, 0.0f);
          h[t][g * 4 + 1] = fmaxf(acc[t][g * 4 + 1] + bv.y, 0.0f);
          h[t][g * 4 + 2] = fmaxf(acc[t][g * 4 + 2] + bv.z, 0.0f);
          h[t][g * 4 + 3] = fmaxf(acc[t][g * 4 + 3] + bv.w, 0.0f);
        }

      if (SAVE && li >= L7 && live) {
        // activations for the backward (layout: mlp_layout.h "Training record")
        float* grp = P.saved + (tile * 4 + wave) * (int64_t)kSavedGroupFloats;
        int o16[16];
        lane_block_offsets(j, hh, o16);
        store_block(grp + (li - L7) * kBlockFloats, h, o16);
        if (li >= T0) {   // ReLU sign bits for the dgrad kernel
          uint32_t* mk = reinterpret_cast<uint32_t*>(grp + kMaskOff) + (li - T0) * 256 + lane;
#pragma unroll
          for (int w = 0; w < 4; ++w) {
            uint32_t m = 0;
#pragma unroll
            for (int bt = 0; bt < 32; ++bt) m |= (h[2 * w + (bt >> 4)][bt & 15] > 0.0f ? 1u : 0u) << bt;
            mk[w * 64] = m;
          }
        }
      }
      if (li == L7) {   // park the trunk feature for the second head
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            f32x4 v = {h[t][g * 4 + 0], h[t][g * 4 + 1], h[t][g * 4 + 2], h[t][g * 4 + 3]};
            __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(ws + ((t * 4 + g) * kThreads + tid) * 4));   // streaming: keep the weights in L2
          }
      }
      if (li == T2) {   // transient head: rgb_t (sigmoid), sigma_t, uncert (softplus) (layers/...light.py:135-137)
        const f32x16 a = part_head(p, h);
        const float* hb = bias_lds + kHeadBiasOff + 1;
        rgb_t[0] = sigmoid(a[0] + hb[0]); rgb_t[1] = sigmoid(a[1] + hb[1]); rgb_t[2] = sigmoid(a[2] + hb[2]);
        sig_t = softplus(a[3] + hb[3]);
        unc = softplus(a[0] + hb[4]);       // row 4 lives in register 0 of the upper lane half
      }
      if (li == R2) {   // static rgb (layers/...light.py:122)
        const f32x16 a = part_head(p, h);
        const float* hb = bias_lds + kHeadBiasOff + 6;
        rgb_s[0] = sigmoid(a[0] + hb[0]); rgb_s[1] = sigmoid(a[1] + hb[1]); rgb_s[2] = sigmoid(a[2] + hb[2]);
      }
    }

    // ------------------------------------------------------------------ outputs (36 B / sample)
    if (live) {
      if (hh == 0) {
        float2* o = reinterpret_cast<float2*>(P.rgb + s * 6);
        o[0] = make_float2(rgb_s[0], rgb_t[0]);
        o[1] = make_float2(rgb_s[1], rgb_t[1]);
        o[2] = make_float2(rgb_s[2], rgb_t[2]);
        *reinterpret_cast<float2*>(P.density + s * 2) = make_float2(sig_s, sig_t);
      } else {
        P.uncert[s] = unc;
      }
    }
  }
}

// =====================================================================================================================
// Round 3: the same forward with the two accumulator sets pinned to the AGPR file and touched only by the generated blocks of
// fp32_asm.inc.h (gen_fp32_asm.py: register map, operand layout, hazards).  Layer li reads set (li even ? Q : P) as its B operand
// straight from the accumulators and writes set (li even ? P : Q); bias + ReLU rewrite a set in place.  The compiled code no
// longer holds h[8] / acc[8] (256 registers), so the trunk feature stays ON THE CU as an ordinary 128-register value between the
// end of the trunk and the colour head (the compiled kernel above parks it in a global slab: 141 GB of cache traffic per
// 480x640x128 image, profiles/traffic.json).  Same arithmetic, instruction for instruction: exact fp32 products, k order, bias
// added last -- bit-identical outputs (tests/test_gpu_parity.py::test_exact_fp32_asm_kernel_bit_identical_to_compiled).
// Inference only (no record): the recording variant of the exact kernel stays the compiled one.
// =====================================================================================================================
#include "fp32_asm.inc.h"

#define TP32_CLOB TP32_ACT_CLOBBERS, TP32_ALL_AGPRS, "memory"
#define TP32_GEN_CASE(NAME, TS) if constexpr (ts == TS) asm volatile(NAME##TS : : [a] "v"(a) : TP32_CLOB)
template <bool SRC_P, int ts>
__device__ __forceinline__ void asm32_gen(unsigned a) {
  if constexpr (SRC_P) {
    TP32_GEN_CASE(TP32_GEN_PQ_, 0); TP32_GEN_CASE(TP32_GEN_PQ_, 1); TP32_GEN_CASE(TP32_GEN_PQ_, 2); TP32_GEN_CASE(TP32_GEN_PQ_, 3);
    TP32_GEN_CASE(TP32_GEN_PQ_, 4); TP32_GEN_CASE(TP32_GEN_PQ_, 5); TP32_GEN_CASE(TP32_GEN_PQ_, 6); TP32_GEN_CASE(TP32_GEN_PQ_, 7);
  } else {
    TP32_GEN_CASE(TP32_GEN_QP_, 0); TP32_GEN_CASE(TP32_GEN_QP_, 1); TP32_GEN_CASE(TP32_GEN_QP_, 2); TP32_GEN_CASE(TP32_GEN_QP_, 3);
    TP32_GEN_CASE(TP32_GEN_QP_, 4); TP32_GEN_CASE(TP32_GEN_QP_, 5); TP32_GEN_CASE(TP32_GEN_QP_, 6); TP32_GEN_CASE(TP32_GEN_QP_, 7);
  }
}
// kind: 0 = 16 k-steps, first with C = 0; 1 = 16 k-steps; 2 = 8 k-steps
template <bool DST_P, int kind>
__device__ __forceinline__ void asm32_extra(unsigned a, unsigned b) {
  if constexpr (DST_P) {
    if constexpr (kind == 0) asm volatile(TP32_EXTRA16Z_P : : [a] "v"(a), [b] "v"(b) : TP32_CLOB);
    else if constexpr (kind == 1) asm volatile(TP32_EXTRA16_P : : [a] "v"(a), [b] "v"(b) : TP32_CLOB);
    else asm volatile(TP32_EXTRA8_P : : [a] "v"(a), [b] "v"(b) : TP32_CLOB);
  } else {
    if constexpr (kind == 0) asm volatile(TP32_EXTRA16Z_Q : : [a] "v"(a), [b] "v"(b) : TP32_CLOB);
    else if constexpr (kind == 1) asm volatile(TP32_EXTRA16_Q : : [a] "v"(a), [b] "v"(b) : TP32_CLOB);
    else asm volatile(TP32_EXTRA8_Q : : [a] "v"(a), [b] "v"(b) : TP32_CLOB);
  }
}
template <bool SET_P>
__device__ __forceinline__ void asm32_act(unsigned bl) {
  if constexpr (SET_P) asm volatile(TP32_ACT_P : : [bl] "v"(bl) : TP32_CLOB);
  else asm volatile(TP32_ACT_Q : : [bl] "v"(bl) : TP32_CLOB);
}
// training: bias + ReLU of a set AND its activation record (gen_fp32_asm.py: gen_act rec / mask).  `o` = the lane's four swizzled
// byte offsets inside a 4 KB record tile, `rbase` the record block of this wave's group, `mkoff` the byte offset (from rbase) of the
// lane's first ReLU sign word, `live` != 0 for lanes that own a sample of the call.
struct Rec32 { unsigned o[4]; unsigned mkoff; int live; };
template <bool SET_P, bool MASK>
__device__ __forceinline__ void asm32_act_rec(unsigned bl, const Rec32& r, const float* rbase) {
  unsigned mk = 0;
  const uint64_t rb = (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)rbase) |
                      ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((uintptr_t)rbase >> 32)) << 32);
#define TP32_ACT_REC(TXT)                                                                                              \
  asm volatile(TXT : [mk] "+v"(mk)                                                                                     \
               : [bl] "v"(bl), [o0] "v"(r.o[0]), [o1] "v"(r.o[1]), [o2] "v"(r.o[2]), [o3] "v"(r.o[3]), [mkoff] "v"(r.mkoff),   \
                 [live] "v"(r.live), [rbase] "s"(rb)                                                                   \
               : TP32_CLOB, TP32_REC_CLOBBERS)
  if constexpr (SET_P && MASK) TP32_ACT_REC(TP32_ACT_P_RECM);
  else if constexpr (SET_P) TP32_ACT_REC(TP32_ACT_P_REC);
  else if constexpr (MASK) TP32_ACT_REC(TP32_ACT_Q_RECM);
  else TP32_ACT_REC(TP32_ACT_Q_REC);
#undef TP32_ACT_REC
}
template <bool SET_P>
__device__ __forceinline__ f32x16 asm32_head(unsigned a) {
  f32x16 v;
  if constexpr (SET_P) asm volatile(TP32_HEAD_P : "={v[232:247]}"(v) : [a] "v"(a) : TP32_RING_CLOBBERS, TP32_ALL_AGPRS, "memory");
  else asm volatile(TP32_HEAD_Q : "={v[232:247]}"(v) : [a] "v"(a) : TP32_RING_CLOBBERS, TP32_ALL_AGPRS, "memory");
  return v;
}
#define TP32_SF_OUT(F)                                                                                                 \
  "={v[32:47]}"(F[0]), "={v[48:63]}"(F[1]), "={v[64:79]}"(F[2]), "={v[80:95]}"(F[3]), "={v[96:111]}"(F[4]),            \
      "={v[112:127]}"(F[5]), "={v[128:143]}"(F[6]), "={v[144:159]}"(F[7])
#define TP32_SF_IN(F)                                                                                                  \
  "{v[32:47]}"(F[0]), "{v[48:63]}"(F[1]), "{v[64:79]}"(F[2]), "{v[80:95]}"(F[3]), "{v[96:111]}"(F[4]),                 \
      "{v[112:127]}"(F[5]), "{v[128:143]}"(F[6]), "{v[144:159]}"(F[7])

__device__ __forceinline__ unsigned lds_addr(const float* p) { return (unsigned)(uintptr_t)AS3(p); }

// one 256 -> 256 part: 8 chunks, chunk ts contracts tile ts of the source set
template <bool SRC_P>
__device__ __forceinline__ void part_gen_asm(Pipe& p) {
#define TP32_STEP(TS) chunk_begin(p, kNumChunks); asm32_gen<SRC_P, TS>(lds_addr(chunk_ptr(p))); chunk_end(p, kNumChunks)
  TP32_STEP(0); TP32_STEP(1); TP32_STEP(2); TP32_STEP(3); TP32_STEP(4); TP32_STEP(5); TP32_STEP(6); TP32_STEP(7);
#undef TP32_STEP
}
template <bool SET_P>
__device__ __forceinline__ f32x16 part_head_asm(Pipe& p) {
  chunk_begin(p, kNumChunks);
  const f32x16 v = asm32_head<SET_P>(lds_addr(chunk_ptr(p)));
  chunk_end(p, kNumChunks);
  return v;
}
template <bool DST_P, int kind>
__device__ __forceinline__ void part_extra_asm(Pipe& p, const float* staged) {
  chunk_begin(p, kNumChunks);
  asm32_extra<DST_P, kind>(lds_addr(chunk_ptr(p)), lds_addr(staged));
  chunk_end(p, kNumChunks);
}

// SAVE (training): the ACT blocks of the seven recorded layers (trunk feature, T0..T2, R0..R2) also write the activation record and,
// for the six head layers, the ReLU sign words -- the compiled recording kernel above parks the trunk feature in a global slab and
// carries 132 scratch instructions; this one holds it in registers like the inference form.  Same arithmetic: records and outputs are
// bit-identical to the compiled kernel's (TP_FP32_CXX=1 selects that one; test_exact_fp32_asm_recording_bit_identical_to_compiled).
template <bool SAVE>
__global__ __launch_bounds__(kThreads, 1) void mlp_fwd_exact_asm_kernel(Params P) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, hh = lane >> 5;
  float* bias_lds = lds + 2 * kChunkFloats;
  float* enc_lds = bias_lds + kBiasPad + tid;                       // per-lane staging, [k-step][thread] (see the kernel above)
  float* ex_lds = bias_lds + kBiasPad + kEncFloats + tid;

  Pipe p;
  p.stream = P.packed; p.lds = lds; p.chunk = 0; p.buf = 0; p.wave = __builtin_amdgcn_readfirstlane(wave); p.lane = lane;
  for (int i = tid; i < kBiasFloats; i += kThreads) bias_lds[i] = P.packed[(size_t)kNumChunks * kChunkFloats + i];
  dma_chunk(p, 0, 0);
  __syncthreads();

  for (int64_t tile = blockIdx.x; tile < P.n_tiles; tile += gridDim.x) {
    const int64_t s_raw = tile * kTileSamples + wave * 32 + j;
    const bool live = s_raw < P.n_samples;
    const int64_t s = live ? s_raw : P.n_samples - 1;
    const int64_t q = s / P.N;
    const int b = (int)(q / P.R);
    float x[3], vu[3];
    if (P.center != nullptr) {
      const float z = P.depth[s];
      float nrm = 0.f;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float d = P.ray[3 * q + c];
        x[c] = tp::add_rn(P.center[3 * q + c], tp::mul_rn(d, z));
        nrm = tp::add_rn(nrm, tp::mul_rn(d, d));
        vu[c] = d;
      }
      const float den = fmaxf(sqrtf(nrm), 1e-12f);
#pragma unroll
      for (int c = 0; c < 3; ++c) vu[c] = tp::div_rn(vu[c], den);
    } else {
#pragma unroll
      for (int c = 0; c < 3; ++c) { x[c] = P.points[3 * s + c]; vu[c] = P.ray_unit[3 * s + c]; }
    }
#pragma nounroll
    for (int r = 0; r < 30; ++r) {
      const int c = r / 10;
      const float xc = c == 0 ? x[0] : (c == 1 ? x[1] : x[2]);
      const float arg = tp::mul_rn(xc, ldexpf(3.14159274101257324f, r - c * 10));
      enc_lds[r * kThreads] = tp::sincos_sel(arg, hh);
    }
    enc_lds[30 * kThreads] = hh ? x[1] : x[0];
    enc_lds[31 * kThreads] = hh ? 0.0f : x[2];
    // the colour head's extra inputs do not depend on the network: staged now, so that nothing but the blocks runs between the
    // trunk and the heads ([ray_unit, PE(ray_unit), x, light], layers/...light.py:104-117)
#pragma nounroll
    for (int r = 0; r < 12; ++r) {
      const int c = r >> 2;
      const float vc = c == 0 ? vu[0] : (c == 1 ? vu[1] : vu[2]);
      const float arg = tp::mul_rn(vc, ldexpf(3.14159274101257324f, r & 3));
      ex_lds[r * kThreads] = tp::sincos_sel(arg, hh);
    }
    ex_lds[12 * kThreads] = hh ? vu[1] : vu[0];
    ex_lds[13 * kThreads] = hh ? x[0] : vu[2];
    ex_lds[14 * kThreads] = hh ? x[2] : x[1];
#pragma unroll
    for (int r = 15; r < 39; ++r) ex_lds[r * kThreads] = P.lat_light[b * 48 + (r - 15) + 24 * hh];
    ex_lds[39 * kThreads] = 0.0f;
    // LDS addresses that depend on the lane are RE-DERIVED from the hardware thread id wherever they are used (an empty volatile
    // asm keeps the derivations apart), like the record operands below: carried through the tile next to the 128 registers of the
    // trunk feature, the recording variant's extra live values made the compiler spill into AGPRs, which belong to the blocks
    const auto fresh_tid = [] { int t_ = threadIdx.x; asm volatile("" : "+v"(t_)); return t_; };
    const auto bias_of = [&](int li) { return lds_addr(bias_lds) + (((unsigned)fresh_tid() >> 5) & 1u) * 512u + (unsigned)li * 1024u; };
    const auto enc_now = [&] { return bias_lds + kBiasPad + fresh_tid(); };
    const auto ex_now = [&] { return bias_lds + kBiasPad + kEncFloats + fresh_tid(); };
    if constexpr (SAVE) {
      if (live) {   // mlp_rgb.0 input columns 256..285 for the weight gradient
        float* sx = P.saved + (tile * 4 + wave) * (int64_t)kSavedGroupFloats + SV_EX * kBlockFloats;
#pragma unroll
        for (int r = 0; r < 15; ++r) sx[blk_off(x40_col(r, hh), j)] = ex_lds[r * kThreads];
      }
    }
    // The record operands of slot `sl` are RE-DERIVED from the hardware thread id at every recording block (behind an empty volatile
    // asm, so that nothing of it is hoisted): carried through the tile next to the 128 registers of the trunk feature they made the
    // compiler spill into AGPRs -- which belong to the blocks (check_asm_ownership.py fails the build on that).
    // Sign words of slot sl >= 1: kMaskOff + ((sl - 1) * 4 + w) * 64 + lane.
    const auto rec_now = [&](int sl, const float*& base) {
      int t_ = threadIdx.x;
      asm volatile("" : "+v"(t_));
      const unsigned ln = (unsigned)t_ & 63u, jj = ln & 31u, h_ = ln >> 5, q4 = jj >> 2;
      const int wv = __builtin_amdgcn_readfirstlane(t_ >> 6);
      Rec32 r;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const unsigned xv = ((unsigned)(c & 1) + 4u * (unsigned)(c >> 1) + 2u * h_) & 7u;   // (f >> 1) & 7 of the registers of class c
        r.o[c] = ((((q4 ^ xv) << 2) | (jj & 3u)) + 128u * h_) * 4u;
      }
      r.live = (tile * kTileSamples + wv * 32 + (int64_t)jj) < P.n_samples ? 1 : 0;
      r.mkoff = (unsigned)((kMaskOff + ((sl > 0 ? sl : 1) - 1) * 256 - sl * kBlockFloats) * 4) + ln * 4u;
      base = P.saved + (tile * 4 + wv) * (int64_t)kSavedGroupFloats + (int64_t)sl * kBlockFloats;
      return r;
    };

    // ---- trunk: L0 (extras only) .. L7; set parity: even layers write P, odd layers write Q
    part_extra_asm<true, 0>(p, enc_now());
    part_extra_asm<true, 1>(p, enc_now() + 16 * kThreads);
    asm32_act<true>(bias_of(L0));
#pragma nounroll
    for (int li = L1; li <= L6; li += 2) {                                      // (L1, L2), (L3, L4), (L5, L6)
      part_gen_asm<true>(p);
      asm32_act<false>(bias_of(li));
      part_gen_asm<false>(p);
      if (li + 1 == L4) {                                                       // [x, PE(x)] again (layers/...light.py:90-91)
        part_extra_asm<true, 1>(p, enc_now());
        part_extra_asm<true, 1>(p, enc_now() + 16 * kThreads);
      }
      asm32_act<true>(bias_of(li + 1));
    }
    const f32x16 hs = part_head_asm<true>(p);                                   // static density = softplus(row 0 of mlp_feat.7)
    float dn_ = 0.0f;
    if (P.density_noise != nullptr) { const int t_ = fresh_tid(); dn_ = density_noise_of(P, tile, t_ >> 6, t_ & 31); }
    const float sig_s = softplus(hs[0] + bias_lds[kHeadBiasOff + 0] + dn_);
    part_gen_asm<true>(p);                                                      // L7 -> Q
    if constexpr (SAVE) { const float* rb_; const Rec32 rc_ = rec_now(SV_FEAT, rb_); asm32_act_rec<false, false>(bias_of(L7), rc_, rb_); }
    else asm32_act<false>(bias_of(L7));
    f32x16 F[8];
    asm volatile(TP32_STASH_Q : TP32_SF_OUT(F) : : TP32_ALL_AGPRS);            // the trunk feature, held until the colour head

    // ---- transient head: T0 (Q + latent -> P), T1 (-> Q), T2 (-> P)
    {
      float* lat_lds = enc_now();                                               // (the [x, PE(x)] rows are dead after L4)
      const int t_ = fresh_tid();
      const int64_t sl_ = tile * kTileSamples + (t_ >> 6) * 32 + (t_ & 31);
      const int b_ = (int)(((sl_ < P.n_samples ? sl_ : P.n_samples - 1) / P.N) / P.R);
#pragma unroll
      for (int r = 0; r < 8; ++r) lat_lds[r * kThreads] = P.lat_trans[b_ * 16 + r + 8 * ((t_ >> 5) & 1)];
    }
    part_gen_asm<false>(p);
    part_extra_asm<true, 2>(p, enc_now());
    if constexpr (SAVE) { const float* rb_; const Rec32 rc_ = rec_now(SV_T0, rb_); asm32_act_rec<true, true>(bias_of(T0), rc_, rb_); }
    else asm32_act<true>(bias_of(T0));
    part_gen_asm<true>(p);
    if constexpr (SAVE) { const float* rb_; const Rec32 rc_ = rec_now(SV_T1, rb_); asm32_act_rec<false, true>(bias_of(T1), rc_, rb_); }
    else asm32_act<false>(bias_of(T1));
    part_gen_asm<false>(p);
    if constexpr (SAVE) { const float* rb_; const Rec32 rc_ = rec_now(SV_T2, rb_); asm32_act_rec<true, true>(bias_of(T2), rc_, rb_); }
    else asm32_act<true>(bias_of(T2));
    const f32x16 ht = part_head_asm<true>(p);
    const float* hb = bias_lds + kHeadBiasOff + 1;
    const float rgb_t0 = sigmoid(ht[0] + hb[0]), rgb_t1 = sigmoid(ht[1] + hb[1]), rgb_t2 = sigmoid(ht[2] + hb[2]);
    const float sig_t = softplus(ht[3] + hb[3]);
    const float unc = softplus(ht[0] + hb[4]);                                  // row 4 lives in register 0 of the upper lane half

    // ---- colour head: R0 (feature restored into P, + extras -> Q), R1 (-> P), R2 (-> Q)
    asm volatile(TP32_RESTORE_P : : TP32_SF_IN(F) : TP32_ALL_AGPRS);
    part_gen_asm<true>(p);
    part_extra_asm<false, 1>(p, ex_now());
    part_extra_asm<false, 1>(p, ex_now() + 16 * kThreads);
    part_extra_asm<false, 2>(p, ex_now() + 32 * kThreads);
    if constexpr (SAVE) { const float* rb_; const Rec32 rc_ = rec_now(SV_R0, rb_); asm32_act_rec<false, true>(bias_of(R0), rc_, rb_); }
    else asm32_act<false>(bias_of(R0));
    part_gen_asm<false>(p);
    if constexpr (SAVE) { const float* rb_; const Rec32 rc_ = rec_now(SV_R1, rb_); asm32_act_rec<true, true>(bias_of(R1), rc_, rb_); }
    else asm32_act<true>(bias_of(R1));
    part_gen_asm<true>(p);
    if constexpr (SAVE) { const float* rb_; const Rec32 rc_ = rec_now(SV_R2, rb_); asm32_act_rec<false, true>(bias_of(R2), rc_, rb_); }
    else asm32_act<false>(bias_of(R2));
    const f32x16 hr = part_head_asm<false>(p);
    const float* hc = bias_lds + kHeadBiasOff + 6;
    const float rgb_s0 = sigmoid(hr[0] + hc[0]), rgb_s1 = sigmoid(hr[1] + hc[1]), rgb_s2 = sigmoid(hr[2] + hc[2]);

    const int to_ = fresh_tid();
    const int64_t so_ = tile * kTileSamples + (to_ >> 6) * 32 + (to_ & 31);
    if (so_ < P.n_samples) {   // streaming stores: the 3.7 MB weight stream is what should stay in the 4 MB L2, not 36 B per sample of outputs
      using f32x2 = __attribute__((ext_vector_type(2))) float;
      const int64_t s = so_;
      if (((to_ >> 5) & 1) == 0) {
        f32x2* o = reinterpret_cast<f32x2*>(P.rgb + s * 6);
        __builtin_nontemporal_store(f32x2{rgb_s0, rgb_t0}, o);
        __builtin_nontemporal_store(f32x2{rgb_s1, rgb_t1}, o + 1);
        __builtin_nontemporal_store(f32x2{rgb_s2, rgb_t2}, o + 2);
        __builtin_nontemporal_store(f32x2{sig_s, sig_t}, reinterpret_cast<f32x2*>(P.density + s * 2));
      } else {
        __builtin_nontemporal_store(unc, P.uncert + s);
      }
    }
  }
}

// standalone positional encoding (API parity with NeRF.positional_encoding)
__global__ void posenc_kernel(const float* __restrict__ x, int64_t n, int C, int L, float* __restrict__ out) {
  const int64_t total = n * C * 2 * L;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int l = (int)(e % L), sc = (int)((e / L) % 2), c = (int)((e / (2 * L)) % C);
    const int64_t i = e / (2 * L * C);
    const float arg = tp::mul_rn(x[i * C + c], ldexpf(3.14159274101257324f, l));
    out[e] = tp::sincos_sel(arg, sc);
  }
}

int persistent_grid(int64_t n_tiles) {
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
  return (int)(n_tiles < cus ? n_tiles : cus);
}

}  // namespace

int tp_launch_mlp_fwd_f16x3(const tp_mlp_fwd_args* a, int grid, hipStream_t stream);   // mlp_fwd_f16x3.hip

extern "C" size_t tp_mlp_workspace_bytes(int64_t n_samples) {
  const int64_t tiles = (n_samples + kTileSamples - 1) / kTileSamples;
  const int64_t wgs = tiles < 1024 ? tiles : 1024;   // upper bound on the persistent grid
  return (size_t)(wgs > 0 ? wgs : 1) * kTileSamples * 256 * sizeof(float);
}

extern "C" size_t tp_mlp_saved_bytes(int64_t n_samples) {
  const int64_t groups = ((n_samples + kTileSamples - 1) / kTileSamples) * 4;
  return (size_t)groups * kSavedGroupFloats * sizeof(float);
}

extern "C" int tp_mlp_fwd(const tp_mlp_fwd_args* a, tp_stream_t stream) {
  TP_REQUIRE(a && a->packed && a->lat_trans && a->lat_light && a->rgb && a->density && a->uncert && a->workspace,
             "null pointer");
  TP_REQUIRE(a->B > 0 && a->R > 0 && a->N > 0, "bad sizes");
  TP_REQUIRE((a->center && a->ray && a->depth) || (a->points && a->ray_unit), "need (center,ray,depth) or (points,ray_unit)");
  TP_REQUIRE(a->precision == TP_MLP_FP32 || a->precision == TP_MLP_F16X3, "unknown precision");
  if (a->precision == TP_MLP_F16X3) {
    const int64_t tiles = ((int64_t)a->B * a->R * a->N + kTileSamples - 1) / kTileSamples;
    return tp_launch_mlp_fwd_f16x3(a, persistent_grid(tiles), (hipStream_t)stream);
  }
  Params P;
  P.packed = (const float*)a->packed;
  P.center = a->center; P.ray = a->ray; P.depth = a->depth; P.points = a->points; P.ray_unit = a->ray_unit;
  P.lat_trans = a->lat_trans; P.lat_light = a->lat_light;
  P.B = a->B; P.R = a->R; P.N = a->N;
  P.n_samples = (int64_t)a->B * a->R * a->N;
  P.n_tiles = (P.n_samples + kTileSamples - 1) / kTileSamples;
  P.rgb = a->rgb; P.density = a->density; P.uncert = a->uncert; P.saved = a->saved; P.workspace = (float*)a->workspace;
  P.density_noise = a->density_noise;
  static unsigned long long attr_devices = 0;
  if (tp::first_use_on_device(attr_devices)) {
    hipError_t e = hipFuncSetAttribute((const void*)mlp_fwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       kLdsFloats * (int)sizeof(float));
    if (e == hipSuccess)
      e = hipFuncSetAttribute((const void*)mlp_fwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              kLdsFloats * (int)sizeof(float));
    if (e == hipSuccess)
      e = hipFuncSetAttribute((const void*)mlp_fwd_exact_asm_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              kLdsFloats * (int)sizeof(float));
    if (e == hipSuccess)
      e = hipFuncSetAttribute((const void*)mlp_fwd_exact_asm_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              kLdsFloats * (int)sizeof(float));
    if (e != hipSuccess) { tp::set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
  }
  const int grid = persistent_grid(P.n_tiles);
  // the block kernels (trunk feature on the CU), inference and recording; TP_FP32_CXX=1 selects the compiled kernels (A/B, bit-identity tests)
  const char* env_cxx = getenv("TP_FP32_CXX");                      // (read per call: a test switches it inside one process)
  const bool exact_cxx = env_cxx != nullptr && env_cxx[0] == '1';
  if (P.saved != nullptr && exact_cxx)
    hipLaunchKernelGGL(mlp_fwd_kernel<true>, dim3(grid), dim3(kThreads), kLdsFloats * sizeof(float), (hipStream_t)stream, P);
  else if (P.saved != nullptr)
    hipLaunchKernelGGL(mlp_fwd_exact_asm_kernel<true>, dim3(grid), dim3(kThreads), kLdsFloats * sizeof(float), (hipStream_t)stream, P);
  else if (exact_cxx)
    hipLaunchKernelGGL(mlp_fwd_kernel<false>, dim3(grid), dim3(kThreads), kLdsFloats * sizeof(float), (hipStream_t)stream, P);
  else
    hipLaunchKernelGGL(mlp_fwd_exact_asm_kernel<false>, dim3(grid), dim3(kThreads), kLdsFloats * sizeof(float), (hipStream_t)stream, P);
  return tp::check_launch("tp_mlp_fwd");
}

extern "C" int tp_posenc(const float* x, int64_t n, int C, int L, float* out, tp_stream_t stream) {
  TP_REQUIRE(x && out && C > 0 && L > 0 && n >= 0, "bad arguments");
  if (n == 0) return 0;
  int64_t blocks = (n * C * 2 * L + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(posenc_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, n, C, L, out);
  return tp::check_launch("tp_posenc");
}
