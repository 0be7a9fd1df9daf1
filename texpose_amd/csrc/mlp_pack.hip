// Packs the reference state-dict weights (W [out,in]) into the MFMA-fragment-ordered,
// execution-ordered stream consumed by mlp_fwd.hip (layout: mlp_layout.h).  One thread per packed
// float; the trunk part is packed once (frozen, reference layers/...light.py:34,236-239), the
// heads are re-packed after every optimiser step (442k floats).
#include "tp_common.h"
#include "mlp_layout.h"

namespace {
using namespace tp_layout;

struct W {
  const float* w[16];
  const float* b[16];
};

__global__ void pack_kernel(W w, int chunk0, int chunk1, float* __restrict__ out) {
  const int64_t n = (int64_t)(chunk1 - chunk0) * kChunkFloats;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    const int c = chunk0 + (int)(e / kChunkFloats), idx = (int)(e % kChunkFloats);
    const ChunkDesc d = chunk_desc(c);
    int row, col;
    chunk_src(d, idx, row, col);
    out[(int64_t)c * kChunkFloats + idx] = row < 0 ? 0.0f : w.w[d.mat][(int64_t)row * d.in_dim + col];
  }
}

// f16x3 stream (mlp_layout.h): hi = fp16(W * 2^shift), lo = fp16(W * 2^shift - hi)
__global__ void pack16_kernel(W w, int chunk0, int chunk1, _Float16* __restrict__ out, int rb) {
  const int64_t n = (int64_t)(chunk1 - chunk0) * kChunkHalves;
  const float scale = (float)(1 << kF16WeightShift);
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    const int c = chunk0 + (int)(e / kChunkHalves), idx = (int)(e % kChunkHalves);
    const ChunkDesc d = rb ? chunk_desc_rb(c) : chunk_desc(c);
    int part, row, col;
    chunk16_src(d, idx, part, row, col);
    float v = row < 0 ? 0.0f : w.w[d.mat][(int64_t)row * d.in_dim + col] * scale;
    const _Float16 hi = (_Float16)v;
    out[(int64_t)c * kChunkHalves + idx] = part == 0 ? hi : (_Float16)(v - (float)hi);
  }
}

// ray-bias stream: the transposed fp32 columns of mlp_rgb.0 / mlp_trans.0 that the per-ray bias pre-kernel contracts (mlp_layout.h)
__global__ void pack_rb_aux_kernel(W w, float* __restrict__ out, int trunk, int heads) {
  float* aux = out + kRbAuxOff;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= kRbAuxFloats) return;
  if (e >= kRbAuxHeads) {                            // rows of the narrow output layers (the density row belongs to the trunk)
    int mat, row, col;
    rb_head_src(e - kRbAuxHeads, mat, row, col);
    if (mat == W_FEAT0 + 7 ? trunk : heads) aux[e] = w.w[mat][(int64_t)row * 256 + col];
    return;
  }
  if (!heads) return;
  const int c = e >> 8, f = e & 255;
  if (c < 27) aux[e] = w.w[W_RGB0][(int64_t)f * 334 + 256 + c];
  else if (c < 75) aux[e] = w.w[W_RGB0][(int64_t)f * 334 + 286 + (c - 27)];
  else aux[e] = w.w[W_TRANS0][(int64_t)f * 272 + 256 + (c - 75)];
}

__global__ void pack_bias_kernel(W w, int wide0, int wide1, float* __restrict__ out, int nch) {
  float* bias = out + (int64_t)nch * kChunkFloats;
  const int n = (wide1 - wide0) * 256;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < n) {
    const int wide = wide0 + e / 256, rem = e % 256, h = rem / 128, t = (rem / 16) % 8, r = rem % 16;
    const int mat = wide <= L7 ? W_FEAT0 + wide : (wide <= T2 ? W_TRANS0 + (wide - T0) : W_RGB0 + (wide - R0));
    const int off = wide == L7 ? 1 : 0;
    bias[bias_index(wide, h, t, r)] = w.b[mat][off + feat_of(t, r, h)];
  }
  if (e < 16) {
    float v = 0.0f;
    if (e == 0) { if (wide0 <= L7) v = w.b[W_FEAT0 + 7][0]; else return; }
    else if (e < 6) { if (wide1 > T0) v = w.b[W_TRANS0 + 3][e - 1]; else return; }
    else if (e < 9) { if (wide1 > T0) v = w.b[W_RGB0 + 3][e - 6]; else return; }
    bias[kHeadBiasOff + e] = v;
  }
}
// Training: everything that has to be rebuilt from the head weights after an optimiser step, in ONE launch -- the f16x3 head chunks of
// the forward stream (pack16_kernel's part), the head biases (pack_bias_kernel's), and the transposed f16x3 image the data-gradient
// kernel streams (packT16_kernel of mlp_fwd_f16x3.hip: same arithmetic).  Block ranges: [0, kBiasBlocks) biases,
// [kBiasBlocks, kBiasBlocks + kTBlocks) transposed image, the rest the forward chunks.
constexpr int kBiasBlocks = ((kNumWide - kFirstHeadWide) * 256 + 255) / 256;
constexpr int kTBlocks = 512, kFwdBlocks = 1024;
__global__ void pack_heads16_kernel(W w, _Float16* __restrict__ out, _Float16* __restrict__ out_t) {
  const float scale = (float)(1 << kF16WeightShift);
  if ((int)blockIdx.x < kBiasBlocks) {
    float* bias = reinterpret_cast<float*>(out) + (int64_t)kNumChunks * kChunkFloats;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < (kNumWide - kFirstHeadWide) * 256) {
      const int wide = kFirstHeadWide + e / 256, rem = e % 256, h = rem / 128, t = (rem / 16) % 8, r = rem % 16;
      const int mat = wide <= T2 ? W_TRANS0 + (wide - T0) : W_RGB0 + (wide - R0);
      bias[bias_index(wide, h, t, r)] = w.b[mat][feat_of(t, r, h)];
    }
    if (e >= 1 && e < 9) bias[kHeadBiasOff + e] = e < 6 ? w.b[W_TRANS0 + 3][e - 1] : w.b[W_RGB0 + 3][e - 6];
    return;
  }
  if ((int)blockIdx.x < kBiasBlocks + kTBlocks) {
    if (out_t == nullptr) return;
    const int64_t n = (int64_t)kNumChunksT * kChunkHalves;
    for (int64_t e = (int64_t)(blockIdx.x - kBiasBlocks) * blockDim.x + threadIdx.x; e < n; e += (int64_t)kTBlocks * blockDim.x) {
      int part, mat, o, f;
      chunkT16_src((int)(e / kChunkHalves), (int)(e % kChunkHalves), part, mat, o, f);
      const float v = o < 0 ? 0.0f : w.w[mat][(int64_t)o * 256 + f] * scale;
      const _Float16 hi = (_Float16)v;
      out_t[e] = part == 0 ? hi : (_Float16)(v - (float)hi);
    }
    return;
  }
  const int64_t n = (int64_t)(kNumChunks - kFirstHeadChunk) * kChunkHalves;
  for (int64_t e = (int64_t)(blockIdx.x - kBiasBlocks - kTBlocks) * blockDim.x + threadIdx.x; e < n; e += (int64_t)kFwdBlocks * blockDim.x) {
    const int c = kFirstHeadChunk + (int)(e / kChunkHalves), idx = (int)(e % kChunkHalves);
    const ChunkDesc d = chunk_desc(c);
    int part, row, col;
    chunk16_src(d, idx, part, row, col);
    float v = row < 0 ? 0.0f : w.w[d.mat][(int64_t)row * d.in_dim + col] * scale;
    const _Float16 hi = (_Float16)v;
    out[(int64_t)c * kChunkHalves + idx] = part == 0 ? hi : (_Float16)(v - (float)hi);
  }
}
}  // namespace

// Host-side packer (same layout functions, no GPU involved): used by the CPU tests to check the
// stream layout against the oracle, and as the reference for the device packer.
extern "C" int tp_mlp_pack_host(const tp_mlp_weights* p, float* out) {
  TP_REQUIRE(p && out, "null pointer");
  const float* w[16]; const float* b[16];
  for (int i = 0; i < 8; ++i) { w[W_FEAT0 + i] = p->feat_w[i]; b[W_FEAT0 + i] = p->feat_b[i]; }
  for (int i = 0; i < 4; ++i) {
    w[W_RGB0 + i] = p->rgb_w[i]; b[W_RGB0 + i] = p->rgb_b[i];
    w[W_TRANS0 + i] = p->trans_w[i]; b[W_TRANS0 + i] = p->trans_b[i];
  }
  for (int i = 0; i < 16; ++i) TP_REQUIRE(w[i] && b[i], "null weight pointer");
  for (int c = 0; c < kNumChunks; ++c) {
    const ChunkDesc d = chunk_desc(c);
    for (int idx = 0; idx < kChunkFloats; ++idx) {
      int row, col;
      chunk_src(d, idx, row, col);
      out[(int64_t)c * kChunkFloats + idx] = row < 0 ? 0.0f : w[d.mat][(int64_t)row * d.in_dim + col];
    }
  }
  float* bias = out + (int64_t)kNumChunks * kChunkFloats;
  for (int wide = 0; wide < kNumWide; ++wide) {
    const int mat = wide <= L7 ? W_FEAT0 + wide : (wide <= T2 ? W_TRANS0 + (wide - T0) : W_RGB0 + (wide - R0));
    const int off = wide == L7 ? 1 : 0;
    for (int h = 0; h < 2; ++h)
      for (int t = 0; t < 8; ++t)
        for (int r = 0; r < 16; ++r) bias[bias_index(wide, h, t, r)] = b[mat][off + feat_of(t, r, h)];
  }
  for (int e = 0; e < 16; ++e) bias[kHeadBiasOff + e] = 0.0f;
  bias[kHeadBiasOff + 0] = b[W_FEAT0 + 7][0];
  for (int e = 0; e < 5; ++e) bias[kHeadBiasOff + 1 + e] = b[W_TRANS0 + 3][e];
  for (int e = 0; e < 3; ++e) bias[kHeadBiasOff + 6 + e] = b[W_RGB0 + 3][e];
  return 0;
}

extern "C" size_t tp_mlp_packed_bytes(void) { return (size_t)tp_layout::kPackedFloats * sizeof(float); }

extern "C" int tp_mlp_pack_heads_f16x3(const tp_mlp_weights* p, void* packed, void* packed_t, tp_stream_t stream) {
  TP_REQUIRE(p && packed, "null pointer");
  W w;
  for (int i = 0; i < 16; ++i) { w.w[i] = nullptr; w.b[i] = nullptr; }
  for (int i = 0; i < 4; ++i) {
    w.w[W_RGB0 + i] = p->rgb_w[i]; w.b[W_RGB0 + i] = p->rgb_b[i];
    w.w[W_TRANS0 + i] = p->trans_w[i]; w.b[W_TRANS0 + i] = p->trans_b[i];
    TP_REQUIRE(p->rgb_w[i] && p->rgb_b[i] && p->trans_w[i] && p->trans_b[i], "null weight pointer");
  }
  hipLaunchKernelGGL(pack_heads16_kernel, dim3(kBiasBlocks + kTBlocks + kFwdBlocks), dim3(256), 0, (hipStream_t)stream, w, (_Float16*)packed,
                     (_Float16*)packed_t);
  return tp::check_launch("tp_mlp_pack_heads_f16x3");
}

extern "C" int tp_mlp_pack(const tp_mlp_weights* p, int parts, void* packed, tp_stream_t stream) {
  TP_REQUIRE(p && packed, "null pointer");
  TP_REQUIRE((parts & ~(TP_PACK_ALL | TP_PACK_F16X3 | TP_PACK_RAYBIAS)) == 0 && (parts & TP_PACK_ALL) != 0, "bad parts mask");
  const int rb = (parts & TP_PACK_RAYBIAS) ? 1 : 0;
  TP_REQUIRE(!rb || (parts & TP_PACK_F16X3), "TP_PACK_RAYBIAS is a variant of the TP_PACK_F16X3 stream");
  W w;
  for (int i = 0; i < 8; ++i) { w.w[W_FEAT0 + i] = p->feat_w[i]; w.b[W_FEAT0 + i] = p->feat_b[i]; }
  for (int i = 0; i < 4; ++i) {
    w.w[W_RGB0 + i] = p->rgb_w[i]; w.b[W_RGB0 + i] = p->rgb_b[i];
    w.w[W_TRANS0 + i] = p->trans_w[i]; w.b[W_TRANS0 + i] = p->trans_b[i];
  }
  const bool trunk = parts & TP_PACK_TRUNK, heads = parts & TP_PACK_HEADS;
  for (int i = 0; i < 16; ++i) {
    const bool is_trunk = i < 8;
    if ((is_trunk && trunk) || (!is_trunk && heads)) TP_REQUIRE(w.w[i] && w.b[i], "null weight pointer");
  }
  const int nch = rb ? kNumChunksRB : kNumChunks;
  const int first_head = rb ? kFirstHeadChunkRB : kFirstHeadChunk;
  const int c0 = trunk ? 0 : first_head, c1 = heads ? nch : first_head;
  const int w0 = trunk ? 0 : kFirstHeadWide, w1 = heads ? kNumWide : kFirstHeadWide;
  if (parts & TP_PACK_F16X3)
    hipLaunchKernelGGL(pack16_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, w, c0, c1, (_Float16*)packed, rb);
  else
    hipLaunchKernelGGL(pack_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, w, c0, c1, (float*)packed);
  hipLaunchKernelGGL(pack_bias_kernel, dim3(((w1 - w0) * 256 + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, w0,
                     w1, (float*)packed, nch);
  if (rb)
    hipLaunchKernelGGL(pack_rb_aux_kernel, dim3((kRbAuxFloats + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, (float*)packed,
                       trunk ? 1 : 0, heads ? 1 : 0);
  return tp::check_launch("tp_mlp_pack");
}
