// Packed weight-stream layout shared by the pack kernel, the fused MLP forward and its tests.
//
// The fused MLP (mlp_fwd.hip) keeps activations in registers with SAMPLES on the MFMA column
// (lane) axis and FEATURES on the accumulator-register axis, so the weight matrix is always the
// A operand of v_mfma_f32_32x32x2_f32 and one layer's output registers are directly the next
// layer's B operand.  Consequences for the layout:
//
//  * accumulator tile t (32 features), register r (0..15), lane half h (lane>>5) holds feature
//        feat_of(t, r, h) = 32 t + (r & 3) + 8 (r >> 2) + 4 h            (MFMA C/D map)
//    so k-step (ts, r) of the NEXT layer contracts features feat_of(ts, r, 0|1), and the A
//    fragment for output tile t must hold  W[32 t + (lane & 31)][feat_of(ts, r, lane >> 5)].
//  * the whole network is consumed as a linear stream of 32 KiB chunks in execution order, each
//    chunk = 16 k-steps x 8 output tiles x 64 lanes (wide layers) or 128 k-steps x 1 tile
//    (3/5/1-row heads); inside a chunk four fragments are interleaved per lane so that one
//    ds_read_b128 feeds four MFMAs:   float index = ((kstep*2 + t/4)*64 + lane)*4 + t%4
//    (heads: ((kstep/4)*64 + lane)*4 + kstep%4).
//  * per-sample inputs that are not produced by a previous layer (positional encodings, raw
//    coordinates, latents) live in "extra" k-steps whose (k-step, half) -> input-column maps are
//    enc_col / x40_col / x8_col below.
//
// Reference: layers/nerf_static_transient_light.py:16-61 (layer shapes), :76-145 (concat orders).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define TP_HD __host__ __device__ __forceinline__
#else
#define TP_HD inline
#endif

namespace tp_layout {

constexpr int kChunkFloats = 8192;           // 32 KiB
constexpr int kNumChunks = 115;
constexpr int kNumWide = 14;                 // wide (256-out) layers: L0..L7, T0..T2, R0..R2
constexpr int kBiasFloats = kNumWide * 256 + 16;
constexpr int64_t kPackedFloats = (int64_t)kNumChunks * kChunkFloats + kBiasFloats;
constexpr int kFirstHeadChunk = 61;          // chunks >= this belong to the trainable heads (T*, R*)
constexpr int kFirstHeadWide = 8;            // wide-layer index of T0

// wide-layer indices
enum { L0 = 0, L1, L2, L3, L4, L5, L6, L7, T0, T1, T2, R0, R1, R2 };
// weight-matrix ids
enum { W_FEAT0 = 0, W_RGB0 = 8, W_TRANS0 = 12 };

enum ChunkKind { CK_GEN = 0, CK_ENC = 1, CK_X8 = 2, CK_X40 = 3, CK_HEAD = 4, CK_RBX = 5 };

struct ChunkDesc {
  int kind;     // ChunkKind
  int mat;      // weight matrix id (W_FEAT0 + i, W_RGB0 + i, W_TRANS0 + i)
  int sub;      // GEN: ts (0..7); ENC/X40: 16-k-step block index; HEAD: number of valid rows
  int row_off;  // first output row (1 for L7: row 0 is the density head)
  int col_off;  // first input column of this part
  int in_dim;   // row length of the weight matrix
};

TP_HD int feat_of(int t, int r, int h) { return 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h; }

// positional-encoded 3D point, 32 k-steps: input vector = [x(3), PE(x) (c*20 + s*10 + l)], 63 wide
TP_HD int enc_col(int r, int h) {
  if (r < 30) return 3 + (r / 10) * 20 + h * 10 + (r % 10);
  if (r == 30) return h;            // x[0], x[1]
  return h == 0 ? 2 : -1;           // x[2], pad
}
// rgb-head extras, 40 k-steps: [ray_unit(3), PE(ray_unit) (c*8+s*4+l) (24), x(3), light(48)], 78 wide
TP_HD int x40_col(int r, int h) {
  if (r < 12) return 3 + (r / 4) * 8 + h * 4 + (r % 4);
  if (r == 12) return h;                    // ray_unit[0], ray_unit[1]
  if (r == 13) return h == 0 ? 2 : 27;      // ray_unit[2], x[0]
  if (r == 14) return 28 + h;               // x[1], x[2]
  if (r < 39) return 30 + (r - 15) + 24 * h;  // light[a], light[24+a]
  return -1;
}
// transient-head extras, 8 k-steps: trans latent (16)
TP_HD int x8_col(int r, int h) { return r < 8 ? r + 8 * h : -1; }

TP_HD ChunkDesc chunk_desc(int c) {
  // schedule (must match mlp_fwd.hip): see the table in DESIGN.md
  if (c < 2) return {CK_ENC, W_FEAT0 + 0, c, 0, 0, 63};
  if (c < 26) { int l = 1 + (c - 2) / 8; return {CK_GEN, W_FEAT0 + l, (c - 2) % 8, 0, 0, 256}; }
  if (c < 34) return {CK_GEN, W_FEAT0 + 4, c - 26, 0, 0, 319};
  if (c < 36) return {CK_ENC, W_FEAT0 + 4, c - 34, 0, 256, 319};
  if (c < 52) { int l = 5 + (c - 36) / 8; return {CK_GEN, W_FEAT0 + l, (c - 36) % 8, 0, 0, 256}; }
  if (c == 52) return {CK_HEAD, W_FEAT0 + 7, 1, 0, 0, 256};
  if (c < 61) return {CK_GEN, W_FEAT0 + 7, c - 53, 1, 0, 256};
  if (c < 69) return {CK_GEN, W_TRANS0 + 0, c - 61, 0, 0, 272};
  if (c == 69) return {CK_X8, W_TRANS0 + 0, 0, 0, 256, 272};
  if (c < 86) { int l = 1 + (c - 70) / 8; return {CK_GEN, W_TRANS0 + l, (c - 70) % 8, 0, 0, 256}; }
  if (c == 86) return {CK_HEAD, W_TRANS0 + 3, 5, 0, 0, 256};
  if (c < 95) return {CK_GEN, W_RGB0 + 0, c - 87, 0, 0, 334};
  if (c < 98) return {CK_X40, W_RGB0 + 0, c - 95, 0, 256, 334};
  if (c < 114) { int l = 1 + (c - 98) / 8; return {CK_GEN, W_RGB0 + l, (c - 98) % 8, 0, 0, 256}; }
  return {CK_HEAD, W_RGB0 + 3, 3, 0, 0, 256};
}

// ---- "ray-bias" variant of the f16x3 forward stream (evaluation renders whose 128-sample tiles lie inside one ray: N % 128 == 0).
// The inputs of mlp_rgb.0 that are constant along a ray -- [ray_unit | PE(ray_unit)] (27 columns) and the light code (48) -- and the
// transient code of mlp_trans.0 (16 columns, constant per image) do not go through the matrix cores as 80 + 16 "extra" input columns
// of every sample: their products with the weights are added to the layer's bias ONCE per ray / image by a small pre-kernel
// (mlp_fwd_f16x3.hip, rb_*_kernel) and the tile seeds its accumulators with that per-ray bias.  What is left of the extras of R0 is
// x (3 columns), which the tile already staged for L0 / L4: k-step 3 of the encoding stage holds [PE slots 48..59 | x0 x1 x2 | 0], and
// the one remaining extra chunk (CK_RBX) multiplies it with zeros for the 12 encoding slots and mlp_rgb.0 columns 283..285 for x.
// Stream: the 115 chunks without the CK_X8 chunk (69), with the three CK_X40 chunks (95..97) replaced by one CK_RBX chunk and without
// the three CK_HEAD chunks (below) = 109 chunks, then the bias block, then (in the room of the chunks saved) the fp32 tables:
//   aux + 0      [27][256]  mlp_rgb.0 columns 256..282   (view)       aux = packed + kNumChunksRB * kChunkFloats + kBiasFloats
//   aux + 6912   [48][256]  mlp_rgb.0 columns 286..333   (light)
//   aux + 19200  [16][256]  mlp_trans.0 columns 256..271 (transient)
//   aux + 23296  [9][256]   the rows of the three narrow output layers in lane-read order (rb_head_src)
constexpr int kNumChunksRB = 109;
constexpr int kFirstHeadChunkRB = kFirstHeadChunk - 1;   // (the density head's chunk, 52, is not in this stream)
constexpr int64_t kRbAuxOff = (int64_t)kNumChunksRB * kChunkFloats + kBiasFloats;
constexpr int kRbAuxView = 0, kRbAuxLight = 27 * 256, kRbAuxTrans = (27 + 48) * 256, kRbAuxHeads = (27 + 48 + 16) * 256;
constexpr int kRbHeadRows = 9;                            // density | transient 0..4 | rgb 0..2
constexpr int kRbAuxFloats = kRbAuxHeads + kRbHeadRows * 256;
static_assert(kRbAuxOff + kRbAuxFloats <= kPackedFloats, "the ray-bias stream and its aux block fit the standard packed buffer");
// The three narrow output layers of this kernel run as fp32 dot products on the vector ALU from a table that stays in LDS
// (gen_wide_asm.py: gen_head_valu), so the stream has no head chunks either: 115 - CK_X8 - 3 CK_X40 + CK_RBX - 3 CK_HEAD = 109.  The
// aux block carries the nine rows UNSCALED in the order a lane reads them: float (k * 2 + h) * 4 + i of a row =
// W[row][feat_of(k >> 2, 4 (k & 3) + i, h)]  (k: 16-byte read 0..31 = accumulator registers 4 (k & 3) .. + 3 of source tile k >> 2).
TP_HD ChunkDesc chunk_desc_rb(int c) {
  if (c < 52) return chunk_desc(c);
  if (c < 68) return chunk_desc(c + 1);
  if (c < 84) return chunk_desc(c + 2);
  if (c < 92) return chunk_desc(c + 3);
  if (c == 92) return {CK_RBX, W_RGB0 + 0, 0, 0, 256, 334};
  return chunk_desc(c + 5);
}
// (weight matrix, row, col) of float `e` (0 .. 9 * 256 - 1) of the head table
TP_HD void rb_head_src(int e, int& mat, int& row, int& col) {
  const int r = e >> 8, f = e & 255, i = f & 3, h = (f >> 2) & 1, k = f >> 3;
  mat = r == 0 ? W_FEAT0 + 7 : (r < 6 ? W_TRANS0 + 3 : W_RGB0 + 3);
  row = r == 0 ? 0 : (r < 6 ? r - 1 : r - 6);
  col = feat_of(k >> 2, 4 * (k & 3) + i, h);
}

// Source element of packed float `idx` (0..8191) of chunk c: returns (row, col) of the weight matrix
// chunk_desc(c).mat, or row = -1 for a zero pad.
TP_HD void chunk_src(const ChunkDesc& d, int idx, int& row, int& col) {
  const int sub = idx & 3, lane = (idx >> 2) & 63, i = lane & 31, h = lane >> 5;
  row = -1; col = 0;
  if (d.kind == CK_HEAD) {
    const int s = (idx >> 8) * 4 + sub;            // k-step 0..127
    if (i < d.sub) { row = i; col = feat_of(s >> 4, s & 15, h); }
    return;
  }
  const int g = (idx >> 8) & 1, rr = idx >> 9;     // tile group, k-step within the chunk
  const int t = g * 4 + sub;
  int k;
  if (d.kind == CK_GEN) k = feat_of(d.sub, rr, h);
  else if (d.kind == CK_ENC) k = enc_col(d.sub * 16 + rr, h);
  else if (d.kind == CK_X8) k = x8_col(rr, h);
  else k = x40_col(d.sub * 16 + rr, h);
  if (k < 0) return;
  row = d.row_off + 32 * t + i;
  col = d.col_off + k;
}

// bias block: [wide layer][h][t][r] then 16 head scalars: b7[0], T3 bias[0..4], R3 bias[0..2]
TP_HD int bias_index(int wide, int h, int t, int r) { return ((wide * 2 + h) * 8 + t) * 16 + r; }
constexpr int kHeadBiasOff = kNumWide * 256;

}  // namespace tp_layout

// ================================================================================================
// Training record + backward layouts
// ================================================================================================
// The forward (train mode) records, per group of 32 consecutive samples (one wave's tile), seven
// post-ReLU activations and the per-sample non-feature inputs of the rgb head:
//   block b of group g at   saved + g * kSavedGroupFloats + b * 8192,   [256 features][32 samples]
//   slot 0: trunk feature   1..3: transient head h0,h1,h2   4..6: rgb head h0,h1,h2
//   slot 7 (32 rows only): rows 0..26 = [ray_unit, PE(ray_unit)], 27..29 = x   (mlp_rgb.0 columns 256..285)
//   then the ReLU sign bits of slots 1..6 (what the dgrad kernel needs of them)
// Inside a 128-byte feature row the eight 16-byte sample quads are XOR-swizzled with (f>>1)&7 so that the
// weight-gradient GEMM (samples = MFMA k) can ds_read_b128 its A/B fragments from a lane-linear LDS copy
// of the block without bank conflicts.  The backward writes its dz blocks in the same format.
namespace tp_layout {

constexpr int kBlockFloats = 8192;                        // [256][32]
constexpr int kSavedSlots = 7;
// after the 7 blocks + the narrow block: ReLU masks of slots 1..6 as per-lane bit words,
// word w (0..3) of slot sl for lane l at kMaskOff + (((sl-1)*4 + w)*64 + l); bit b <-> tile 2w + b/16, register b%16
constexpr int kMaskOff = kSavedSlots * kBlockFloats + 1024;
constexpr int kSavedGroupFloats = kMaskOff + 6 * 4 * 64;
enum { SV_FEAT = 0, SV_T0 = 1, SV_T1 = 2, SV_T2 = 3, SV_R0 = 4, SV_R1 = 5, SV_R2 = 6, SV_EX = 7 };

constexpr TP_HD int blk_off(int f, int j) { return f * 32 + ((((j >> 2) ^ ((f >> 1) & 7)) << 2) | (j & 3)); }

// dz record written by the dgrad kernel: per group 6 wide blocks + 2 narrow (32-row) blocks
//   0: dzT2  1: dzT1  2: dzT0  3: dzR2  4: dzR1  5: dzR0   then  dzT3 (5 rows), dzR3 (3 rows)
constexpr int kDzGroupFloats = 6 * kBlockFloats + 2 * 1024;
enum { DZ_T2 = 0, DZ_T1 = 1, DZ_T0 = 2, DZ_R2 = 3, DZ_R1 = 4, DZ_R0 = 5 };
constexpr int kDzT3Off = 6 * kBlockFloats, kDzR3Off = 6 * kBlockFloats + 1024;

// Transposed weight stream of the dgrad kernel (dh_in = W^T dz_out), same chunk format as the forward:
//   head T: [W3^T: 1 chunk, 3 k-steps] [W2^T: 8 chunks] [W1^T: 8 chunks]   then the same for head R
constexpr int kNumChunksT = 34;
constexpr int64_t kPackedTFloats = (int64_t)kNumChunksT * kChunkFloats;

// (matrix id, ts or -1 for the narrow chunk) of transposed chunk c
TP_HD void chunkT_desc(int c, int& mat, int& ts, int& rows) {
  const int head = c / 17, k = c % 17;
  const int base = head == 0 ? W_TRANS0 : W_RGB0;
  if (k == 0) { mat = base + 3; ts = -1; rows = head == 0 ? 5 : 3; return; }
  mat = base + (k <= 8 ? 2 : 1);
  ts = (k - 1) % 8;
  rows = 256;
}
// source element of packed float idx of transposed chunk c: value = W[mat][o][f]; o<0 => zero
TP_HD void chunkT_src(int c, int idx, int& mat, int& o, int& f) {
  int ts, rows;
  chunkT_desc(c, mat, ts, rows);
  const int sub = idx & 3, lane = (idx >> 2) & 63, i = lane & 31, h = lane >> 5;
  const int g = (idx >> 8) & 1, rr = idx >> 9, t = g * 4 + sub;
  f = 32 * t + i;                                   // output (input-feature) row of this fragment
  if (ts < 0) { o = 2 * rr + h; if (o >= rows) o = -1; }   // narrow: k-step rr contracts outputs 2rr, 2rr+1
  else o = feat_of(ts, rr, h);
}

}  // namespace tp_layout

// ================================================================================================
// f16x3 stream: every fp32 weight is carried as an unevaluated sum hi + lo of two fp16 numbers
// (22-bit significand) of W * 2^kF16WeightShift; the kernel forms hi*hi + hi*lo + lo*hi on the f16 matrix
// cores with fp32 accumulation (products of two fp16 are exact in fp32).  Same 115-chunk schedule as the fp32
// stream; a 32 KiB chunk holds 2 k-steps of 16 features:
//     half index = ((((s*8 + t)*2 + part)*64 + lane)*8 + j),  part 0 = hi, 1 = lo,
//     lane = (i = out row in tile t, h),  k (feature within the 16) = 8h + j for "extra" inputs, and
//     feature 32*ts + 16 s + 8 (j>>2) + 4 h + (j&3) when the B operand is a previous accumulator tile ts
// (the MFMA C/D register r = 8 s + j of lane half h is row (r&3) + 8 (r>>2) + 4 h).
// Head chunks hold 16 k-steps of one tile: half index = (((s16*2 + part)*64 + lane)*8 + j).
// ================================================================================================
namespace tp_layout {

constexpr int kF16WeightShift = 8;
constexpr int kChunkHalves = 16384;

// Transposed f16x3 stream of the split-fp16 dgrad kernel: the 34 chunks of chunkT_desc in the f16x3 chunk format
// (2 k-steps of 16 contracted OUTPUT features o, 8 tiles of 32 input-feature rows f).  value = W[mat][o][f] * 2^shift.
// The narrow chunks carry their 5 / 3 outputs in k-step 0, slots 8 h + j.
TP_HD void chunkT16_src(int c, int idx, int& part, int& mat, int& o, int& f) {
  int ts, rows;
  chunkT_desc(c, mat, ts, rows);
  const int j = idx & 7, lane = (idx >> 3) & 63, i = lane & 31, h = lane >> 5;
  part = (idx >> 9) & 1;
  const int t = (idx >> 10) & 7, s = idx >> 13;
  f = 32 * t + i;
  if (ts < 0) { o = s == 0 ? 8 * h + j : -1; if (o >= rows) o = -1; }
  else o = 32 * ts + 16 * s + 8 * (j >> 2) + 4 * h + (j & 3);          // acc_feat16(ts, s, h, j)
}

TP_HD int acc_feat16(int ts, int s, int h, int j) { return 32 * ts + 16 * s + 8 * (j >> 2) + 4 * h + (j & 3); }

// "extra" inputs in natural column order, 16 per k-step: slot = 16 s' + 8 h + j
TP_HD int enc_slot_col(int slot) { return slot < 60 ? 3 + slot : (slot < 63 ? slot - 60 : -1); }   // [PE(x) | x | pad]
TP_HD int x40_slot_col(int slot) { return slot < 78 ? slot : -1; }                                   // natural order
TP_HD int x8_slot_col(int slot) { return slot < 16 ? slot : -1; }

// (row, col) of the weight matrix for half `idx` (0..16383) of chunk c (part is decoded by the caller)
TP_HD void chunk16_src(const ChunkDesc& d, int idx, int& part, int& row, int& col) {
  const int j = idx & 7, lane = (idx >> 3) & 63, i = lane & 31, h = lane >> 5;
  part = (idx >> 9) & 1;
  row = -1; col = 0;
  if (d.kind == CK_HEAD) {
    const int s16 = idx >> 10;
    if (i < d.sub) { row = i; col = acc_feat16(s16 >> 1, s16 & 1, h, j); }
    return;
  }
  const int t = (idx >> 10) & 7, s = idx >> 13;
  int k;
  if (d.kind == CK_GEN) k = acc_feat16(d.sub, s, h, j);
  else if (d.kind == CK_ENC) k = enc_slot_col((d.sub * 2 + s) * 16 + 8 * h + j);
  else if (d.kind == CK_X8) k = s == 0 ? x8_slot_col(8 * h + j) : -1;
  else if (d.kind == CK_RBX) k = (s == 0 && h == 1 && j >= 4 && j <= 6) ? 27 + (j - 4) : -1;   // slots 60..62 of the encoding stage = x
  else k = x40_slot_col((d.sub * 2 + s) * 16 + 8 * h + j);
  if (k < 0) return;
  row = d.row_off + 32 * t + i;
  col = d.col_off + k;
}

}  // namespace tp_layout
