// K18: the feature-loss chain of the generator step as ONE C-ABI call.
//
// Reference: model/nerf_adapt_st_gan.py:758-766 -- loss.feat = P(rgb, image m + image_syn pad) + 5 P(rgb m + image (1 - m), image) --
// over layers/perceptual_loss.py:8-45: P(a, b) = mse(F(norm(a)), F(norm(b))), F = torchvision VGG19 features[:15] (conv 3-64, 64-64,
// pool, 64-128, 128-128, pool, 128-256, 256-256, 256-256; ReLU after every convolution but the last), frozen, targets detached.
// The generator step needs the VALUE of the term and its gradient wrt the rendered colours, nothing else.
//
// Round 4 ran this as 23 launches of general-purpose pieces under autograd (input stack, 7 convolutions, 2 pools, pair loss and the
// same again backwards): 193 us alone on the GPU, 287 us inside the iteration, on the critical path between the render and its
// backward.  Here the chain is written out as the 17 launches it needs, with everything that is not a convolution folded into a
// convolution's epilogue, and every convolution's operand loads issued up front:
//   * forward convolution: bias + ReLU in the epilogue; the two layers in front of a pool write the POOLED map and the arg-max byte
//     (the full-resolution map is never stored: nothing reads it);
//   * backward (data gradient, transposed convolution): the ReLU derivative of the layer BELOW is applied to the result in the
//     epilogue (`act > 0`), or, where the layer below is a pool, the result is routed to the arg-max position of its 2x2 window
//     (un-pool + ReLU derivative in one epilogue) -- so every backward kernel gathers a PLAIN cotangent (round 4 gathered cotangent
//     and mask: 3 loads per product instead of 2);
//   * only the first 2B images (the two `fake` stacks) are differentiated: the targets carry no gradient;
//   * the pair loss and its cotangent are one launch;
//   * k loop: each wavefront loads the operands of up to eight channel pairs (144 values per lane) before its first MFMA -- the
//     weights come from the Infinity Cache every iteration (the MLP kernels sweep the L2 in between), and round 4's two pairs per
//     trip paid that latency four times per layer.
// Arithmetic: exact fp32 products on v_mfma_f32_32x32x2_f32, fp32 accumulation, fixed summation order (run-to-run deterministic);
// the same split of a tile's k range over 4 wavefronts x S workgroups as K12, whose values this chain reproduces up to the order of
// the k sums.
#include "conv_mma.h"

namespace {

enum { EPI_PLAIN = 0, EPI_POOL = 1, EPI_MASK = 2, EPI_UNPOOL = 3 };

struct FcP {
  const float* in;            // F: x [N,C,H,W];  T: gy [N,Co,H,W]
  const float* w;             // packed: F [C][9][Co], T [Co][9 flipped][C]
  const float* bias;          // F: [Co]
  float* out;                 // F plain: [N,Co,H,W]; F pool: [N,Co,H/2,W/2]; T plain / mask: [N,C,H,W]; T unpool: [N,C,2H,2W]
  const float* act;           // T mask: the layer's input activation [.,C,H,W]; T unpool: the pooled activation [.,C,H,W]
  const unsigned char* arg;   // T unpool: arg-max bytes [.,C,H,W]
  unsigned char* arg_out;     // F pool: [N,Co,H/2,W/2]
  float* ws;
  unsigned* cnt;
  int N, C, H, W, Co;
  int lw, lp;                 // log2(W), log2(H * W)
  int S, tiles_n, relu;
};

// T = false: y = relu?(conv(x, w) + bias) [+ 2x2 max pool].  T = true: gx = conv^T(gy, w) [masked by act > 0 | un-pooled].
template <bool T, int EPI, int NP>
__global__ __launch_bounds__(256) void fc_conv_kernel(FcP p) {
  __shared__ float lds[kReduceLdsFloats];                  // (reduce_tiles; the pool epilogue's [32][33] totals fit)
  const int t = threadIdx.x, w = t >> 6, lane = t & 63, col = lane & 31, h = lane >> 5;
  const int s = blockIdx.x % p.S, tile = blockIdx.x / p.S, nt = tile % p.tiles_n, mt = tile / p.tiles_n;
  const int P = 1 << p.lp, M = p.N << p.lp;
  const int CK = T ? p.Co : p.C, CN = T ? p.C : p.Co;          // contracted / produced channels
  const int m = mt * 32 + col, mc = min(m, M - 1);
  const int n = mc >> p.lp, pp = mc & (P - 1), y0 = pp >> p.lw, x0 = pp & (p.W - 1);
  int off[9];
  bool ok[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const int yy = y0 + k / 3 - 1, xx = x0 + k % 3 - 1;
    ok[k] = m < M && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
    off[k] = ok[k] ? yy * p.W + xx : 0;
  }
  const float* ia = p.in + (size_t)n * CK * P;
  const int cn = min(nt * 32 + col, CN - 1);
  // weight element of (produced channel cn, contracted channel ck, neighbourhood index k): PACKED [ck][k][cn] (tp_feat_chain_pack; the
  // backward image has its taps flipped), so that the 32 lanes of a half-wavefront read 128 consecutive bytes -- with the tensor's own
  // [Co][C][3][3] layout every lane read its own row, 32 cache lines per load instruction, and the vector memory pipeline's line rate,
  // not latency, set the kernel's time (10-13 us per layer, profiles/r5)
  const float* wp = p.w + cn;
  const size_t wstep = (size_t)9 * CN;                          // per contracted channel
  int qb, qe;
  k_range((CK + 1) >> 1, p.S, s, w, qb, qe);
  f32x16 acc[1] = {};
  for (int q0 = qb; q0 < qe; q0 += NP) {
    float a[NP][9], b[NP][9];
    float live[NP];
#pragma unroll
    for (int u = 0; u < NP; ++u) {
      const int ck = 2 * min(q0 + u, qe - 1) + h, cc = min(ck, CK - 1);
      live[u] = (q0 + u < qe && ck < CK) ? 1.f : 0.f;
      const float* ic = ia + (size_t)cc * P;
      const float* wc = wp + wstep * cc;
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        a[u][k] = ic[off[k]];
        b[u][k] = wc[(size_t)k * CN];
      }
    }
#pragma unroll
    for (int u = 0; u < NP; ++u)
#pragma unroll
      for (int k = 0; k < 9; ++k) acc[0] = mfma(ok[k] ? a[u][k] : 0.f, b[u][k] * live[u], acc[0]);
  }
  float out[1][4];
  ConvP rp;
  rp.ws = p.ws; rp.cnt = p.cnt; rp.S = p.S;
  if (!reduce_tiles<1>(acc, out, lds, rp, tile, s)) return;
  // thread (w, lane): rows 8w + 4h + 0..3 of the tile (four consecutive positions of one image row), column `col`
  const int r0 = 8 * w + 4 * h, m0 = mt * 32 + r0, oc = nt * 32 + col;
  const bool inside = m0 < M && oc < CN;
  f32x4 v = {out[0][0], out[0][1], out[0][2], out[0][3]};
  const int n0 = min(m0, M - 4) >> p.lp, p0 = min(m0, M - 4) & (P - 1);
  if (!T) {
    if (p.bias) v += p.bias[min(oc, CN - 1)];
    if (p.relu)
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.f);
  }
  if (EPI == EPI_PLAIN) {
    if (inside) *reinterpret_cast<f32x4*>(p.out + ((size_t)n0 * CN + oc) * P + p0) = v;
  } else if (EPI == EPI_MASK) {
    if (inside) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(p.act + ((size_t)n0 * CN + oc) * P + p0);
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = a[i] > 0.f ? v[i] : 0.f;
      *reinterpret_cast<f32x4*>(p.out + ((size_t)n0 * CN + oc) * P + p0) = v;
    }
  } else if (EPI == EPI_UNPOOL) {
    // positions (y, x .. x+3) of the pooled map -> rows 2y, 2y+1, columns 2x .. 2x+7 of the full map: the cotangent goes to the window's
    // arg-max where the pooled activation is positive (ReLU sits in front of the pool), zeros elsewhere
    if (inside) {
      const size_t e = ((size_t)n0 * CN + oc) * P + p0;
      const f32x4 a = *reinterpret_cast<const f32x4*>(p.act + e);
      const unsigned arg4 = *reinterpret_cast<const unsigned*>(p.arg + e);
      const int y = p0 >> p.lw, x = p0 & (p.W - 1), W2 = 2 * p.W;
      float* row0 = p.out + (((size_t)n0 * CN + oc) * (2 * p.H) + 2 * y) * W2 + 2 * x;
      float r[2][8];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const unsigned id = (arg4 >> (8 * i)) & 3u;
        const float g = a[i] > 0.f ? v[i] : 0.f;
        r[0][2 * i] = id == 0 ? g : 0.f; r[0][2 * i + 1] = id == 1 ? g : 0.f;
        r[1][2 * i] = id == 2 ? g : 0.f; r[1][2 * i + 1] = id == 3 ? g : 0.f;
      }
#pragma unroll
      for (int dy = 0; dy < 2; ++dy) {
        *reinterpret_cast<f32x4*>(row0 + dy * W2) = f32x4{r[dy][0], r[dy][1], r[dy][2], r[dy][3]};
        *reinterpret_cast<f32x4*>(row0 + dy * W2 + 4) = f32x4{r[dy][4], r[dy][5], r[dy][6], r[dy][7]};
      }
    }
  } else {   // EPI_POOL: the tile's 32 positions are 32 / W whole rows of one image, an even number: whole 2x2 windows
    __syncthreads();                                            // (reduce_tiles' LDS reads are done)
#pragma unroll
    for (int i = 0; i < 4; ++i) lds[(r0 + i) * 33 + col] = v[i];
    __syncthreads();
    const int j = t >> 5, c = t & 31, hw = p.W >> 1;            // pooled element j (0..7) of channel c
    const int py = j / hw, px = j - py * hw;
    const int q00 = (2 * py) * p.W + 2 * px;
    const float c0 = lds[q00 * 33 + c], c1 = lds[(q00 + 1) * 33 + c], c2 = lds[(q00 + p.W) * 33 + c], c3 = lds[(q00 + p.W + 1) * 33 + c];
    float best = c0;
    unsigned id = 0;
    if (c1 > best) { best = c1; id = 1; }
    if (c2 > best) { best = c2; id = 2; }
    if (c3 > best) { best = c3; id = 3; }
    const int mb = mt * 32, occ = nt * 32 + c;
    if (mb < M && occ < CN) {
      const int nb = mb >> p.lp, pb = mb & (P - 1);
      const size_t e = ((size_t)nb * CN + occ) * (P >> 2) + (size_t)((pb >> p.lw) >> 1) * hw + (size_t)py * hw + px;
      p.out[e] = best;
      p.arg_out[e] = (unsigned char)id;
    }
  }
}

// ---- pair loss + its cotangent in one launch.  feat [4n] = features of [fake1 | fake2 | real1 | real2] (n elements each);
// out = {l1 + w2 l2, l1, l2} with l_k = mean((fake_k - real_k)^2);  g [2n] = scale * d out[0] / d (fake1 | fake2).
constexpr int kLossBlock = 256;
__global__ __launch_bounds__(kLossBlock) void fc_pair_loss_kernel(const float* __restrict__ feat, int n, float w2, float scale, float* __restrict__ g,
                                                                 float* __restrict__ part, unsigned* __restrict__ ticket, float* __restrict__ out) {
  __shared__ float red[2][kLossBlock];
  __shared__ int last;
  const int t = threadIdx.x, nb = gridDim.x;
  const float s1 = 2.f * scale / (float)n, s2 = s1 * w2;
  float a1 = 0.f, a2 = 0.f;
  for (int i = (blockIdx.x * kLossBlock + t) * 4; i < n; i += nb * kLossBlock * 4) {
    const f32x4 f1 = *reinterpret_cast<const f32x4*>(feat + i), f2 = *reinterpret_cast<const f32x4*>(feat + n + i);
    const f32x4 r1 = *reinterpret_cast<const f32x4*>(feat + 2 * (size_t)n + i), r2 = *reinterpret_cast<const f32x4*>(feat + 3 * (size_t)n + i);
    const f32x4 d1 = f1 - r1, d2 = f2 - r2;
#pragma unroll
    for (int k = 0; k < 4; ++k) { a1 += d1[k] * d1[k]; a2 += d2[k] * d2[k]; }
    *reinterpret_cast<f32x4*>(g + i) = d1 * s1;
    *reinterpret_cast<f32x4*>(g + n + i) = d2 * s2;
  }
  red[0][t] = a1; red[1][t] = a2;
  __syncthreads();
  for (int s = kLossBlock >> 1; s > 0; s >>= 1) {
    if (t < s) { red[0][t] += red[0][t + s]; red[1][t] += red[1][t + s]; }
    __syncthreads();
  }
  // per-block partial sums meet in the last block to arrive (the hand-over of conv_mma.h's reduce_tiles: device-scope stores,
  // drained, then one device-scope counter), which adds them in block order
  if (t == 0) {
    __hip_atomic_store(part + 2 * blockIdx.x, red[0][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(part + 2 * blockIdx.x + 1, red[1][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    last = (__hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(nb - 1));
    if (last) {
      float t1 = 0.f, t2 = 0.f;
      for (int b = 0; b < nb; ++b) {
        t1 += __hip_atomic_load(part + 2 * b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        t2 += __hip_atomic_load(part + 2 * b + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      const float l1 = t1 / (float)n, l2 = t2 / (float)n;
      out[0] = l1 + w2 * l2; out[1] = l1; out[2] = l2;
      __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// one launch packs all seven weights both ways: fwd[l][(ci * 9 + k) * Co + co] = W[co][ci][k], bwd[l][(co * 9 + k) * C + ci] = W[co][ci][8 - k]
struct PackP { const float* w[TP_FEAT_CHAIN_LAYERS]; long long end[TP_FEAT_CHAIN_LAYERS]; int C[TP_FEAT_CHAIN_LAYERS], Co[TP_FEAT_CHAIN_LAYERS]; float* fwd; float* bwd; };
__global__ __launch_bounds__(256) void fc_pack_kernel(PackP q) {
  const long long total = q.end[TP_FEAT_CHAIN_LAYERS - 1];
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    int l = 0;
    while (e >= q.end[l]) ++l;
    const long long base = l ? q.end[l - 1] : 0, i = e - base;          // i = (co * C + ci) * 9 + k in the tensor's own layout
    const int C = q.C[l], Co = q.Co[l];
    const int k = (int)(i % 9), ci = (int)((i / 9) % C), co = (int)(i / (9LL * C));
    const float v = q.w[l][i];
    q.fwd[base + ((long long)ci * 9 + k) * Co + co] = v;
    q.bwd[base + ((long long)co * 9 + (8 - k)) * C + ci] = v;
  }
}

constexpr int kLayers = TP_FEAT_CHAIN_LAYERS;
const int kCin[kLayers] = {3, 64, 64, 128, 128, 256, 256};
const int kCout[kLayers] = {64, 64, 128, 128, 256, 256, 256};
const int kPoolAfter[kLayers] = {0, 1, 0, 1, 0, 0, 0};           // a 2x2 max pool follows layers 1 and 3

struct Layout {          // float offsets into the workspace
  int64_t x0, act[kLayers], g[kLayers + 1], arg[2], ws, part, total;
  int64_t ws_floats, n_counters;
};

// plan of one launch: rows = N * H * W positions, `cols` produced channels, k units = contracted channel pairs
Plan conv_plan(int N, int H, int W, int cols, int contracted) { return plan(N * H * W, cols, (contracted + 1) / 2, 1, 256, 2); }

int layout(int B, int H, int W, Layout* L) {
  if (B <= 0 || H != 16 || W != 16) return -1;          // (pool epilogue: W <= 16; un-pool epilogue: pooled W >= 4)
  const int64_t n4 = 4 * (int64_t)B, nf = 2 * (int64_t)B;
  int64_t o = 0;
  auto take = [&](int64_t n) { const int64_t at = o; o += (n + 3) & ~(int64_t)3; return at; };
  L->x0 = take(n4 * 3 * H * W);
  int h = H;
  int64_t ws = 0, cnt = 0;
  for (int l = 0; l < kLayers; ++l) {
    const Plan f = conv_plan((int)n4, h, h, kCout[l], kCin[l]);
    const Plan b = conv_plan((int)nf, h, h, kCin[l], kCout[l]);
    ws = ws > (int64_t)f.ws_floats ? ws : (int64_t)f.ws_floats;
    ws = ws > (int64_t)b.ws_floats ? ws : (int64_t)b.ws_floats;
    cnt = cnt > (int64_t)f.tiles_m * f.tiles_n ? cnt : (int64_t)f.tiles_m * f.tiles_n;
    cnt = cnt > (int64_t)b.tiles_m * b.tiles_n ? cnt : (int64_t)b.tiles_m * b.tiles_n;
    // cotangent handed DOWN by layer l's backward: of its input -- or, where that input is a pooled map, of the full-resolution map
    // in front of the pool (the un-pool epilogue writes that directly)
    L->g[l] = take(nf * kCin[l] * h * h * (l > 0 && kPoolAfter[l - 1] ? 4 : 1));
    if (kPoolAfter[l]) h /= 2;
    L->act[l] = take(n4 * kCout[l] * h * h);                    // layer l's output (pooled where a pool follows)
  }
  L->g[kLayers] = take(nf * kCout[kLayers - 1] * h * h);        // cotangent of the features (from the loss)
  L->arg[0] = take((n4 * 64 * (H / 2) * (H / 2) + 3) / 4);
  L->arg[1] = take((n4 * 128 * (H / 4) * (H / 4) + 3) / 4);
  L->part = take(2 * 64);
  L->ws = take(ws);
  L->ws_floats = ws;
  L->n_counters = cnt + 1;                                      // + the loss kernel's ticket
  L->total = o;
  return 0;
}

template <bool T, int EPI>
void launch_conv(const FcP& p, const Plan& q, int pairs_per_wave, hipStream_t st) {
  const dim3 grid((unsigned)(q.tiles_m * q.tiles_n * q.S)), block(256);
  if (pairs_per_wave > 4) hipLaunchKernelGGL((fc_conv_kernel<T, EPI, 8>), grid, block, 0, st, p);
  else hipLaunchKernelGGL((fc_conv_kernel<T, EPI, 4>), grid, block, 0, st, p);
}
}  // namespace

extern "C" {

int64_t tp_feat_chain_workspace(int32_t B, int32_t H, int32_t W, int64_t* n_counters) {
  Layout L;
  if (layout(B, H, W, &L) != 0) return -1;
  if (n_counters) *n_counters = L.n_counters;
  return L.total;
}

int64_t tp_feat_chain_packed_floats(void) {
  int64_t total = 0;
  for (int l = 0; l < kLayers; ++l) total += (int64_t)kCout[l] * kCin[l] * 9;
  return 2 * total;
}

int tp_feat_chain_pack(const float* const* w, float* packed, tp_stream_t stream) {
  TP_REQUIRE(w && packed, "null argument");
  PackP q{};
  long long o = 0;
  for (int l = 0; l < kLayers; ++l) {
    TP_REQUIRE(w[l] != nullptr, "null weight");
    q.w[l] = w[l]; q.C[l] = kCin[l]; q.Co[l] = kCout[l];
    o += (long long)kCout[l] * kCin[l] * 9;
    q.end[l] = o;
  }
  q.fwd = packed; q.bwd = packed + o;
  hipLaunchKernelGGL(fc_pack_kernel, dim3(512), dim3(256), 0, (hipStream_t)stream, q);
  return tp::check_launch("tp_feat_chain_pack");
}

int tp_feat_chain(const tp_feat_chain_args* a, tp_stream_t stream) {
  TP_REQUIRE(a && a->rgb && a->gathered && a->loss && a->g_rgb && a->workspace && a->counters, "null argument");
  Layout L;
  TP_REQUIRE(layout(a->B, a->H, a->W, &L) == 0, "covers 16x16 patches");
  TP_REQUIRE(a->workspace_floats >= L.total && a->n_counters >= L.n_counters, "workspace / counters too small (tp_feat_chain_workspace)");
  for (int l = 0; l < kLayers; ++l) TP_REQUIRE(a->bias[l], "null bias");
  TP_REQUIRE(a->packed != nullptr, "packed weights missing (tp_feat_chain_pack)");
  const float* wf[kLayers];
  const float* wb[kLayers];
  {
    int64_t o = 0, total = 0;
    for (int l = 0; l < kLayers; ++l) total += (int64_t)kCout[l] * kCin[l] * 9;
    for (int l = 0; l < kLayers; ++l) { wf[l] = a->packed + o; wb[l] = a->packed + total + o; o += (int64_t)kCout[l] * kCin[l] * 9; }
  }
  hipStream_t st = (hipStream_t)stream;
  float* W0 = a->workspace;
  const int B = a->B, n4 = 4 * B, nf = 2 * B;
  // 1. the four image stacks, masked, concatenated, normalised
  tp_feat_inputs_args fi{};
  fi.rgb = a->rgb; fi.gathered = a->gathered; fi.B = B; fi.P = a->H * a->W; fi.n_channels = a->n_channels;
  fi.c_image = a->c_image; fi.c_image_syn = a->c_image_syn; fi.c_mask = a->c_mask; fi.c_mask_syn = a->c_mask_syn;
  for (int c = 0; c < 3; ++c) { fi.mean[c] = a->mean[c]; fi.std[c] = a->std[c]; }
  if (const int rc = tp_feat_inputs_fwd(&fi, W0 + L.x0, stream)) return rc;
  // 2. forward
  int h = a->H, n_pool = 0;
  const float* x = W0 + L.x0;
  int res[kLayers];
  for (int l = 0; l < kLayers; ++l) {
    res[l] = h;
    FcP p{};
    p.in = x; p.w = wf[l]; p.bias = a->bias[l]; p.out = W0 + L.act[l]; p.ws = W0 + L.ws; p.cnt = (unsigned*)a->counters;
    p.N = n4; p.C = kCin[l]; p.H = h; p.W = h; p.Co = kCout[l]; p.lw = ilog2(h); p.lp = 2 * p.lw; p.relu = l + 1 < kLayers;
    const Plan q = conv_plan(n4, h, h, kCout[l], kCin[l]);
    p.S = q.S; p.tiles_n = q.tiles_n;
    const int per_wave = ((((kCin[l] + 1) / 2) + q.S - 1) / q.S + 3) / 4;
    if (kPoolAfter[l]) {
      p.arg_out = (unsigned char*)(W0 + L.arg[n_pool++]);
      launch_conv<false, EPI_POOL>(p, q, per_wave, st);
      h /= 2;
    } else {
      launch_conv<false, EPI_PLAIN>(p, q, per_wave, st);
    }
    x = W0 + L.act[l];
  }
  // 3. loss + cotangent of the features of the two fake stacks
  {
    const int n = B * kCout[kLayers - 1] * h * h;
    int blocks = (n + kLossBlock * 4 - 1) / (kLossBlock * 4);
    blocks = blocks < 1 ? 1 : blocks > 64 ? 64 : blocks;
    hipLaunchKernelGGL(fc_pair_loss_kernel, dim3(blocks), dim3(kLossBlock), 0, st, W0 + L.act[kLayers - 1], n, a->w2, a->scale,
                       W0 + L.g[kLayers], W0 + L.part, (unsigned*)a->counters + (L.n_counters - 1), a->loss);
  }
  // 4. backward through the first 2B images: data gradients only (frozen network)
  for (int l = kLayers - 1; l >= 0; --l) {
    const int hl = res[l];
    FcP p{};
    p.in = W0 + L.g[l + 1]; p.w = wb[l]; p.out = W0 + L.g[l]; p.ws = W0 + L.ws; p.cnt = (unsigned*)a->counters;
    p.N = nf; p.C = kCin[l]; p.H = hl; p.W = hl; p.Co = kCout[l]; p.lw = ilog2(hl); p.lp = 2 * p.lw;
    const Plan q = conv_plan(nf, hl, hl, kCin[l], kCout[l]);
    p.S = q.S; p.tiles_n = q.tiles_n;
    const int per_wave = ((((kCout[l] + 1) / 2) + q.S - 1) / q.S + 3) / 4;
    if (l == 0) {
      launch_conv<true, EPI_PLAIN>(p, q, per_wave, st);          // the network input: no activation below
    } else if (kPoolAfter[l - 1]) {
      // this layer's input is pool(relu(conv_{l-1})): un-pool into the cotangent of that layer's full-resolution output.  g[l] then
      // holds [nf, C, 2h, 2h] -- which is what layout() reserved (layer l-1's resolution)
      p.act = W0 + L.act[l - 1];
      p.arg = (const unsigned char*)(W0 + L.arg[l == 2 ? 0 : 1]);
      launch_conv<true, EPI_UNPOOL>(p, q, per_wave, st);
    } else {
      p.act = W0 + L.act[l - 1];
      launch_conv<true, EPI_MASK>(p, q, per_wave, st);
    }
  }
  if (const int rc = tp::check_launch("tp_feat_chain")) return rc;
  // 5. d / d rgb of the two fake stacks
  return tp_feat_inputs_bwd(&fi, W0 + L.g[0], a->g_rgb, stream);
}
}
