// K11: the stride-2 4x4 convolutions of the PatchGAN ladder as implicit GEMMs on the fp32 matrix cores.
// reference layers/discriminator.py:94-115 (spectral_norm(Conv2d(c_in, c_out, (4,4), (2,2), (1,1), bias=False)) per ladder
// stage) and model/nerf_adapt_st_gan.py:129-171, 794-807 (disc_trainstep, compute_grad2: the R1 penalty differentiates the
// data gradient once more).  A convolution is bilinear in (x, W), so three kernels are closed under differentiation:
//     F  y  = conv(x, W)            y [n,co,oy,ox] = sum_{ci,ky,kx} W[co,ci,ky,kx] x[n,ci,2oy-1+ky,2ox-1+kx]
//     D  gx = conv^T(gy, W)         gx[n,ci,iy,ix] = sum_{co,ky,kx} W[co,ci,ky,kx] gy[n,co,(iy+1-ky)/2,(ix+1-kx)/2]
//     G  gW = corr(gy, x)           gW[co,ci,ky,kx] = sum_{n,oy,ox} gy[n,co,oy,ox] x[n,ci,2oy-1+ky,2ox-1+kx]
// (d F = F(dx, W) + F(x, dW); the backward of D wrt (gy, W) is (F, G); of G wrt (gy, x) is (F, D)): texpose_amd/autograd_ops.py
// composes them to any order.  MIOpen runs each of these as three layout transposes + an NHWC igemm (or a naive fallback in
// the double backward): 90 launches and 0.6 ms of a 3.5 ms B=4 iteration; here each is ONE launch.
//
// All three are v_mfma_f32_32x32x2f32 GEMMs (exact fp32 products, fp32 accumulation) with the im2col gather folded into the
// operand loads -- every lane loads its own A and B element straight from the NCHW tensors (L2-resident: the largest operand,
// the 256->512 weight, is 8 MB) in an order that makes the loads 16-byte or contiguous across lanes:
//   F: rows = output positions, columns = co, k = (ci, tap): per ci the lower lane half contracts taps ky in {0,1}, the upper
//      half ky in {2,3}: 8 gathered x values + W[co,ci,2h..2h+1,:] (two 16-byte loads) feed 8 MFMAs.
//   D: rows = positions (a,b) of ONE parity class of input pixels (iy,ix) = (2a+py, 2b+px), four classes = four accumulator
//      tiles sharing the loads; k = (co, 2x2 taps of the class); the lane halves take even / odd co: the 3x3 neighbourhood of
//      gy (9 loads) + the 16 taps of W[co,ci] (four 16-byte loads) feed 16 MFMAs.
//   G: rows = co, columns = (ci, tap), k = positions: gy as 16-byte loads along a row of the map, x gathered.
// One workgroup = 4 wavefronts on ONE output tile, each with a quarter of the k range; tiles with a long k are further split
// over S workgroups whose partial sums meet in a workspace, the last one to arrive (device-scope counter) adds them in
// slice order: fixed summation order, run-to-run deterministic, no float atomics.
#include "conv_mma.h"

namespace {
// ---------------------------------------------------------------------------------------------------------------- F
__global__ __launch_bounds__(256) void conv4s2_fwd_kernel(ConvP p) {
  __shared__ float lds[kReduceLdsFloats];
  const int t = threadIdx.x, w = t >> 6, lane = t & 63, col = lane & 31, h = lane >> 5;
  const int s = blockIdx.x % p.S, tile = blockIdx.x / p.S, nt = tile % p.tiles_n, mt = tile / p.tiles_n;
  const int P = 1 << p.lp, M = p.N << p.lp, HW = p.H * p.W;
  // A operand: position m = (n, oy, ox); this lane half gathers rows ky = 2h, 2h+1 of the 4x4 window
  const int m = mt * 32 + col, mc = min(m, M - 1);
  const int n = mc >> p.lp, pp = mc & (P - 1), oy = pp >> p.low, ox = pp & (p.OW - 1);
  int off[8];
  bool ok[8];
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int iy = 2 * oy - 1 + 2 * h + r, ix = 2 * ox - 1 + j;
      ok[r * 4 + j] = m < M && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
      off[r * 4 + j] = ok[r * 4 + j] ? iy * p.W + ix : 0;
    }
  const float* xa = p.x + (size_t)n * p.C * HW;
  // B operand: W[co, ci, 2h..2h+1, 0..3]
  const int co = min(nt * 32 + col, p.Co - 1);
  const float* wb = p.w + (size_t)co * p.C * 16 + 8 * h;
  int cb, ce;
  k_range(p.C, p.S, s, w, cb, ce);
  f32x16 acc[1] = {};
  constexpr int kU = 8;                              // channels whose operands are in flight together (one load latency per batch)
  for (int c0 = cb; c0 < ce; c0 += kU) {
    float a[kU][8];
    f32x4 b[kU][2];
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int ci = min(c0 + u, ce - 1);
      const float* xc = xa + (size_t)ci * HW;
#pragma unroll
      for (int q = 0; q < 8; ++q) a[u][q] = xc[off[q]];
      b[u][0] = *reinterpret_cast<const f32x4*>(wb + (size_t)ci * 16);
      b[u][1] = *reinterpret_cast<const f32x4*>(wb + (size_t)ci * 16 + 4);
    }
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      if (c0 + u < ce) {                              // (wave-uniform; the loads above stay clamped and unconditional)
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[0] = mfma(ok[q] ? a[u][q] : 0.f, b[u][q >> 2][q & 3], acc[0]);
      }
    }
  }
  float out[1][4];
  if (!reduce_tiles<1>(acc, out, lds, p, tile, s)) return;
  // thread: rows m0 .. m0+3 (four consecutive positions of one image), column co
  const int m0 = mt * 32 + 8 * w + 4 * h, oc = nt * 32 + col;
  if (m0 < M && oc < p.Co) {
    const int n0 = m0 >> p.lp, p0 = m0 & (P - 1);
    *reinterpret_cast<f32x4*>(p.out + ((size_t)n0 * p.Co + oc) * P + p0) = f32x4{out[0][0], out[0][1], out[0][2], out[0][3]};
  }
}

// ---------------------------------------------------------------------------------------------------------------- F + IN
// The forward convolution with the InstanceNorm2d + LeakyReLU that follows it in the PatchGAN ladder (reference
// layers/discriminator.py:94-115; K9 csrc/inorm_lrelu.hip) in its epilogue: a workgroup owns NT row tiles = 32 NT consecutive
// positions of 32 output channels, i.e. WHOLE (image, channel) instances when the map has P = 16 (NT = 1: two images) or P = 64
// (NT = 2: one image) positions.  The workgroup that ends up with the split-K totals normalises them: per instance 256 / n_inst
// adjacent lanes hold P / parts rows each; mean and the sum of squared deviations by xor-shuffles inside the lane group (fixed order).
// Outputs: y = lrelu(xhat), xhat, rstd -- what tp_inorm_lrelu_fwd returns -- and no z.  One launch instead of two per ladder stage.
struct InP { float* xhat; float* rstd; float eps, slope; };

// (pa, qa) and (pb, qb): TWO independent problems in one launch -- workgroups [0, na) work on the first, the rest on the second (na < 0: one
// problem).  The discriminator step runs its real and its fake pass as such pairs (texpose_amd/disc_step.py): the kernels are latency-sized
// at these shapes, so two side by side take the time of one.
template <int NT>
__global__ __launch_bounds__(256) void conv4s2_fwd_in_kernel(ConvP pa, InP qa, ConvP pb, InP qb, int na) {
  __shared__ float lds[NT * 32 * 33 > kReduceLdsFloats ? NT * 32 * 33 : kReduceLdsFloats];    // (reduce_tiles, then the [rows][33] totals)
  const bool second = na >= 0 && (int)blockIdx.x >= na;
  const ConvP& p = second ? pb : pa;
  const InP& q = second ? qb : qa;
  const int bid = (int)blockIdx.x - (second ? na : 0);
  const int t = threadIdx.x, w = t >> 6, lane = t & 63, col = lane & 31, h = lane >> 5;
  const int s = bid % p.S, tile = bid / p.S, nt = tile % p.tiles_n, mt = tile / p.tiles_n;
  const int P = 1 << p.lp, M = p.N << p.lp, HW = p.H * p.W;
  if (p.x_copy != nullptr) {                         // (side job: this problem's input, 16 bytes per thread and step; host: a multiple of 4 floats)
    const int n4 = (p.N * p.C * HW) >> 2, wgs = (second ? (int)gridDim.x - na : (na >= 0 ? na : (int)gridDim.x));
    for (int i = bid * 256 + t; i < n4; i += wgs * 256) reinterpret_cast<f32x4*>(p.x_copy)[i] = reinterpret_cast<const f32x4*>(p.x)[i];
  }
  int off[NT][8];
  bool ok[NT][8];
  const float* xa[NT];
#pragma unroll
  for (int r_ = 0; r_ < NT; ++r_) {
    const int m = (mt * NT + r_) * 32 + col, mc = min(m, M - 1);
    const int n = mc >> p.lp, pp = mc & (P - 1), oy = pp >> p.low, ox = pp & (p.OW - 1);
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int iy = 2 * oy - 1 + 2 * h + r, ix = 2 * ox - 1 + j;
        ok[r_][r * 4 + j] = m < M && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        off[r_][r * 4 + j] = ok[r_][r * 4 + j] ? iy * p.W + ix : 0;
      }
    xa[r_] = p.x + (size_t)n * p.C * HW;
  }
  const int co = min(nt * 32 + col, p.Co - 1);
  const float* wb = p.w + (size_t)co * p.C * 16 + 8 * h;
  int cb, ce;
  k_range(p.C, p.S, s, w, cb, ce);
  f32x16 acc[NT] = {};
  constexpr int kU = NT > 1 ? 4 : 8;
  for (int c0 = cb; c0 < ce; c0 += kU) {
    float a[NT][kU][8];
    f32x4 b[kU][2];
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int ci = min(c0 + u, ce - 1);
#pragma unroll
      for (int r_ = 0; r_ < NT; ++r_) {
        const float* xc = xa[r_] + (size_t)ci * HW;
#pragma unroll
        for (int qq = 0; qq < 8; ++qq) a[r_][u][qq] = xc[off[r_][qq]];
      }
      b[u][0] = *reinterpret_cast<const f32x4*>(wb + (size_t)ci * 16);
      b[u][1] = *reinterpret_cast<const f32x4*>(wb + (size_t)ci * 16 + 4);
    }
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      if (c0 + u < ce) {
#pragma unroll
        for (int r_ = 0; r_ < NT; ++r_)
#pragma unroll
          for (int qq = 0; qq < 8; ++qq) acc[r_] = mfma(ok[r_][qq] ? a[r_][u][qq] : 0.f, b[u][qq >> 2][qq & 3], acc[r_]);
      }
    }
  }
  float out[NT][4];
  if (!reduce_tiles<NT>(acc, out, lds, p, tile, s)) return;
  // ---- the totals of 32 NT rows x 32 columns through LDS: [row][33]
  __syncthreads();
  float* tl = lds;
#pragma unroll
  for (int r_ = 0; r_ < NT; ++r_)
#pragma unroll
    for (int i = 0; i < 4; ++i) tl[(r_ * 32 + 8 * w + 4 * h + i) * 33 + col] = out[r_][i];
  __syncthreads();
  const int rows = 32 * NT, imgs = rows >> p.lp, n_inst = imgs * 32, parts = 256 / n_inst, rp = P / parts;     // 64 / 4 / 4  or  32 / 8 / 8
  const int inst = t / parts, part = t - inst * parts, im = inst >> 5, c = inst & 31;
  const int r0 = im * P + part * rp;
  float v[8];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    v[i] = i < rp ? tl[(r0 + i) * 33 + c] : 0.f;
    sum += v[i];
  }
  for (int o = 1; o < parts; o <<= 1) sum += __shfl_xor(sum, o, 64);
  const float mean = sum / (float)P;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i)
    if (i < rp) { const float d = v[i] - mean; sq += d * d; }
  for (int o = 1; o < parts; o <<= 1) sq += __shfl_xor(sq, o, 64);
  const float r = 1.0f / sqrtf(sq / (float)P + q.eps);
  const int m0 = mt * rows + im * P, oc = nt * 32 + c;
  if (m0 < M && oc < p.Co) {
    const int n0 = m0 >> p.lp;
    const size_t base = ((size_t)n0 * p.Co + oc) * P + part * rp;
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (i < rp) {
        const float hh = (v[i] - mean) * r;
        q.xhat[base + i] = hh;
        p.out[base + i] = hh > 0.f ? hh : hh * q.slope;
      }
    if (part == 0) q.rstd[(size_t)n0 * p.Co + oc] = r;
  }
}

// ---------------------------------------------------------------------------------------------------------------- D
// IN: a problem with in_gx set also applies the InstanceNorm + LeakyReLU backward of the stage in front of the convolution (K9
// inorm_lrelu_bwd_kernel) to its data gradient: with 4x4 output maps a tile of 32 positions x 32 channels IS 64 complete 8x8 instances, so
// the workgroup that holds the tile's totals lays them out [instance][65] in LDS and runs K9's wavefront-per-instance code on them (lane =
// element, the same butterfly sums: the same bits as the two launches) -- one launch less on the chain that bounds the B=4 iteration.
constexpr int kDgInLds = 64 * 65;
template <bool IN>
__global__ __launch_bounds__(256) void conv4s2_dgrad_kernel(ConvP pa, ConvP pb, int na) {       // (two problems: see conv4s2_fwd_in_kernel)
  __shared__ float lds[IN && kDgInLds > kReduceLdsFloats ? kDgInLds : kReduceLdsFloats];
  const bool second = na >= 0 && (int)blockIdx.x >= na;
  const ConvP& p = second ? pb : pa;
  const int bid = (int)blockIdx.x - (second ? na : 0);
  const int t = threadIdx.x, w = t >> 6, lane = t & 63, col = lane & 31, h = lane >> 5;
  const int s = bid % p.S, tile = bid / p.S, nt = tile % p.tiles_n, mt = tile / p.tiles_n;
  const int P = 1 << p.lp, M = p.N << p.lp;
  // A operand: gy in the 3x3 neighbourhood of (a, b), channel co = 2q + h
  const int m = mt * 32 + col, mc = min(m, M - 1);
  const int n = mc >> p.lp, pp = mc & (P - 1), a0 = pp >> p.low, b0 = pp & (p.OW - 1);
  int off[9];
  bool ok[9];
#pragma unroll
  for (int dy = 0; dy < 3; ++dy)
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int oy = a0 + dy - 1, ox = b0 + dx - 1;
      ok[dy * 3 + dx] = m < M && oy >= 0 && oy < p.OH && ox >= 0 && ox < p.OW;
      off[dy * 3 + dx] = ok[dy * 3 + dx] ? oy * p.OW + ox : 0;
    }
  const float* ga = p.gy + (size_t)n * p.Co * P;
  // B operand: the 16 taps of W[co, ci]
  const int ci = min(nt * 32 + col, p.C - 1);
  const float* wb = p.w + (size_t)ci * 16;
  int qb, qe;
  k_range((p.Co + 1) >> 1, p.S, s, w, qb, qe);
  f32x16 acc[4] = {};
  constexpr int kQ = 4;                              // channel pairs in flight together
  for (int q0 = qb; q0 < qe; q0 += kQ) {
    float g[kQ][9];
    f32x4 wv[kQ][4];
    float live[kQ];
#pragma unroll
    for (int u = 0; u < kQ; ++u) {
      const int co = 2 * min(q0 + u, qe - 1) + h, cc = min(co, p.Co - 1);
      live[u] = (q0 + u < qe && co < p.Co) ? 1.f : 0.f;
      const float* gc = ga + (size_t)cc * P;
#pragma unroll
      for (int k = 0; k < 9; ++k) g[u][k] = gc[off[k]];
      const float* wc = wb + (size_t)cc * p.C * 16;
#pragma unroll
      for (int k = 0; k < 4; ++k) wv[u][k] = *reinterpret_cast<const f32x4*>(wc + 4 * k);
    }
#pragma unroll
    for (int u = 0; u < kQ; ++u) {
      if (q0 + u >= qe) continue;                    // (wave-uniform; the loads above stay clamped and unconditional)
#pragma unroll
      for (int py = 0; py < 2; ++py)
#pragma unroll
        for (int px = 0; px < 2; ++px)
#pragma unroll
          for (int jy = 0; jy < 2; ++jy)
#pragma unroll
            for (int jx = 0; jx < 2; ++jx) {
              // pixel row 2a+py receives output row a (ky = 1+py) and output row a-1 (py = 0, ky = 3) / a+1 (py = 1, ky = 0)
              const int dy = jy == 0 ? 1 : (py == 0 ? 0 : 2), ky = jy == 0 ? 1 + py : (py == 0 ? 3 : 0);
              const int dx = jx == 0 ? 1 : (px == 0 ? 0 : 2), kx = jx == 0 ? 1 + px : (px == 0 ? 3 : 0);
              const float av = ok[dy * 3 + dx] ? g[u][dy * 3 + dx] : 0.f;
              acc[py * 2 + px] = mfma(av, wv[u][ky][kx] * live[u], acc[py * 2 + px]);
            }
    }
  }
  float out[4][4];
  if (!reduce_tiles<4>(acc, out, lds, p, tile, s)) return;
  // thread: positions (a, b..b+3) of one image, channel ci -> pixels (2a+py, 2b .. 2b+7)
  const int m0 = mt * 32 + 8 * w + 4 * h, oc = nt * 32 + col;
  if (m0 < M && oc < p.C && !(IN && p.skip_out)) {
    const int n0 = m0 >> p.lp, p0 = m0 & (P - 1), a = p0 >> p.low, b = p0 & (p.OW - 1);
#pragma unroll
    for (int py = 0; py < 2; ++py) {
      float* dst = p.out + (((size_t)n0 * p.C + oc) * p.H + 2 * a + py) * p.W + 2 * b;
      *reinterpret_cast<f32x4*>(dst) = f32x4{out[py * 2][0], out[py * 2 + 1][0], out[py * 2][1], out[py * 2 + 1][1]};
      *reinterpret_cast<f32x4*>(dst + 4) = f32x4{out[py * 2][2], out[py * 2 + 1][2], out[py * 2][3], out[py * 2 + 1][3]};
    }
  }
  if (IN && p.in_gx != nullptr) {                    // (host: OH = OW = 4, so P = 16, a tile = images 2 mt, 2 mt + 1, b = 0)
    // wave w takes instances 16 w .. 16 w + 15 of the tile (instance il = image (il >> 5) of the tile, channel il & 31), lane = element:
    // ALL their operands are requested before anything else (a loop with the loads inside paid one memory latency per instance: the
    // workgroup that finishes the tile took 25 us longer and the B=4 iteration ran 4 % SLOWER than with two launches)
    float hh[16], ad[16], rs[16];
    size_t at[16];
    bool live[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int il = w * 16 + k, n = 2 * mt + (il >> 5), c = nt * 32 + (il & 31);
      live[k] = n < p.N && c < p.C;                  // (wave-uniform)
      at[k] = ((size_t)min(n, p.N - 1) * p.C + min(c, p.C - 1)) * 64 + lane;
      hh[k] = p.in_xhat[at[k]];
      rs[k] = p.in_rstd[at[k] >> 6];
      ad[k] = p.in_addend ? p.in_addend[at[k]] : 0.f;
    }
    __syncthreads();                                 // (reduce_tiles' LDS reads are done)
    const int im = w >> 1, a = 2 * (w & 1) + h;
    float* tl = lds + (im * 32 + col) * 65;
#pragma unroll
    for (int py = 0; py < 2; ++py)
#pragma unroll
      for (int px = 0; px < 2; ++px)
#pragma unroll
        for (int i = 0; i < 4; ++i) tl[(2 * a + py) * 8 + 2 * i + px] = out[py * 2 + px][i];
    __syncthreads();
    float av[16], sa[16], sah[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const float g = lds[(w * 16 + k) * 65 + lane];
      av[k] = g * (hh[k] > 0.f ? 1.0f : p.in_slope);
      sa[k] = 0.f; sah[k] = 0.f;
      sa[k] += av[k];
      sah[k] += av[k] * hh[k];
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1)
#pragma unroll
      for (int k = 0; k < 16; ++k) { sa[k] += __shfl_xor(sa[k], o, 64); sah[k] += __shfl_xor(sah[k], o, 64); }
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const float ma = sa[k] / 64.f, mah = sah[k] / 64.f;
      const float v = rs[k] * (av[k] - ma - hh[k] * mah);
      if (live[k]) p.in_gx[at[k]] = p.in_addend ? v + ad[k] : v;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------- G
__global__ __launch_bounds__(256) void conv4s2_wgrad_kernel(ConvP pa, ConvP pb, int na) {       // (two problems: see conv4s2_fwd_in_kernel)
  __shared__ float lds[kReduceLdsFloats];
  const bool second = na >= 0 && (int)blockIdx.x >= na;
  const ConvP& p = second ? pb : pa;
  const int bid = (int)blockIdx.x - (second ? na : 0);
  const int t = threadIdx.x, w = t >> 6, lane = t & 63, col = lane & 31, h = lane >> 5;
  const int s = bid % p.S, tile = bid / p.S, nt = tile % p.tiles_n, mt = tile / p.tiles_n;
  const int P = 1 << p.lp, M = p.N << p.lp, HW = p.H * p.W, KW = p.C * 16;
  // A operand: gy[n, co, p .. p+3]  (rows = co);  B operand: x gathered at (ci, ky, kx) = column j
  const int co = min(mt * 32 + col, p.Co - 1);
  const int j = min(nt * 32 + col, KW - 1), ci = j >> 4, ky = (j >> 2) & 3, kx = j & 3;
  const float* xb = p.x + (size_t)ci * HW;
  int bb, be;                                        // blocks of 8 positions: this half takes positions 8*blk + 4h + 0..3
  k_range((M + 7) >> 3, p.S, s, w, bb, be);
  f32x16 acc[1] = {};
  for (int k0 = bb; k0 < be; k0 += 2) {
    f32x4 g[2];
    float xv[2][4];
    bool okx[2][4];
    float live[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int blk = min(k0 + u, be - 1), m0 = blk * 8 + 4 * h, mc = min(m0, M - 4);
      live[u] = (k0 + u < be && m0 < M) ? 1.f : 0.f;
      const int n = mc >> p.lp, pp = mc & (P - 1), oy = pp >> p.low, ox = pp & (p.OW - 1);
      g[u] = *reinterpret_cast<const f32x4*>(p.gy + ((size_t)n * p.Co + co) * P + pp);
      const int iy = 2 * oy - 1 + ky;
      const float* xr = xb + (size_t)n * p.C * HW;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ix = 2 * (ox + i) - 1 + kx;
        okx[u][i] = iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        xv[u][i] = xr[okx[u][i] ? iy * p.W + ix : 0];
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[0] = mfma(g[u][i] * live[u], okx[u][i] ? xv[u][i] : 0.f, acc[0]);
  }
  float out[1][4];
  if (!reduce_tiles<1>(acc, out, lds, p, tile, s)) return;
  const int r0 = mt * 32 + 8 * w + 4 * h, oc = nt * 32 + col;
  if (oc < KW)
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (r0 + i < p.Co) p.out[(size_t)(r0 + i) * KW + oc] = out[0][i];
}

// ---------------------------------------------------------------------------------------------------------------- K12
// 3x3 / stride 1 / padding 1 convolution + bias + ReLU of the perceptual-loss feature network (reference
// layers/perceptual_loss.py:8-45: torchvision VGG19 features[:15], frozen; model/nerf_adapt_st_gan.py:758-766 calls it on the
// rendered and the real patches).  Same machinery: rows = positions, columns = output channels, k = (channel pair, 9 taps),
// the lane halves take the even / odd channel of a pair.  T = false: y = relu?(conv(x, w) + bias).  T = true: the data
// gradient gx = conv^T(gy * (mask > 0), w) -- the weight is read transposed and flipped, and the ReLU derivative of the
// layer's own output is applied while the cotangent is gathered.  (No weight gradient: the network is frozen.)
struct Conv3P {
  const float* in;     // F: x [N,C,H,W];  T: gy [N,Co,H,W]
  const float* w;      // [Co,C,3,3]
  const float* bias;   // F: [Co] or null
  const float* mask;   // T: forward output [N,Co,H,W] (gy counts where mask > 0) or null
  float* out;          // F: [N,Co,H,W];  T: [N,C,H,W]
  float* ws;
  unsigned* cnt;
  int N, C, H, W, Co;
  int lw, lp;          // log2(W), log2(H * W)
  int S, tiles_n, relu;
};

template <bool T>
__global__ __launch_bounds__(256) void conv3s1_kernel(Conv3P p) {
  __shared__ float lds[kReduceLdsFloats];
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
  const float* ma = T && p.mask ? p.mask + (size_t)n * CK * P : nullptr;
  const int cn = min(nt * 32 + col, CN - 1);
  // weight element of (produced channel cn, contracted channel ck, neighbourhood index k)
  const size_t wbase = T ? (size_t)cn * 9 + 8 : (size_t)cn * p.C * 9;
  const size_t wstep = T ? (size_t)p.C * 9 : 9;                 // per contracted channel
  int qb, qe;
  k_range((CK + 1) >> 1, p.S, s, w, qb, qe);
  f32x16 acc[1] = {};
  for (int q0 = qb; q0 < qe; q0 += 2) {
    float a[2][9], b[2][9], mk[2][9];
    float live[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int ck = 2 * min(q0 + u, qe - 1) + h, cc = min(ck, CK - 1);
      live[u] = (q0 + u < qe && ck < CK) ? 1.f : 0.f;
      const float* ic = ia + (size_t)cc * P;
      const float* wc = p.w + wbase + wstep * cc;
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        a[u][k] = ic[off[k]];
        b[u][k] = T ? *(wc - k) : wc[k];
        if (T) mk[u][k] = ma ? ma[(size_t)cc * P + off[k]] : 1.f;
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        const bool on = T ? (ok[k] && mk[u][k] > 0.f) : ok[k];
        acc[0] = mfma(on ? a[u][k] : 0.f, b[u][k] * live[u], acc[0]);
      }
  }
  float out[1][4];
  ConvP rp;
  rp.ws = p.ws; rp.cnt = p.cnt; rp.S = p.S;
  if (!reduce_tiles<1>(acc, out, lds, rp, tile, s)) return;
  const int m0 = mt * 32 + 8 * w + 4 * h, oc = nt * 32 + col;
  if (m0 < M && oc < CN) {
    f32x4 v = {out[0][0], out[0][1], out[0][2], out[0][3]};
    if (!T) {
      if (p.bias) v += p.bias[oc];
      if (p.relu)
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.f);
    }
    const int n0 = m0 >> p.lp, p0 = m0 & (P - 1);
    *reinterpret_cast<f32x4*>(p.out + ((size_t)n0 * CN + oc) * P + p0) = v;
  }
}

}  // namespace

extern "C" {

static int conv_plan(const tp_conv4s2_args* a, int op, Plan* q, ConvP* p) {
  TP_REQUIRE(a && a->N > 0 && a->C > 0 && a->Co > 0, "bad sizes");
  const int OH = a->H / 2, OW = a->W / 2;
  const int lw = ilog2(a->W), low = ilog2(OW), lh = ilog2(OH);
  TP_REQUIRE(a->H >= 8 && a->W >= 8 && lw >= 0 && low >= 0 && lh >= 0 && ilog2(a->H) >= 0, "H and W must be powers of two >= 8");
  TP_REQUIRE((int64_t)a->N * OH * OW <= (int64_t)1 << 28, "too many positions");
  p->N = a->N; p->C = a->C; p->H = a->H; p->W = a->W; p->Co = a->Co; p->OH = OH; p->OW = OW;
  p->lw = lw; p->low = low; p->lp = lh + low;
  const int M = a->N * OH * OW;
  if (op == 0) *q = plan(M, a->Co, a->C, 1, 256, 4);                    // k unit: one ci (8 MFMAs)
  else if (op == 1) *q = plan(M, a->C, (a->Co + 1) / 2, 4, 256, 2);     // k unit: a co pair (16 MFMAs)
  else *q = plan(a->Co, a->C * 16, (M + 7) / 8, 1, 256, 8);             // k unit: 8 positions (4 MFMAs)
  p->S = q->S; p->tiles_n = q->tiles_n;
  return 0;
}

int64_t tp_conv4s2_workspace(const tp_conv4s2_args* a, int op, int64_t* n_counters) {
  Plan q; ConvP p;
  if (conv_plan(a, op, &q, &p) != 0) return -1;
  if (n_counters) *n_counters = (int64_t)q.tiles_m * q.tiles_n;
  return (int64_t)q.ws_floats;
}

static int conv_fill(const tp_conv4s2_args* a, int op, Plan* q, ConvP* p) {
  const int rc = conv_plan(a, op, q, p);
  if (rc != 0) return rc;
  TP_REQUIRE(a->out && a->counters && (!q->ws_floats || a->workspace), "out / counters / workspace missing");
  TP_REQUIRE((op == 1 || a->x) && (op == 2 || a->w) && (op == 0 || a->gy), "operand missing");
  p->x = a->x; p->w = a->w; p->gy = a->gy; p->out = a->out; p->ws = a->workspace; p->cnt = (unsigned*)a->counters;
  p->x_copy = nullptr;
  p->in_xhat = a->in_xhat; p->in_rstd = a->in_rstd; p->in_addend = a->in_addend; p->in_gx = a->in_gx; p->in_slope = a->in_slope; p->skip_out = a->skip_out;
  if (a->in_gx != nullptr) {
    TP_REQUIRE(op == 1 && a->in_xhat && a->in_rstd && a->H == 8 && a->W == 8, "the fused InstanceNorm backward: data gradient onto 8x8 maps, xhat and rstd given");
  } else {
    TP_REQUIRE(!a->skip_out, "skip_out without the fused InstanceNorm backward");
  }
  return 0;
}
// one problem (b == nullptr) or two in one launch (different counters / workspaces: they run side by side)
static int conv_launch(const tp_conv4s2_args* a, const tp_conv4s2_args* b, int op, tp_stream_t stream) {
  Plan qa, qb; ConvP pa, pb;
  if (const int rc = conv_fill(a, op, &qa, &pa)) return rc;
  const unsigned ga = (unsigned)(qa.tiles_m * qa.tiles_n * qa.S);
  unsigned gb = 0;
  int na = -1;
  pb = pa;
  if (b != nullptr) {
    TP_REQUIRE(op != 0, "pairs: data gradient, weight gradient, forward + InstanceNorm");
    if (const int rc = conv_fill(b, op, &qb, &pb)) return rc;
    TP_REQUIRE(a->counters != b->counters && (a->workspace != b->workspace || !a->workspace) && a->out != b->out, "the two problems of a pair need their own counters / workspace / output");
    gb = (unsigned)(qb.tiles_m * qb.tiles_n * qb.S);
    na = (int)ga;
  }
  const dim3 grid(ga + gb), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (op == 0) hipLaunchKernelGGL(conv4s2_fwd_kernel, grid, block, 0, st, pa);
  else if (op == 1 && (pa.in_gx || pb.in_gx)) hipLaunchKernelGGL(conv4s2_dgrad_kernel<true>, grid, block, 0, st, pa, pb, na);
  else if (op == 1) hipLaunchKernelGGL(conv4s2_dgrad_kernel<false>, grid, block, 0, st, pa, pb, na);
  else hipLaunchKernelGGL(conv4s2_wgrad_kernel, grid, block, 0, st, pa, pb, na);
  return tp::check_launch(op == 0 ? "tp_conv4s2_fwd" : op == 1 ? "tp_conv4s2_dgrad" : "tp_conv4s2_wgrad");
}

int tp_conv4s2_fwd(const tp_conv4s2_args* a, tp_stream_t stream) { return conv_launch(a, nullptr, 0, stream); }

// row tiles per workgroup of the fused convolution + InstanceNorm launch (0: this map size is not covered)
static int fwd_in_nt(const tp_conv4s2_args* a) {
  const int P = (a->H / 2) * (a->W / 2);
  return P == 16 ? 1 : P == 64 ? 2 : 0;
}
static int conv_in_plan(const tp_conv4s2_args* a, Plan* q, ConvP* p) {
  if (const int rc = conv_plan(a, 0, q, p)) return rc;
  const int nt = fwd_in_nt(a);
  TP_REQUIRE(nt > 0, "the fused convolution + InstanceNorm launch covers 4x4 and 8x8 output maps");
  const int M = a->N * p->OH * p->OW;
  TP_REQUIRE(M % (32 * nt) == 0 || nt == 1, "whole images per workgroup needed");
  *q = plan(M, a->Co, a->C, nt, 256, 4);
  q->tiles_m = (M + 32 * nt - 1) / (32 * nt);
  const int tiles = q->tiles_m * q->tiles_n;
  const int target = conv_target_wgs(256);
  int S = tiles >= target ? 1 : (target + tiles - 1) / tiles;
  const int max_s = (a->C + 15) / 16;
  if (S > max_s) S = max_s;
  if (S < 1) S = 1;
  q->S = S;
  q->ws_floats = S > 1 ? (size_t)tiles * S * nt * 4 * 256 : 0;
  p->S = S; p->tiles_n = q->tiles_n;
  return 0;
}
int64_t tp_conv4s2_fwd_inorm_workspace(const tp_conv4s2_args* a, int64_t* n_counters) {
  Plan q; ConvP p;
  if (conv_in_plan(a, &q, &p) != 0) return -1;
  if (n_counters) *n_counters = (int64_t)q.tiles_m * q.tiles_n;
  return (int64_t)q.ws_floats;
}
static int conv_in_fill(const tp_conv4s2_args* a, float* xhat, float* rstd, Plan* q, ConvP* p) {
  if (const int rc = conv_in_plan(a, q, p)) return rc;
  TP_REQUIRE(a->x && a->w && a->out && xhat && rstd && a->counters && (!q->ws_floats || a->workspace), "operand / counters / workspace missing");
  TP_REQUIRE((a->N * p->OH * p->OW) % 16 == 0, "whole instances per tile needed");
  p->x = a->x; p->w = a->w; p->gy = nullptr; p->out = a->out; p->ws = a->workspace; p->cnt = (unsigned*)a->counters;
  p->x_copy = a->x_copy;
  TP_REQUIRE(a->x_copy == nullptr || ((((int64_t)a->N * a->C * a->H * a->W) & 3) == 0 && (((uintptr_t)a->x | (uintptr_t)a->x_copy) & 15) == 0 && a->x_copy != a->x),
             "x_copy: 16-byte aligned, a multiple of 4 floats, not x itself");
  return 0;
}
static int conv_in_launch(const tp_conv4s2_args* a, float* xhat_a, float* rstd_a, const tp_conv4s2_args* b, float* xhat_b, float* rstd_b,
                          float eps, float slope, tp_stream_t stream) {
  Plan qa, qb; ConvP pa, pb;
  if (const int rc = conv_in_fill(a, xhat_a, rstd_a, &qa, &pa)) return rc;
  InP ia{xhat_a, rstd_a, eps, slope}, ib = ia;
  unsigned ga = (unsigned)(qa.tiles_m * qa.tiles_n * qa.S), gb = 0;
  int na = -1;
  pb = pa;
  if (b != nullptr) {
    if (const int rc = conv_in_fill(b, xhat_b, rstd_b, &qb, &pb)) return rc;
    TP_REQUIRE(fwd_in_nt(a) == fwd_in_nt(b), "the two problems of a pair need the same map size");
    TP_REQUIRE(a->counters != b->counters && (a->workspace != b->workspace || !a->workspace) && a->out != b->out, "the two problems of a pair need their own counters / workspace / output");
    ib = InP{xhat_b, rstd_b, eps, slope};
    gb = (unsigned)(qb.tiles_m * qb.tiles_n * qb.S);
    na = (int)ga;
  }
  const dim3 grid(ga + gb), block(256);
  if (fwd_in_nt(a) == 1) hipLaunchKernelGGL(conv4s2_fwd_in_kernel<1>, grid, block, 0, (hipStream_t)stream, pa, ia, pb, ib, na);
  else hipLaunchKernelGGL(conv4s2_fwd_in_kernel<2>, grid, block, 0, (hipStream_t)stream, pa, ia, pb, ib, na);
  return tp::check_launch("tp_conv4s2_fwd_inorm");
}
int tp_conv4s2_fwd_inorm(const tp_conv4s2_args* a, float eps, float slope, float* xhat, float* rstd, tp_stream_t stream) {
  return conv_in_launch(a, xhat, rstd, nullptr, nullptr, nullptr, eps, slope, stream);
}
int tp_conv4s2_fwd_inorm_pair(const tp_conv4s2_args* a, float* xhat_a, float* rstd_a, const tp_conv4s2_args* b, float* xhat_b, float* rstd_b,
                              float eps, float slope, tp_stream_t stream) {
  TP_REQUIRE(a && b, "null argument");
  return conv_in_launch(a, xhat_a, rstd_a, b, xhat_b, rstd_b, eps, slope, stream);
}
int tp_conv4s2_dgrad(const tp_conv4s2_args* a, tp_stream_t stream) { return conv_launch(a, nullptr, 1, stream); }
int tp_conv4s2_wgrad(const tp_conv4s2_args* a, tp_stream_t stream) { return conv_launch(a, nullptr, 2, stream); }
int tp_conv4s2_dgrad_pair(const tp_conv4s2_args* a, const tp_conv4s2_args* b, tp_stream_t stream) {
  TP_REQUIRE(a && b, "null argument");
  return conv_launch(a, b, 1, stream);
}
int tp_conv4s2_wgrad_pair(const tp_conv4s2_args* a, const tp_conv4s2_args* b, tp_stream_t stream) {
  TP_REQUIRE(a && b, "null argument");
  return conv_launch(a, b, 2, stream);
}

static int conv3_plan(const tp_conv3s1_args* a, int transposed, Plan* q, Conv3P* p) {
  TP_REQUIRE(a && a->N > 0 && a->C > 0 && a->Co > 0, "bad sizes");
  const int lw = ilog2(a->W), lh = ilog2(a->H);
  TP_REQUIRE(a->H >= 4 && a->W >= 4 && lw >= 0 && lh >= 0, "H and W must be powers of two >= 4");
  TP_REQUIRE((int64_t)a->N * a->H * a->W <= (int64_t)1 << 28, "too many positions");
  p->N = a->N; p->C = a->C; p->H = a->H; p->W = a->W; p->Co = a->Co; p->lw = lw; p->lp = lw + lh; p->relu = a->relu;
  const int M = a->N * a->H * a->W;
  *q = plan(M, transposed ? a->C : a->Co, ((transposed ? a->Co : a->C) + 1) / 2, 1, 256, 2);   // k unit: a channel pair (9 MFMAs)
  p->S = q->S; p->tiles_n = q->tiles_n;
  return 0;
}

int64_t tp_conv3s1_workspace(const tp_conv3s1_args* a, int op, int64_t* n_counters) {
  Plan q; Conv3P p;
  if (op != TP_CONV_FWD && op != TP_CONV_DGRAD) return -1;
  if (conv3_plan(a, op == TP_CONV_DGRAD, &q, &p) != 0) return -1;
  if (n_counters) *n_counters = (int64_t)q.tiles_m * q.tiles_n;
  return (int64_t)q.ws_floats;
}

static int conv3_launch(const tp_conv3s1_args* a, int transposed, tp_stream_t stream) {
  Plan q; Conv3P p;
  const int rc = conv3_plan(a, transposed, &q, &p);
  if (rc != 0) return rc;
  TP_REQUIRE(a->in && a->w && a->out && a->counters && (!q.ws_floats || a->workspace), "operand / counters / workspace missing");
  p.in = a->in; p.w = a->w; p.bias = transposed ? nullptr : a->bias; p.mask = transposed ? a->mask : nullptr; p.out = a->out;
  p.ws = a->workspace; p.cnt = (unsigned*)a->counters;
  const dim3 grid((unsigned)(q.tiles_m * q.tiles_n * q.S)), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (transposed) hipLaunchKernelGGL(conv3s1_kernel<true>, grid, block, 0, st, p);
  else hipLaunchKernelGGL(conv3s1_kernel<false>, grid, block, 0, st, p);
  return tp::check_launch(transposed ? "tp_conv3s1_dgrad" : "tp_conv3s1_fwd");
}

int tp_conv3s1_fwd(const tp_conv3s1_args* a, tp_stream_t stream) { return conv3_launch(a, 0, stream); }
int tp_conv3s1_dgrad(const tp_conv3s1_args* a, tp_stream_t stream) { return conv3_launch(a, 1, stream); }
}
