// K6: evaluation metrics after the render: squared error and SSIM sums of the static render
// against the masked image, reference model/nerf_adapt_st_gan.py:340-362 with
// external/pohsun_ssim/pytorch_ssim/__init__.py:7-37 (11x11 Gaussian sigma 1.5 as a depthwise conv with ZERO
// padding, C1 = 0.01^2, C2 = 0.03^2).  The optional resize of evaluate_full (bilinear align_corners=False for
// the colours, 'nearest' for the mask; 480x640 in the reference) is fused into the tile load, so nothing but the
// two sums per image leaves the chip.
//
// One workgroup = one 32x32 output tile of one image and one colour channel loop: tile + 5-pixel halo of both
// images in LDS (42x42x2), the five moments (x, y, xx, yy, xy) filtered separably: horizontal pass into LDS
// [42 rows][32 cols][5], vertical pass per thread.  HBM-bound and tiny (7.4 MB per 480x640 pair); the point of
// the kernel is that PSNR + SSIM cost one launch instead of ~25.  Per-tile partial sums are reduced in fixed
// order by a second one-workgroup launch (deterministic, no atomics).
#include "tp_common.h"

namespace {
constexpr int kTile = 32, kHalo = 5, kIn = kTile + 2 * kHalo;   // 42
constexpr int kWin = 11;

struct Win { float w[kWin]; };

__device__ __forceinline__ float src_coord(int dst, float scale) {   // area_pixel_compute_source_index, align_corners=False
  const float s = scale * ((float)dst + 0.5f) - 0.5f;
  return s < 0.0f ? 0.0f : s;
}

__global__ __launch_bounds__(256) void eval_metrics_kernel(tp_eval_metrics_args a, Win win, float* partial) {
  __shared__ float t1[kIn][kIn + 1], t2[kIn][kIn + 1];
  __shared__ float hz[5][kIn][kTile + 1];
  __shared__ float red[2][256];
  const int tid = threadIdx.x;
  const int b = blockIdx.z, ty0 = blockIdx.y * kTile, tx0 = blockIdx.x * kTile;
  const int H = a.out_h, W = a.out_w, h = a.h, w = a.w;
  const bool resize = (H != h) || (W != w);
  const float sy = (float)h / (float)H, sx = (float)w / (float)W;
  const float* rgb = a.rgb_static + (int64_t)b * h * w * 3;       // [h,w,3]
  const float* msk = a.obj_mask + (int64_t)b * h * w;
  float sse = 0.0f, ssim = 0.0f;

  for (int c = 0; c < 3; ++c) {
    const float* img = a.image + ((int64_t)b * 3 + c) * h * w;
    __syncthreads();
    for (int i = tid; i < kIn * kIn; i += 256) {
      const int ly = i / kIn, lx = i - ly * kIn;
      const int Y = ty0 + ly - kHalo, X = tx0 + lx - kHalo;
      float v1 = 0.0f, v2 = 0.0f;                                   // zero padding outside the (resized) image
      if (Y >= 0 && Y < H && X >= 0 && X < W) {
        if (!resize) {
          v1 = rgb[((int64_t)Y * w + X) * 3 + c];
          v2 = img[(int64_t)Y * w + X] * msk[(int64_t)Y * w + X];
        } else {
          const float fy = src_coord(Y, sy), fx = src_coord(X, sx);
          const int y0 = (int)fy, x0 = (int)fx;
          const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
          const float ly1 = fy - (float)y0, lx1 = fx - (float)x0, ly0 = 1.0f - ly1, lx0 = 1.0f - lx1;
          const float r00 = rgb[((int64_t)y0 * w + x0) * 3 + c], r01 = rgb[((int64_t)y0 * w + x1) * 3 + c];
          const float r10 = rgb[((int64_t)y1 * w + x0) * 3 + c], r11 = rgb[((int64_t)y1 * w + x1) * 3 + c];
          v1 = ly0 * (lx0 * r00 + lx1 * r01) + ly1 * (lx0 * r10 + lx1 * r11);
          const float i00 = img[(int64_t)y0 * w + x0], i01 = img[(int64_t)y0 * w + x1];
          const float i10 = img[(int64_t)y1 * w + x0], i11 = img[(int64_t)y1 * w + x1];
          const float vi = ly0 * (lx0 * i00 + lx1 * i01) + ly1 * (lx0 * i10 + lx1 * i11);
          // 'nearest': floor(dst * scale), clamped (upsample_nearest2d, legacy index rule)
          const int ny = min((int)floorf((float)Y * sy), h - 1), nx = min((int)floorf((float)X * sx), w - 1);
          v2 = vi * msk[(int64_t)ny * w + nx];
        }
      }
      t1[ly][lx] = v1;
      t2[ly][lx] = v2;
    }
    __syncthreads();
    // squared error of the tile's own pixels
    for (int i = tid; i < kTile * kTile; i += 256) {
      const int ly = i / kTile, lx = i - ly * kTile;
      if (ty0 + ly < H && tx0 + lx < W) {
        const float d = t1[ly + kHalo][lx + kHalo] - t2[ly + kHalo][lx + kHalo];
        sse += d * d;
      }
    }
    // horizontal 11-tap pass of the five moments
    for (int i = tid; i < kIn * kTile; i += 256) {
      const int ly = i / kTile, lx = i - ly * kTile;
      float m1 = 0.f, m2 = 0.f, m11 = 0.f, m22 = 0.f, m12 = 0.f;
#pragma unroll
      for (int k = 0; k < kWin; ++k) {
        const float x = t1[ly][lx + k], y = t2[ly][lx + k], g = win.w[k];
        m1 += g * x; m2 += g * y; m11 += g * (x * x); m22 += g * (y * y); m12 += g * (x * y);
      }
      hz[0][ly][lx] = m1; hz[1][ly][lx] = m2; hz[2][ly][lx] = m11; hz[3][ly][lx] = m22; hz[4][ly][lx] = m12;
    }
    __syncthreads();
    // vertical pass + SSIM
    for (int i = tid; i < kTile * kTile; i += 256) {
      const int ly = i / kTile, lx = i - ly * kTile;
      if (ty0 + ly < H && tx0 + lx < W) {
        float mu1 = 0.f, mu2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
        for (int k = 0; k < kWin; ++k) {
          const float g = win.w[k];
          mu1 += g * hz[0][ly + k][lx]; mu2 += g * hz[1][ly + k][lx];
          e11 += g * hz[2][ly + k][lx]; e22 += g * hz[3][ly + k][lx]; e12 += g * hz[4][ly + k][lx];
        }
        const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
        const float s1 = e11 - mu1_sq, s2 = e22 - mu2_sq, s12 = e12 - mu12;
        const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
        ssim += ((2.0f * mu12 + C1) * (2.0f * s12 + C2)) / ((mu1_sq + mu2_sq + C1) * (s1 + s2 + C2));
      }
    }
  }
  red[0][tid] = sse;
  red[1][tid] = ssim;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) { red[0][tid] += red[0][tid + s]; red[1][tid] += red[1][tid + s]; }
    __syncthreads();
  }
  if (tid == 0) {
    const int64_t t = ((int64_t)b * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    partial[2 * t] = red[0][0];
    partial[2 * t + 1] = red[1][0];
  }
}

// fixed-order reduction of the per-tile partials: out[b] = (sum sq. error, sum ssim) in fp64 (thread t adds tiles
// t, t+64, ... in order, then a tree over the 64 lanes)
__global__ __launch_bounds__(64) void eval_metrics_finalize(const float* partial, int tiles, double* out) {
  __shared__ double red[2][64];
  const int b = blockIdx.x, tid = threadIdx.x;
  double s0 = 0.0, s1 = 0.0;
  for (int t = tid; t < tiles; t += 64) { s0 += partial[2 * ((int64_t)b * tiles + t)]; s1 += partial[2 * ((int64_t)b * tiles + t) + 1]; }
  red[0][tid] = s0;
  red[1][tid] = s1;
  __syncthreads();
  for (int s = 32; s > 0; s >>= 1) {
    if (tid < s) { red[0][tid] += red[0][tid + s]; red[1][tid] += red[1][tid + s]; }
    __syncthreads();
  }
  if (tid == 0) { out[2 * b] = red[0][0]; out[2 * b + 1] = red[1][0]; }
}

}  // namespace

extern "C" int64_t tp_eval_metrics_workspace_bytes(int B, int out_h, int out_w) {
  const int64_t tiles = (int64_t)((out_h + kTile - 1) / kTile) * ((out_w + kTile - 1) / kTile);
  return B * tiles * 2 * (int64_t)sizeof(float);
}

extern "C" int tp_eval_metrics(const tp_eval_metrics_args* a, tp_stream_t stream) {
  TP_REQUIRE(a != nullptr, "tp_eval_metrics: null args");
  TP_REQUIRE(a->rgb_static && a->image && a->obj_mask && a->workspace && a->out, "tp_eval_metrics: null pointer");
  TP_REQUIRE(a->B > 0 && a->h > 0 && a->w > 0 && a->out_h > 0 && a->out_w > 0, "tp_eval_metrics: bad sizes");
  Win win;
  double sum = 0.0, g[kWin];
  for (int x = 0; x < kWin; ++x) { g[x] = exp(-(double)((x - kWin / 2) * (x - kWin / 2)) / (2.0 * 1.5 * 1.5)); sum += g[x]; }
  // the reference builds the window in fp32 (torch.Tensor of python floats, then / sum)
  float gf[kWin], sf = 0.0f;
  for (int x = 0; x < kWin; ++x) { gf[x] = (float)g[x]; sf += gf[x]; }
  for (int x = 0; x < kWin; ++x) win.w[x] = gf[x] / sf;
  (void)sum;
  const dim3 grid((a->out_w + kTile - 1) / kTile, (a->out_h + kTile - 1) / kTile, a->B);
  hipLaunchKernelGGL(eval_metrics_kernel, grid, dim3(256), 0, (hipStream_t)stream, *a, win, (float*)a->workspace);
  int rc = tp::check_launch("tp_eval_metrics");
  if (rc) return rc;
  hipLaunchKernelGGL(eval_metrics_finalize, dim3(a->B), dim3(64), 0, (hipStream_t)stream, (const float*)a->workspace,
                     (int)(grid.x * grid.y), a->out);
  return tp::check_launch("tp_eval_metrics(finalize)");
}
