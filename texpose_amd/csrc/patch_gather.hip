// K5: patch gather for the photometric / perceptual / PatchGAN inputs (gfx950).
//
// Replaces the 8 F.grid_sample calls the reference issues per training step, twice (nerf step and
// discriminator step): model/nerf_adapt_st_gan.py:444-461 (sample_geometry), :726-745
// (compute_loss).  image / image_syn / nocs / normal are bilinear with align_corners=True;
// obj_mask / mask_syn are binarised (>0) and sampled 'nearest' with the default
// align_corners=False (the two conventions disagree by up to half a
// pixel and x=+1 rounds out of range).  One thread per patch pixel gathers all 14 channels, so
// the coordinate algebra is done once instead of 8 times; output is channel-major [B,14,P].
// HBM-bound gather: <= 14 ch x 4 taps x 4 B read + 56 B written per patch pixel.
#include "tp_common.h"

namespace {

// Workgroups are dispatched round-robin over the 8 XCDs (each with its own L2): with the natural order the 16 workgroups of one
// 64 x 64 patch -- which sample overlapping rows of the same image region -- land on all eight, and every L2 fetches the region for
// itself (PMC: 5x the algorithmic bytes).  `xcd_major` (set when the grid is a multiple of 8) gives XCD x the CONSECUTIVE logical
// blocks [x n / 8, (x + 1) n / 8): an image's blocks share one L2.
template <bool SPLIT>
__global__ void patch_gather_kernel(tp_patch_gather_args a, int xcd_major) {
  const unsigned nb = gridDim.x;
  const unsigned lb = xcd_major ? (blockIdx.x & 7u) * (nb >> 3) + (blockIdx.x >> 3) : blockIdx.x;
  // SPLIT (small launches: latency-sized): four threads per patch pixel, one per source image (image, synthetic image, nocs, normal: 3
  // channels x 4 taps each) -- one thread per pixel issued ~56 loads in a row: 7.8 us for the 1,024 pixels of a B=4 step, behind the
  // render on BOTH chains of the training iteration (4.5 us split).  Large launches are bandwidth-sized and keep one thread per pixel
  // (B=32, 64 x 64 patches: 15.6 us, 20.9 split).
  const int64_t q4 = (int64_t)lb * blockDim.x + threadIdx.x;
  const int64_t q = SPLIT ? q4 >> 2 : q4;
  const int64_t total = (int64_t)a.B * a.P;
  if (q >= total) return;
  const int b = (int)(q / a.P), p = (int)(q - (int64_t)b * a.P);
  const int H = a.H, W = a.W;
  const int64_t HW = (int64_t)H * W;
  const float x = a.coords[2 * q], y = a.coords[2 * q + 1];
  // bilinear, align_corners=True
  const float ix = tp::mul_rn(tp::add_rn(x, 1.0f), (float)(W - 1) * 0.5f);
  const float iy = tp::mul_rn(tp::add_rn(y, 1.0f), (float)(H - 1) * 0.5f);
  const float fx = floorf(ix), fy = floorf(iy);
  const float w = tp::sub_rn(ix, fx), e = tp::sub_rn(1.0f, w), n = tp::sub_rn(iy, fy), s = tp::sub_rn(1.0f, n);
  const float nw = tp::mul_rn(e, s), ne = tp::mul_rn(w, s), sw = tp::mul_rn(e, n), se = tp::mul_rn(w, n);
  const int x0 = (int)fminf(fmaxf(fx, -2.0f), (float)W + 1.0f), y0 = (int)fminf(fmaxf(fy, -2.0f), (float)H + 1.0f);
  const bool xin0 = x0 >= 0 && x0 <= W - 1, xin1 = x0 + 1 >= 0 && x0 + 1 <= W - 1;
  const bool yin0 = y0 >= 0 && y0 <= H - 1, yin1 = y0 + 1 >= 0 && y0 + 1 <= H - 1;
  const bool ok00 = xin0 && yin0, ok10 = xin1 && yin0, ok01 = xin0 && yin1, ok11 = xin1 && yin1;
  const int64_t o00 = (int64_t)y0 * W + x0;
  // nearest, align_corners=False, round half to even
  const float jx = rintf(tp::sub_rn(tp::mul_rn(tp::add_rn(x, 1.0f), (float)W * 0.5f), 0.5f));
  const float jy = rintf(tp::sub_rn(tp::mul_rn(tp::add_rn(y, 1.0f), (float)H * 0.5f), 0.5f));
  const bool nok = jx >= 0.0f && jx <= (float)(W - 1) && jy >= 0.0f && jy <= (float)(H - 1);
  const int64_t on = nok ? (int64_t)jy * W + (int64_t)jx : 0;
  const float m = nok ? (a.obj_mask[b * HW + on] > 0.0f ? 1.0f : 0.0f) : 0.0f;
  const float ms = nok ? (a.mask_syn[b * HW + on] > 0.0f ? 1.0f : 0.0f) : 0.0f;

  float* out = a.out + (int64_t)b * 14 * a.P + p;
#pragma unroll
  for (int gi = 0; gi < (SPLIT ? 1 : 4); ++gi) {
  const int grp = SPLIT ? (int)(q4 & 3) : gi;
  const float* src = grp == 0 ? a.image : grp == 1 ? a.image_syn : grp == 2 ? a.nocs : a.normal;
  float val[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float* pl = src + ((int64_t)b * 3 + c) * HW;
    float r = tp::mul_rn(ok00 ? pl[o00] : 0.0f, nw);
    r = tp::fma_rn(ok10 ? pl[o00 + 1] : 0.0f, ne, r);
    r = tp::fma_rn(ok01 ? pl[o00 + W] : 0.0f, sw, r);
    r = tp::fma_rn(ok11 ? pl[o00 + W + 1] : 0.0f, se, r);
    if (grp >= 2) r = tp::mul_rn(r, ms);  // nocs_sample / normal_sample = gathered * mask_syn (:459-460)
    out[(int64_t)(grp * 3 + c) * a.P] = r;
    val[c] = r;
  }
  if (grp == 0) {
    out[(int64_t)12 * a.P] = m;
    out[(int64_t)13 * a.P] = ms;
  }
  if (a.disc_rgb != nullptr) {                       // (csrc/train_misc.hip disc_inputs_kernel on the values this thread has just formed)
    const int nc = a.disc_geo ? 9 : 3;
    float* real = a.disc_real + (int64_t)b * nc * a.P + p;
    float* fake = a.disc_fake + (int64_t)b * nc * a.P + p;
    if (grp == 0) {
      const float pad = (ms == 1.f && m == 0.f) ? 1.f : 0.f;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float r = a.disc_rgb[q * 3 + c];
        real[(int64_t)c * a.P] = tp::add_rn(tp::mul_rn(val[c], m), tp::mul_rn(r, pad));
        fake[(int64_t)c * a.P] = r;
      }
    } else if (grp >= 2 && a.disc_geo) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        real[(int64_t)(3 * (grp - 1) + c) * a.P] = val[c];
        fake[(int64_t)(3 * (grp - 1) + c) * a.P] = val[c];
      }
    }
  }
  }
}

}  // namespace

extern "C" int tp_patch_gather(const tp_patch_gather_args* a, tp_stream_t stream) {
  TP_REQUIRE(a && a->coords && a->image && a->image_syn && a->nocs && a->normal && a->obj_mask && a->mask_syn &&
                 a->out, "null pointer");
  TP_REQUIRE(a->B > 0 && a->P > 0 && a->H > 0 && a->W > 0, "bad sizes");
  TP_REQUIRE(a->disc_rgb == nullptr || (a->disc_real && a->disc_fake), "disc_rgb given: disc_real and disc_fake expected");
  const bool split = (int64_t)a->B * a->P <= 16384;
  const int64_t total = (int64_t)a->B * a->P * (split ? 4 : 1);
  const unsigned blocks = (unsigned)((total + 255) / 256);
  if (split) hipLaunchKernelGGL(patch_gather_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, *a, (blocks % 8 == 0 && blocks >= 16) ? 1 : 0);
  else hipLaunchKernelGGL(patch_gather_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, *a, (blocks % 8 == 0 && blocks >= 16) ? 1 : 0);
  return tp::check_launch("tp_patch_gather");
}
