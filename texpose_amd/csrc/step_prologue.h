// Device bodies shared by the launches in front of a training render: the patch sampler's coordinates (tp_patch_coords) and the per-image
// latent rows (tp_latent_rows_fwd) as kernels of their own (train_misc.hip), and as parts of the ray-generation launch of a captured
// training step (raygen.hip, tp_raygen_train): one launch instead of three in front of the MLP forward.  Same arithmetic either way.
#pragma once
#include "tp_common.h"

namespace tp_prologue {

// ---- patch coordinates (reference tools/patch_sampler.py:64-114): s = u0 * (hi - lo) + lo;  x = lattice_j * s + (u1 * 2 - 1) * (1 - s);
// y likewise with u2.  u == NULL: the three uniforms of image b are drawn here, Philox4x32-10 with key = seed and counter
// (b, c_lo, 'patc', c_hi), c = *counter (the step counter of a captured training step) -- words x, y, z -> scale, x shift, y shift.
struct Sampler {
  const float* u; const float* lattice; const float* lo_dev; const uint64_t* counter;
  float* coords; float* scales;
  uint64_t seed;
  float lo_host, span_host, hi;
  int B, p, random_scale, random_shift;
};

// element e = (image b, row i, column j) of the B x p x p grid -> (x, y); writes coords[e] and, from the image's first element, scales[b]
__device__ __forceinline__ void patch_coord(const Sampler& q, int e, float& x, float& y) {
  const int pp = q.p * q.p, b = e / pp, r = e - b * pp, i = r / q.p, j = r - i * q.p;
  const float lo = q.lo_dev ? *q.lo_dev : q.lo_host;
  const float span = q.lo_dev ? tp::sub_rn(q.hi, lo) : q.span_host;      // (a device-side bound is subtracted in fp32, like torch)
  float u0, u1, u2;
  if (q.u != nullptr) { u0 = q.u[b]; u1 = q.u[q.B + b]; u2 = q.u[2 * q.B + b]; }
  else {
    const uint64_t c = q.counter ? *q.counter : 0;
    const uint4 w = tp::philox4x32_10(make_uint4((uint32_t)b, (uint32_t)c, 0x70617463u, (uint32_t)(c >> 32)),
                                      make_uint2((uint32_t)q.seed, (uint32_t)(q.seed >> 32)));
    u0 = tp::u01(w.x); u1 = tp::u01(w.y); u2 = tp::u01(w.z);
  }
  const float s = q.random_scale ? tp::add_rn(tp::mul_rn(u0, span), lo) : tp::add_rn(0.f, lo);
  x = tp::mul_rn(q.lattice[j], s); y = tp::mul_rn(q.lattice[i], s);
  if (q.random_shift) {
    const float room = tp::sub_rn(1.f, s);
    x = tp::add_rn(x, tp::mul_rn(tp::sub_rn(tp::mul_rn(u1, 2.0f), 1.0f), room));
    y = tp::add_rn(y, tp::mul_rn(tp::sub_rn(tp::mul_rn(u2, 2.0f), 1.0f), room));
  }
  q.coords[2 * (size_t)e] = x;
  q.coords[2 * (size_t)e + 1] = y;
  if (r == 0) q.scales[b] = s;
}

// ---- per-image latent rows (model/nerf_adapt_st_gan.py:589-593: Embedding.weight[var.idx]) of BOTH tables; element e of B x (Ct + Cl)
struct Rows {
  const float* wt; const float* wl; const int64_t* idx;
  float* ot; float* ol; int64_t* idx_copy;
  int B, Ct, Cl;
};

__device__ __forceinline__ void latent_row_element(const Rows& q, int e) {
  const int C = q.Ct + q.Cl;
  if (e >= q.B * C) return;
  const int b = e / C, c = e - b * C;
  const int64_t r = q.idx[b];
  if (c == 0 && q.idx_copy != nullptr) q.idx_copy[b] = r;
  if (c < q.Ct) q.ot[b * q.Ct + c] = q.wt[r * q.Ct + c];
  else q.ol[b * q.Cl + (c - q.Ct)] = q.wl[r * q.Cl + (c - q.Ct)];
}

}  // namespace tp_prologue
