// K1: fused ray generation + per-ray depth bounds + stratified depth samples (gfx950).
//
// Replaces (reference file:line):
//   tools/ray_sampler.py:40-69    train rays from continuous patch coordinates
//   tools/ray_sampler.py:24-37    bilinear lookup of the z_near / z_far maps
//   camera.py:292-314 + model/nerf_adapt_st_gan.py:573-579,702-710   eval rays: the reference
//       materialises all H*W rays for every 2048-ray chunk and gathers; here only the requested
//       pixels are generated, straight from ray_idx
//   camera.py:415-433, compute_box.py:69-87,270-272, data/lm.py:349-350   AABB slab bounds
//   model/nerf_adapt_st_gan.py:683-700   stratified sample_depth
//
// HBM-bound streaming kernel: algorithmic traffic is 8 B/ray read (+84 B per image of camera
// constants, served from L2) and 24 + 4N B/ray written.  One 256-thread workgroup owns 256
// consecutive rays: phase 1 is one thread per ray (camera algebra, bounds), phase 2 has the whole
// workgroup write the 256*N depth samples of the tile as contiguous 16-byte stores, with the
// per-ray (near, far-near) pair staged in LDS.
#include "tp_common.h"
#include "step_prologue.h"

namespace {

constexpr int kTile = 256;

struct Cam {
  float kinv[9];
  float rt[9];    // R^T
  float tinv[3];  // -R^T t
};

__device__ __forceinline__ void load_cam(const float* __restrict__ intr, const float* __restrict__ pose, int b, Cam& c) {
  const float* K = intr + 9 * b;
  const float a = K[0], bb = K[1], cc = K[2], d = K[3], e = K[4], f = K[5], g = K[6], h = K[7], i = K[8];
  const float A = e * i - f * h, Bc = -(d * i - f * g), C = d * h - e * g;
  const float det = a * A + bb * Bc + cc * C;
  const float r = 1.0f / det;
  c.kinv[0] = A * r;  c.kinv[1] = -(bb * i - cc * h) * r;  c.kinv[2] = (bb * f - cc * e) * r;
  c.kinv[3] = Bc * r; c.kinv[4] = (a * i - cc * g) * r;    c.kinv[5] = -(a * f - cc * d) * r;
  c.kinv[6] = C * r;  c.kinv[7] = -(a * h - bb * g) * r;   c.kinv[8] = (a * e - bb * d) * r;
  const float* P = pose + 12 * b;
#pragma unroll
  for (int r_ = 0; r_ < 3; ++r_)
#pragma unroll
    for (int c_ = 0; c_ < 3; ++c_) c.rt[r_ * 3 + c_] = P[c_ * 4 + r_];
  const float t0 = P[3], t1 = P[7], t2 = P[11];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    float acc = tp::mul_rn(-c.rt[j * 3 + 0], t0);
    acc = tp::fma_rn(-c.rt[j * 3 + 1], t1, acc);
    acc = tp::fma_rn(-c.rt[j * 3 + 2], t2, acc);
    c.tinv[j] = acc;
  }
}

// grid_sample(bilinear, align_corners=True, zeros) of one channel, accumulated the way torch's
// CPU kernel does (one multiply, three fused multiply-adds in nw,ne,sw,se order).
struct Bilin {
  int x0, y0;
  float nw, ne, sw, se;
  bool ok00, ok10, ok01, ok11;
};

__device__ __forceinline__ Bilin bilin_setup(float x, float y, int H, int W) {
  Bilin b;
  const float ix = tp::mul_rn(tp::add_rn(x, 1.0f), (float)(W - 1) * 0.5f);
  const float iy = tp::mul_rn(tp::add_rn(y, 1.0f), (float)(H - 1) * 0.5f);
  const float fx = floorf(ix), fy = floorf(iy);
  const float w = tp::sub_rn(ix, fx), e = tp::sub_rn(1.0f, w);
  const float n = tp::sub_rn(iy, fy), s = tp::sub_rn(1.0f, n);
  b.nw = tp::mul_rn(e, s); b.ne = tp::mul_rn(w, s); b.sw = tp::mul_rn(e, n); b.se = tp::mul_rn(w, n);
  // clamp before the int conversion so that wild coordinates cannot overflow
  const float cx = fminf(fmaxf(fx, -2.0f), (float)W + 1.0f), cy = fminf(fmaxf(fy, -2.0f), (float)H + 1.0f);
  b.x0 = (int)cx; b.y0 = (int)cy;
  const bool xin0 = b.x0 >= 0 && b.x0 <= W - 1, xin1 = b.x0 + 1 >= 0 && b.x0 + 1 <= W - 1;
  const bool yin0 = b.y0 >= 0 && b.y0 <= H - 1, yin1 = b.y0 + 1 >= 0 && b.y0 + 1 <= H - 1;
  b.ok00 = xin0 && yin0; b.ok10 = xin1 && yin0; b.ok01 = xin0 && yin1; b.ok11 = xin1 && yin1;
  return b;
}

__device__ __forceinline__ float bilin_apply(const Bilin& b, float v00, float v10, float v01, float v11) {
  float r = tp::mul_rn(b.ok00 ? v00 : 0.0f, b.nw);
  r = tp::fma_rn(b.ok10 ? v10 : 0.0f, b.ne, r);
  r = tp::fma_rn(b.ok01 ? v01 : 0.0f, b.sw, r);
  r = tp::fma_rn(b.ok11 ? v11 : 0.0f, b.se, r);
  return r;
}

__device__ __forceinline__ float bilin_map(const Bilin& b, const float* __restrict__ m, int W) {
  const float v00 = b.ok00 ? m[b.y0 * W + b.x0] : 0.0f;
  const float v10 = b.ok10 ? m[b.y0 * W + b.x0 + 1] : 0.0f;
  const float v01 = b.ok01 ? m[(b.y0 + 1) * W + b.x0] : 0.0f;
  const float v11 = b.ok11 ? m[(b.y0 + 1) * W + b.x0 + 1] : 0.0f;
  return bilin_apply(b, v00, v10, v01, v11);
}

using tp::philox4x32_10;
using tp::u01;

// ((rand + i) / N) * (far - near) + near, each op rounded as in the reference expression.  POW2: N is a power of two (64, 128, 256:
// every BASELINE configuration), so the correctly-rounded quotient IS the product with 2^-k -- exact, and a third of the instructions
// of an IEEE division (the kernel was as much bound by them as by its stores: 61 us for 167 MB at 480x640x128).
template <bool POW2>
__device__ __forceinline__ float strat(float r, int i, float fN, float invN, float span, float near) {
  const float t = tp::add_rn(r, (float)i);
  const float q = POW2 ? tp::mul_rn(t, invN) : tp::div_rn(t, fN);
  return tp::add_rn(tp::mul_rn(q, span), near);
}
// nerf.depth.param = inverse (reference :699): 1 / (sample + 1e-8)
__device__ __forceinline__ float depth_of(float z, bool inverse) { return inverse ? tp::div_rn(1.0f, tp::add_rn(z, 1e-8f)) : z; }

// camera.convert_NDC (camera.py:325-342), near plane z = 1, every operation rounded as the reference expression rounds it:
//   o' = o + (1 - o_z) / d_z * d;   centre = (sx o'_x / o'_z, sy o'_y / o'_z, 1 - 2 / o'_z),  sx = K00 / K02, sy = K11 / K12
//   ray = (sx (d_x / d_z - o'_x / o'_z), sy (d_y / d_z - o'_y / o'_z), 2 / o'_z)
__device__ __forceinline__ void to_ndc(const float* __restrict__ K, float* o, float* d) {
  const float t = tp::div_rn(tp::sub_rn(1.0f, o[2]), d[2]);
  float c[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) c[j] = tp::add_rn(o[j], tp::mul_rn(t, d[j]));
  const float sx = tp::div_rn(K[0], K[2]), sy = tp::div_rn(K[4], K[5]);
  const float cxz = tp::div_rn(c[0], c[2]), cyz = tp::div_rn(c[1], c[2]), two_z = tp::div_rn(2.0f, c[2]);
  const float rx = tp::sub_rn(tp::div_rn(d[0], d[2]), cxz), ry = tp::sub_rn(tp::div_rn(d[1], d[2]), cyz);
  o[0] = tp::mul_rn(sx, cxz); o[1] = tp::mul_rn(sy, cyz); o[2] = tp::sub_rn(1.0f, two_z);
  d[0] = tp::mul_rn(sx, rx);  d[1] = tp::mul_rn(sy, ry);  d[2] = two_z;
}

struct Args {
  const float* intr; const float* pose; const float* coords; const int64_t* ray_idx;
  const float* z_near; const float* z_far; const float* rnd;
  float amin[3], amax[3], bg_near, bg_far;
  const float* valid_rect;
  uint64_t seed, offset;
  const uint64_t* offset_dev;
  int B, R, H, W, N, pixel_mode, bounds_mode, jitter_mode, ndc, inverse;
  float* center; float* ray; float* near; float* far; float* depth;
  // tp_raygen_train: the launch draws its own patch coordinates (sampler_on) and carries the latent rows as extra workgroups behind the
  // ray_blocks of the ray generation (rows.B > 0)
  tp_prologue::Sampler sampler; tp_prologue::Rows rows;
  int sampler_on, ray_blocks;
};

__device__ __forceinline__ void slab(const float* amin, const float* amax, const float* o, const float* d,
                                     float& tn, float& tf, bool& valid) {
  tn = -INFINITY; tf = INFINITY;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float inv = tp::div_rn(1.0f, d[a]);
    const float ta = tp::mul_rn(tp::sub_rn(amin[a], o[a]), inv);
    const float tb = tp::mul_rn(tp::sub_rn(amax[a], o[a]), inv);
    // torch.minimum/maximum propagate NaN (0*inf when the origin sits on a slab plane of a parallel ray)
    const float lo = (ta != ta || tb != tb) ? NAN : fminf(ta, tb);
    const float hi = (ta != ta || tb != tb) ? NAN : fmaxf(ta, tb);
    tn = (tn != tn || lo != lo) ? NAN : fmaxf(tn, lo);
    tf = (tf != tf || hi != hi) ? NAN : fminf(tf, hi);
  }
  valid = (tf > 0.0f) && (tf > tn);
}

// TILE rays per workgroup of kTile threads: 256 for image-sized launches (one ray per thread, then 256 N depths written by the
// workgroup), 32 for patch-sized ones (a training step has 1,024 rays: 4 workgroups would each walk 16 K depths -- and, with the
// in-kernel Philox draw, 16 x 10 rounds per thread -- on the step's critical path; 32 workgroups take an eighth of that).
template <int TILE, bool POW2>
__global__ __launch_bounds__(kTile) void raygen_kernel(Args a) {
  __shared__ float s_near[TILE];
  __shared__ float s_span[TILE];
  if ((int)blockIdx.x >= a.ray_blocks) {                      // (uniform per workgroup: the latent rows of a training launch)
    tp_prologue::latent_row_element(a.rows, ((int)blockIdx.x - a.ray_blocks) * kTile + (int)threadIdx.x);
    return;
  }
  const int64_t total = (int64_t)a.B * a.R;
  const int64_t tile0 = (int64_t)blockIdx.x * TILE;
  const int64_t q = tile0 + threadIdx.x;
  float near = 0.0f, far = 0.0f;
  if ((int)threadIdx.x < TILE && q < total) {
    const int b = (int)(q / a.R);
    Cam cam;
    load_cam(a.intr, a.pose, b, cam);
    float u, v;
    Bilin bl;
    int64_t pix = 0;
    if (a.pixel_mode == TP_PIX_COORDS) {
      float x, y;
      if (a.sampler_on) tp_prologue::patch_coord(a.sampler, (int)q, x, y);          // (ray q IS element q of the B x p x p grid)
      else { x = a.coords[2 * q]; y = a.coords[2 * q + 1]; }
      bl = bilin_setup(x, y, a.H, a.W);
      u = bilin_apply(bl, (float)bl.x0, (float)(bl.x0 + 1), (float)bl.x0, (float)(bl.x0 + 1));
      v = bilin_apply(bl, (float)bl.y0, (float)bl.y0, (float)(bl.y0 + 1), (float)(bl.y0 + 1));
    } else {
      pix = a.ray_idx[q];
      const int row = (int)(pix / a.W), col = (int)(pix - (int64_t)row * a.W);
      u = (float)col + 0.5f;
      v = (float)row + 0.5f;
    }
    // g = K^-1 [u,v,1];  world = R^T g + tinv;  ray = world - tinv  (camera.py:266-277,308-314)
    float g[3], o[3], d[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      float acc = tp::mul_rn(u, cam.kinv[j * 3 + 0]);
      acc = tp::fma_rn(v, cam.kinv[j * 3 + 1], acc);
      g[j] = tp::add_rn(acc, cam.kinv[j * 3 + 2]);
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      float acc = tp::mul_rn(g[0], cam.rt[j * 3 + 0]);
      acc = tp::fma_rn(g[1], cam.rt[j * 3 + 1], acc);
      acc = tp::fma_rn(g[2], cam.rt[j * 3 + 2], acc);
      const float world = tp::add_rn(acc, cam.tinv[j]);
      o[j] = cam.tinv[j];
      d[j] = tp::sub_rn(world, cam.tinv[j]);
    }
    if (!a.ndc) {
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        a.center[3 * q + j] = o[j];
        a.ray[3 * q + j] = d[j];
      }
    }
    if (a.bounds_mode == TP_BOUNDS_MAP) {
      const float* zn = a.z_near + (int64_t)b * a.H * a.W;
      const float* zf = a.z_far + (int64_t)b * a.H * a.W;
      if (a.pixel_mode == TP_PIX_COORDS) {
        near = bilin_map(bl, zn, a.W);
        far = bilin_map(bl, zf, a.W);
      } else {
        near = zn[pix];
        far = zf[pix];
      }
    } else if (a.bounds_mode == TP_BOUNDS_AABB) {
      float tn, tf; bool ok;
      slab(a.amin, a.amax, o, d, tn, tf, ok);
      tn = ok ? tn : 0.0f;
      tf = ok ? tf : 0.0f;
      near = tn > 0.0f ? tn : a.bg_near;
      far = tf > 0.0f ? tf : a.bg_far;
      if (a.valid_rect != nullptr) {
        // crop pixels without a source pixel in the camera frame: the stored maps are zero-padded there -> fallback range
        const float* vr = a.valid_rect + 4 * b;
        if (!(u >= vr[0] && u < vr[2] && v >= vr[1] && v < vr[3])) { near = a.bg_near; far = a.bg_far; }
      }
    }
    if (a.bounds_mode != TP_BOUNDS_NONE) {
      if (a.near) a.near[q] = near;
      if (a.far) a.far[q] = far;
    }
    if (a.ndc) {                           // (behind the bounds: the reference's ranges come from the metric rays)
      to_ndc(a.intr + 9 * b, o, d);
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        a.center[3 * q + j] = o[j];
        a.ray[3 * q + j] = d[j];
      }
    }
  }
  if (a.depth == nullptr || a.N <= 0 || a.bounds_mode == TP_BOUNDS_NONE) return;
  if ((int)threadIdx.x < TILE) {
    s_near[threadIdx.x] = near;
    s_span[threadIdx.x] = tp::sub_rn(far, near);
  }
  __syncthreads();
  const int64_t rays_here = (total - tile0) < TILE ? (total - tile0) : TILE;
  const int64_t n_el = rays_here * a.N;
  const int64_t e0 = tile0 * a.N;  // first global element of this tile
  const float fN = (float)a.N, invN = 1.0f / fN;
  const uint2 key = make_uint2((uint32_t)a.seed, (uint32_t)(a.seed >> 32));
  const uint64_t off = a.offset + (a.offset_dev ? *a.offset_dev : 0);       // (a captured step: the step counter on the device)
  if ((a.N & 3) == 0) {
    for (int64_t e = (int64_t)threadIdx.x * 4; e < n_el; e += kTile * 4) {
      const int r = (int)(e / a.N), i = (int)(e - (int64_t)r * a.N);
      float4 rr = make_float4(0.5f, 0.5f, 0.5f, 0.5f);
      if (a.jitter_mode == TP_JITTER_GIVEN) {
        rr = *reinterpret_cast<const float4*>(a.rnd + e0 + e);
      } else if (a.jitter_mode == TP_JITTER_PHILOX) {
        const uint64_t cnt = (uint64_t)(e0 + e) >> 2;
        const uint4 w = philox4x32_10(make_uint4((uint32_t)cnt, (uint32_t)off, (uint32_t)(cnt >> 32), (uint32_t)(off >> 32)), key);
        rr = make_float4(u01(w.x), u01(w.y), u01(w.z), u01(w.w));
      }
      const float nr = s_near[r], sp = s_span[r];
      float4 z;
      z.x = strat<POW2>(rr.x, i + 0, fN, invN, sp, nr);
      z.y = strat<POW2>(rr.y, i + 1, fN, invN, sp, nr);
      z.z = strat<POW2>(rr.z, i + 2, fN, invN, sp, nr);
      z.w = strat<POW2>(rr.w, i + 3, fN, invN, sp, nr);
      if (a.inverse) { z.x = depth_of(z.x, true); z.y = depth_of(z.y, true); z.z = depth_of(z.z, true); z.w = depth_of(z.w, true); }
      *reinterpret_cast<float4*>(a.depth + e0 + e) = z;
    }
  } else {
    for (int64_t e = threadIdx.x; e < n_el; e += kTile) {
      const int r = (int)(e / a.N), i = (int)(e - (int64_t)r * a.N);
      float rr = 0.5f;
      if (a.jitter_mode == TP_JITTER_GIVEN) {
        rr = a.rnd[e0 + e];
      } else if (a.jitter_mode == TP_JITTER_PHILOX) {
        const uint64_t ge = (uint64_t)(e0 + e), cnt = ge >> 2;
        const uint4 w = philox4x32_10(make_uint4((uint32_t)cnt, (uint32_t)off, (uint32_t)(cnt >> 32), (uint32_t)(off >> 32)), key);
        const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
        rr = u01(ws[ge & 3]);
      }
      a.depth[e0 + e] = depth_of(strat<POW2>(rr, i, fN, invN, s_span[r], s_near[r]), a.inverse != 0);
    }
  }
}

__global__ void aabb_kernel(float3 lo, float3 hi, const float* __restrict__ o, const float* __restrict__ d, int64_t n,
                            float* __restrict__ t_near, float* __restrict__ t_far, uint8_t* __restrict__ valid) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float amin[3] = {lo.x, lo.y, lo.z}, amax[3] = {hi.x, hi.y, hi.z};
  const float oo[3] = {o[3 * i], o[3 * i + 1], o[3 * i + 2]}, dd[3] = {d[3 * i], d[3 * i + 1], d[3 * i + 2]};
  float tn, tf; bool ok;
  slab(amin, amax, oo, dd, tn, tf, ok);
  t_near[i] = tn; t_far[i] = tf; valid[i] = ok ? 1 : 0;
}

__global__ void sample_depth_kernel(const float* __restrict__ near, const float* __restrict__ far,
                                    const float* __restrict__ rnd, int jitter, uint64_t seed, uint64_t offset,
                                    int64_t n, int N, int inverse, float* __restrict__ depth) {
  const int64_t total = n * N;
  const uint2 key = make_uint2((uint32_t)seed, (uint32_t)(seed >> 32));
  const float fN = (float)N;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = e / N;
    const int i = (int)(e - r * N);
    float rr = 0.5f;
    if (jitter == TP_JITTER_GIVEN) rr = rnd[e];
    else if (jitter == TP_JITTER_PHILOX) {
      const uint64_t cnt = (uint64_t)e >> 2;
      const uint4 w = philox4x32_10(make_uint4((uint32_t)cnt, (uint32_t)offset, (uint32_t)(cnt >> 32),
                                               (uint32_t)(offset >> 32)), key);
      const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
      rr = u01(ws[e & 3]);
    }
    const float nr = near[r];
    depth[e] = depth_of(strat<false>(rr, i, fN, 0.0f, tp::sub_rn(far[r], nr), nr), inverse != 0);
  }
}

}  // namespace

extern "C" int tp_raygen(const tp_raygen_args* p, tp_stream_t stream) { return tp_raygen_train(p, nullptr, nullptr, stream); }

extern "C" int tp_raygen_train(const tp_raygen_args* p, const tp_patch_sampler_job* sj, const tp_latent_rows_job* rj, tp_stream_t stream) {
  TP_REQUIRE(p != nullptr, "null args");
  TP_REQUIRE(p->B > 0 && p->R > 0 && p->H > 0 && p->W > 0, "bad sizes");
  TP_REQUIRE(p->intr && p->pose && p->center && p->ray, "null camera / output pointer");
  TP_REQUIRE(sj == nullptr || (p->pixel_mode == TP_PIX_COORDS && sj->lattice && sj->coords && sj->scales && sj->p > 0 && sj->p * sj->p == p->R),
             "sampler job: TP_PIX_COORDS, R == p * p, lattice / coords / scales expected");
  TP_REQUIRE(rj == nullptr || (rj->w_trans && rj->w_light && rj->idx && rj->out_trans && rj->out_light && rj->B > 0 && rj->C_trans > 0 && rj->C_light > 0),
             "latent-rows job: bad arguments");
  TP_REQUIRE(p->pixel_mode == TP_PIX_COORDS ? (p->coords != nullptr || sj != nullptr) : p->ray_idx != nullptr, "missing pixel source");
  TP_REQUIRE(p->bounds_mode != TP_BOUNDS_MAP || (p->z_near && p->z_far), "missing z_near/z_far maps");
  TP_REQUIRE(p->jitter_mode != TP_JITTER_GIVEN || p->depth == nullptr || p->rand != nullptr, "missing rand tensor");
  Args a;
  a.intr = p->intr; a.pose = p->pose; a.coords = p->coords; a.ray_idx = p->ray_idx;
  a.z_near = p->z_near; a.z_far = p->z_far; a.rnd = p->rand;
  for (int i = 0; i < 3; ++i) { a.amin[i] = p->aabb_min[i]; a.amax[i] = p->aabb_max[i]; }
  a.bg_near = p->bg_near; a.bg_far = p->bg_far; a.valid_rect = p->valid_rect; a.seed = p->seed; a.offset = p->offset; a.offset_dev = p->offset_dev;
  a.B = p->B; a.R = p->R; a.H = p->H; a.W = p->W; a.N = p->N;
  a.pixel_mode = p->pixel_mode; a.bounds_mode = p->bounds_mode; a.jitter_mode = p->jitter_mode;
  TP_REQUIRE(p->depth_param == TP_DEPTH_METRIC || p->depth_param == TP_DEPTH_INVERSE, "unknown depth_param");
  a.ndc = p->ndc != 0; a.inverse = p->depth_param == TP_DEPTH_INVERSE;
  a.center = p->center; a.ray = p->ray; a.near = p->near; a.far = p->far; a.depth = p->depth;
  const int64_t total = (int64_t)p->B * p->R;
  const int tile = total <= 16384 ? 32 : kTile;
  const int64_t ray_blocks = (total + tile - 1) / tile;
  a.sampler = tp_prologue::Sampler{}; a.rows = tp_prologue::Rows{}; a.sampler_on = 0; a.ray_blocks = (int)ray_blocks;
  int64_t blocks = ray_blocks;
  if (sj != nullptr) {
    tp_prologue::Sampler& q = a.sampler;
    q.u = sj->u; q.lattice = sj->lattice; q.lo_dev = sj->lo_dev; q.counter = sj->counter; q.coords = sj->coords; q.scales = sj->scales;
    q.seed = sj->seed; q.lo_host = sj->lo_host; q.span_host = sj->span_host; q.hi = sj->hi; q.B = p->B; q.p = sj->p;
    q.random_scale = sj->random_scale; q.random_shift = sj->random_shift;
    a.sampler_on = 1;
  }
  if (rj != nullptr) {
    tp_prologue::Rows& q = a.rows;
    q.wt = rj->w_trans; q.wl = rj->w_light; q.idx = rj->idx; q.ot = rj->out_trans; q.ol = rj->out_light; q.idx_copy = rj->idx_copy;
    q.B = rj->B; q.Ct = rj->C_trans; q.Cl = rj->C_light;
    blocks += ((int64_t)rj->B * (rj->C_trans + rj->C_light) + kTile - 1) / kTile;
  }
  TP_REQUIRE(blocks < (1ll << 31), "too many rays for one launch");
  const bool pow2 = p->N > 0 && (p->N & (p->N - 1)) == 0;          // (x / 2^k == x * 2^-k exactly: no IEEE division in the depth loop)
  const dim3 grid((unsigned)blocks), block(kTile);
  if (tile == 32) {
    if (pow2) hipLaunchKernelGGL((raygen_kernel<32, true>), grid, block, 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((raygen_kernel<32, false>), grid, block, 0, (hipStream_t)stream, a);
  } else {
    if (pow2) hipLaunchKernelGGL((raygen_kernel<kTile, true>), grid, block, 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((raygen_kernel<kTile, false>), grid, block, 0, (hipStream_t)stream, a);
  }
  return tp::check_launch("tp_raygen");
}

extern "C" int tp_aabb(const float* lo, const float* hi, const float* o, const float* d, int64_t n, float* t_near,
                       float* t_far, uint8_t* valid, tp_stream_t stream) {
  TP_REQUIRE(lo && hi && o && d && t_near && t_far && valid, "null pointer");
  if (n == 0) return 0;
  const int64_t blocks = (n + 255) / 256;
  hipLaunchKernelGGL(aabb_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                     make_float3(lo[0], lo[1], lo[2]), make_float3(hi[0], hi[1], hi[2]), o, d, n, t_near, t_far, valid);
  return tp::check_launch("tp_aabb");
}

extern "C" int tp_sample_depth(const float* near, const float* far, const float* rnd, int jitter, uint64_t seed,
                               uint64_t offset, int64_t n, int N, int depth_param, float* depth, tp_stream_t stream) {
  TP_REQUIRE(near && far && depth && N > 0, "bad arguments");
  TP_REQUIRE(depth_param == TP_DEPTH_METRIC || depth_param == TP_DEPTH_INVERSE, "unknown depth_param");
  TP_REQUIRE(jitter != TP_JITTER_GIVEN || rnd != nullptr, "missing rand tensor");
  if (n == 0) return 0;
  int64_t blocks = (n * N + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(sample_depth_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, near, far, rnd,
                     jitter, seed, offset, n, N, depth_param == TP_DEPTH_INVERSE, depth);
  return tp::check_launch("tp_sample_depth");
}
