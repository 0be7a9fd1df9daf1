/*
 * texpose_amd.h -- C ABI of the MI355X (gfx950) ray-marching library.
 *
 * Drop-in boundary: the reference has no FFI; its boundary is the Python
 * class protocol  Graph.render / NeRF.composite / RaySampler.get_rays ...  selected by module
 * name (reference train.py:18-19, model/base.py:42-44).  texpose_amd/ mirrors that protocol in
 * Python and reaches the GPU only through the entry points below (ctypes, see
 * texpose_amd/_lib.py and INTEGRATION.md).  Every pointer is a DEVICE pointer unless noted,
 * every tensor is dense row-major fp32, indices are int64, `stream` is a hipStream_t.
 * Launchers only enqueue work on `stream`: no allocation, no synchronisation, graph-capturable.
 *
 * Return value: 0 on success, a hipError_t (>0) when the launch failed, <0 for invalid
 * arguments.  tp_last_error() returns a static, thread-local description.
 *
 * "ref:" comments cite the reference file:line each entry point replaces.
 */
#ifndef TEXPOSE_AMD_H
#define TEXPOSE_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* tp_stream_t; /* hipStream_t */

#define TP_ABI_VERSION 13

int tp_abi_version(void);
const char* tp_last_error(void);

/* ------------------------------------------------------------------------------------------
 * K1  fused ray generation + depth bounds + stratified samples
 * ref: tools/ray_sampler.py:24-69 (train rays/bounds), camera.py:292-314 +
 *      model/nerf_adapt_st_gan.py:573-579,702-710 (eval rays + row gather),
 *      camera.py:415-433 / compute_box.py:69-87,270-272 + data/lm.py:349-350 (AABB bounds),
 *      model/nerf_adapt_st_gan.py:683-700 (sample_depth).
 * ------------------------------------------------------------------------------------------ */
enum tp_pixel_mode {
  TP_PIX_COORDS = 0, /* train: coords [B,R,2] in [-1,1] (x,y); u,v = bilinear index-grid lookup */
  TP_PIX_INDEX = 1   /* eval : ray_idx [B,R] int64 row-major pixel index; u,v = col+.5,row+.5 */
};
enum tp_bounds_mode {
  TP_BOUNDS_MAP = 0,  /* z_near/z_far [B,H*W] maps: bilinear (coords) or gathered (index) */
  TP_BOUNDS_AABB = 1, /* in-kernel slab test against aabb_min/aabb_max, misses -> bg range */
  TP_BOUNDS_NONE = 2  /* rays only: near/far/depth outputs untouched */
};
enum tp_jitter_mode {
  TP_JITTER_MID = 0,    /* sample_stratified=false: 0.5 */
  TP_JITTER_GIVEN = 1,  /* rand [B,R,N] supplied by the caller (parity runs) */
  TP_JITTER_PHILOX = 2  /* in-kernel Philox4x32-10 keyed by (seed, offset) */
};
enum tp_depth_param {   /* options nerf.depth.param (model/nerf_adapt_st_gan.py:699) */
  TP_DEPTH_METRIC = 0,
  TP_DEPTH_INVERSE = 1  /* 1 / (sample + 1e-8) */
};

typedef struct tp_raygen_args {
  const float* intr;      /* [B,3,3] */
  const float* pose;      /* [B,3,4] object->camera [R|t] */
  const float* coords;    /* [B,R,2] (TP_PIX_COORDS) or NULL */
  const int64_t* ray_idx; /* [B,R]   (TP_PIX_INDEX)  or NULL */
  const float* z_near;    /* [B,H*W] (TP_BOUNDS_MAP) or NULL */
  const float* z_far;     /* [B,H*W] */
  const float* rand;      /* [B,R,N] (TP_JITTER_GIVEN) or NULL */
  float aabb_min[3];      /* TP_BOUNDS_AABB (host values) */
  float aabb_max[3];
  float bg_near, bg_far;  /* TP_BOUNDS_AABB fallback range */
  const float* valid_rect; /* TP_BOUNDS_AABB, optional: [B,4] device (x0,y0,x1,y1) in pixel coordinates of the H x W image;
                            * pixels whose centre (col+0.5,row+0.5 / bilinear coordinate) lies outside get the fallback
                            * range.  Reproduces the zero padding of the stored bound maps when a detection crop leaves
                            * the camera frame (data/lm.py:455-495 Crop_by_Pad, :349-350).  NULL = no rectangle. */
  uint64_t seed, offset;  /* TP_JITTER_PHILOX */
  const uint64_t* offset_dev; /* TP_JITTER_PHILOX, optional: a device word ADDED to offset (the step counter of a captured training
                               * step: the same launch draws different numbers at every replay) */
  int B, R, H, W, N;      /* N = samples per ray (0: no depth output) */
  int pixel_mode, bounds_mode, jitter_mode;
  int ndc;                /* non-zero: centre / ray re-expressed in normalised device coordinates, near plane z = 1 (camera.py:325-342
                           * convert_NDC; model/nerf_adapt_st_gan.py:581-583 `camera.ndc`).  Applied AFTER the bounds, which the reference
                           * takes from the metric rays. */
  int depth_param;        /* TP_DEPTH_METRIC, or TP_DEPTH_INVERSE: depth = 1 / (sample + 1e-8) (model/nerf_adapt_st_gan.py:699) */
  float* center;          /* [B,R,3] out */
  float* ray;             /* [B,R,3] out (camera-z component == 1 unless ndc) */
  float* near;            /* [B,R] out, may be NULL */
  float* far;             /* [B,R] out, may be NULL */
  float* depth;           /* [B,R,N] out, may be NULL */
} tp_raygen_args;

int tp_raygen(const tp_raygen_args* args /* host struct */, tp_stream_t stream);

/* The ray generation of a TRAINING step with the two small launches in front of it folded in (a captured iteration is a chain of
 * dependent launches: each one fewer is ~5 us of it).  `sampler` (or NULL): the launch draws the patch coordinates itself -- the
 * arguments of tp_patch_coords (K13; tools/patch_sampler.py:64-114), ray q = element q of the B x p x p grid, R == p * p,
 * args->pixel_mode TP_PIX_COORDS, args->coords ignored -- and writes them to sampler->coords / ->scales for the later consumers.
 * `rows` (or NULL): extra workgroups gather the per-image latent rows (the arguments of tp_latent_rows_fwd).  Same values as the
 * separate launches, bit for bit. */
typedef struct tp_patch_sampler_job {
  const float* u;          /* [3,B] uniforms, or NULL: drawn in the kernel from (seed, *counter) */
  int p;
  const float* lattice;    /* [p] linspace(-1, 1, p) */
  const float* lo_dev;     /* device word, or NULL: lo_host */
  float lo_host, span_host, hi;
  int random_scale, random_shift;
  uint64_t seed;
  const uint64_t* counter;
  float* coords;           /* [B,p,p,2] out */
  float* scales;           /* [B] out */
} tp_patch_sampler_job;
typedef struct tp_latent_rows_job {
  const float* w_trans; const float* w_light; const int64_t* idx;
  int B, C_trans, C_light;
  float* out_trans; float* out_light; int64_t* idx_copy /* or NULL */;
} tp_latent_rows_job;
int tp_raygen_train(const tp_raygen_args* args, const tp_patch_sampler_job* sampler, const tp_latent_rows_job* rows, tp_stream_t stream);

/* Standalone slab test (camera.py:415-433): o,d [n,3] -> t_near,t_far [n], valid [n] (uint8). */
int tp_aabb(const float* aabb_min3 /*host*/, const float* aabb_max3 /*host*/, const float* o, const float* d,
            int64_t n, float* t_near, float* t_far, uint8_t* valid, tp_stream_t stream);

/* Standalone Graph.sample_depth (model/nerf_adapt_st_gan.py:683-700): near,far [n] -> depth [n,N]. */
int tp_sample_depth(const float* near, const float* far, const float* rand /*[n,N] or NULL*/, int jitter_mode,
                    uint64_t seed, uint64_t offset, int64_t n, int N, int depth_param /* TP_DEPTH_* */, float* depth, tp_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * K2/K3  positional encoding + static/transient/light MLP
 * ref: layers/nerf_static_transient_light.py:76-145 (forward), :147-166 (forward_samples),
 *      :217-234 (positional_encoding), camera.py:317-322 (points from depth).
 * Architecture is the reference default (options/nerf_lm_adapt_gan.yaml:9-17,37-40): width 256,
 * 8 trunk layers with skip at 4, L_3D=10, L_view=4, 16 transient + 48 light latents.
 * ------------------------------------------------------------------------------------------ */

/* Pointers to the reference state-dict tensors, device memory, W [out,in]. */
typedef struct tp_mlp_weights {
  const float* feat_w[8];  const float* feat_b[8];   /* mlp_feat.{0..7}   */
  const float* rgb_w[4];   const float* rgb_b[4];    /* mlp_rgb.{0..3}    */
  const float* trans_w[4]; const float* trans_b[4];  /* mlp_trans.{0..3}  */
} tp_mlp_weights;

/* Size in bytes of the packed (MFMA-fragment-ordered, streaming-ordered) weight image. */
size_t tp_mlp_packed_bytes(void);
/* Which parts to (re)pack: the trunk is frozen, the heads change every optimiser step. */
enum { TP_PACK_TRUNK = 1, TP_PACK_HEADS = 2, TP_PACK_ALL = 3,
       TP_PACK_F16X3 = 4 /* OR-ed in: build the split-fp16 (hi+lo) stream of the TP_MLP_F16X3 forward; same size */,
       TP_PACK_RAYBIAS = 8 /* OR-ed in with TP_PACK_F16X3: the stream variant of tp_mlp_fwd_args.ray_bias (same buffer size) */ };
int tp_mlp_pack(const tp_mlp_weights* w /*host struct of device ptrs*/, int parts, void* packed, tp_stream_t stream);
/* Training (TP_MLP_F16X3): everything that is rebuilt from the HEAD weights after an optimiser step, in one launch: the head chunks
 * and head biases of the f16x3 forward stream `packed` (what tp_mlp_pack(TP_PACK_HEADS | TP_PACK_F16X3) writes) and, when packed_t is
 * not NULL, the transposed f16x3 image tp_mlp_bwd streams (what it builds itself when called with repack != 0). */
int tp_mlp_pack_heads_f16x3(const tp_mlp_weights* w, void* packed, void* packed_t, tp_stream_t stream);
/* Same image built on the host from HOST weight pointers (no GPU needed; used by the CPU tests). */
int tp_mlp_pack_host(const tp_mlp_weights* w_host, float* packed_host);

/* Bytes of scratch the forward needs for a given launch (per-workgroup spill of the trunk feature). */
size_t tp_mlp_workspace_bytes(int64_t n_samples);
/* Bytes of the optional activation record kept for the backward pass (7 x 256 floats / sample,
 * rounded up to whole 128-sample tiles). */
size_t tp_mlp_saved_bytes(int64_t n_samples);

typedef struct tp_mlp_fwd_args {
  const void* packed;      /* tp_mlp_pack output */
  /* input form A (forward_samples): center,ray [B,R,3], depth [B,R,N] */
  const float* center; const float* ray; const float* depth;
  /* input form B (NeRF.forward): points, ray_unit [B,R,N,3]; used when center == NULL */
  const float* points; const float* ray_unit;
  const float* lat_trans;  /* [B,16] */
  const float* lat_light;  /* [B,48] */
  int B, R, N;
  float* rgb;              /* [B,R,N,3,2] out (last dim: static, transient) */
  float* density;          /* [B,R,N,2]   out */
  float* uncert;           /* [B,R,N,1]   out */
  float* saved;            /* optional activations for tp_mlp_bwd (tp_mlp_saved_bytes) or NULL */
  void* workspace;         /* tp_mlp_workspace_bytes */
  int precision;           /* TP_MLP_FP32 (exact fp32 MFMA) or TP_MLP_F16X3 (packed with TP_PACK_F16X3) */
  int* status;             /* TP_MLP_F16X3: device word, bit 0 is set if an activation left the fp16 range */
  unsigned int* act_max;   /* TP_MLP_F16X3, optional (may be NULL): device word that receives, by atomic max, the fp32 bit
                              pattern of the largest hidden activation handed to the matrix cores in this call (post-ReLU,
                              its fp16 hi part: 11 significant bits).  The range guard fires at 6e4; this word says how far
                              below it a network runs.  Zero it before the calls to be covered. */
  float* ray_bias;         /* optional (NULL = off).  TP_MLP_F16X3 without `saved`, input form A, N % 128 == 0 (every 128-sample tile
                              lies inside one ray), `packed` built with TP_PACK_F16X3 | TP_PACK_RAYBIAS: device scratch of
                              tp_mlp_ray_bias_bytes(B, R).  The inputs of mlp_rgb.0 that are constant along a ray (view encoding 27,
                              light code 48 of its 334 columns; reference layers/nerf_static_transient_light.py:104-118) and the
                              transient code of mlp_trans.0 (:127-129) are then contracted once per ray / image in fp32 by a
                              pre-kernel and enter the layer as a per-ray bias instead of 96 input columns of every sample. */
  const float* density_noise; /* optional (NULL = off), [B*R*N]: added to the STATIC density's pre-activation before its softplus
                              (reference nerf.density_noise_reg, layers/nerf_static_transient_light.py:96-97, train mode: the caller
                              passes randn * density_noise_reg).  Not with `ray_bias`. */
} tp_mlp_fwd_args;
size_t tp_mlp_ray_bias_bytes(int B, int R);
/* TP_MLP_F16X3: every fp32 operand is split into hi + lo fp16 (22-bit significand) and hi*hi + hi*lo + lo*hi is
 * accumulated in fp32 on the f16 matrix cores (16x the fp32-MFMA rate / 3).  Measured error vs an fp64 oracle is
 * within 1.3x of plain fp32.  Requires |activation| < 6e4 (see `status`).  With `saved` it writes
 * the same fp32 activation record as TP_MLP_FP32 (the backward kernels do not depend on how the forward was computed). */
enum { TP_MLP_FP32 = 0, TP_MLP_F16X3 = 1 };

int tp_mlp_fwd(const tp_mlp_fwd_args* args, tp_stream_t stream);

/* Backward of the two trainable heads (autograd of layers/...light.py:102-137; the trunk is frozen,
 * :34,87-100).  Consumes the activation record of a forward run with `saved` != NULL on the same
 * (B,R,N).  When B*R*N is not a multiple of 128 the caller must zero the last tile's part of `saved`
 * BEFORE the forward (the weight-gradient GEMM contracts whole 32-sample groups).  At most 32 images
 * per call.  All gradient outputs are overwritten (not accumulated); results are deterministic
 * (fixed-order split-K reduction, no float atomics). */
size_t tp_mlp_packed_t_bytes(void);                 /* scratch for the transposed head-weight stream */
size_t tp_mlp_bwd_workspace_bytes(int64_t n_samples);
typedef struct tp_mlp_bwd_args {
  tp_mlp_weights weights;  /* rgb_w[0..3], trans_w[0..3] are read (device pointers) */
  void* packed_t;          /* tp_mlp_packed_t_bytes; rebuilt from `weights` when repack != 0 */
  int repack;
  const float* saved;      /* record written by tp_mlp_fwd */
  const float* rgb; const float* density; const float* uncert;        /* forward outputs */
  const float* g_rgb; const float* g_density; const float* g_uncert;  /* their cotangents */
  const float* lat_trans;  /* [B,16] */
  const float* lat_light;  /* [B,48] */
  int B, R, N;
  float* g_rgb_w[4];   float* g_rgb_b[4];     /* out: same shapes as mlp_rgb.{i}.weight / bias   */
  float* g_trans_w[4]; float* g_trans_b[4];   /* out: same shapes as mlp_trans.{i}.weight / bias */
  float* g_lat_trans;  /* [B,16] out */
  float* g_lat_light;  /* [B,48] out */
  void* workspace;         /* tp_mlp_bwd_workspace_bytes */
  int wgrad_precision;     /* arithmetic of the backward GEMMs (dgrad and wgrad).  TP_MLP_FP32: exact fp32 MFMA;
                              TP_MLP_F16X3: split-fp16 products with power-of-two scaling of the gradients (fp32-grade;
                              requires |activation| < 6e4, i.e. a record written by a TP_MLP_F16X3 forward whose status
                              word stayed clear; `packed_t` is then built in the transposed f16x3 format) */
  int dz_max_is_clear;     /* TP_MLP_F16X3: the scale word inside `workspace` is known to be zero -- true after any completed call
                              with the same workspace and B*R*N (the call's last kernel clears it) -- so the call does not
                              memset it.  0 = clear it first (always safe). */
  int wgrad_cus;           /* TP_MLP_F16X3: compute units the weight-gradient launch fills (one workgroup each; its split-K slice
                              counts follow).  0 = all.  A caller that runs other streams beside this call (the captured GAN
                              iteration: the discriminator step) passes 7/8 of the device so that their launches are not starved
                              for the length of the weight gradient; values are identical for every choice up to the order of
                              the fixed-order split-K sums. */
} tp_mlp_bwd_args;
int tp_mlp_bwd(const tp_mlp_bwd_args* args, tp_stream_t stream);

/* Standalone positional encoding (layers/...light.py:217-234): x [n,C] -> [n, 2*C*L]. */
int tp_posenc(const float* x, int64_t n, int C, int L, float* out, tp_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * K4  per-ray alpha composite
 * ref: layers/nerf_static_transient_light.py:168-212
 * ------------------------------------------------------------------------------------------ */
typedef struct tp_composite_args {
  const float* ray;      /* [n,3]     */
  const float* rgb;      /* [n,N,3,2] */
  const float* density;  /* [n,N,2]   */
  const float* depth;    /* [n,N]     */
  const float* uncert;   /* [n,N]     */
  int64_t n; int N; float min_uncert;
  float* out_ray;        /* [n,14]: rgb3 rgb_static3 rgb_transient3 depth opacity opacity_static opacity_transient uncert */
  float* alpha_static;   /* [n,N] or NULL */
  float* alpha_transient;/* [n,N] or NULL */
  float* prob;           /* [n,N] or NULL */
  float* rgb_ray;        /* [n,3] or NULL: compact copy of out_ray[:,0:3] (what the losses and the PatchGAN consume) */
  float* uncert_ray;     /* [n]   or NULL: compact copy of out_ray[:,13] */
} tp_composite_args;
int tp_composite_fwd(const tp_composite_args* args, tp_stream_t stream);

typedef struct tp_composite_bwd_args {
  tp_composite_args fwd;        /* same inputs as the forward (outputs ignored) */
  const float* g_out_ray;       /* [n,14] cotangent of out_ray, or NULL (zeros) when g_rgb_ray / g_uncert_ray carry it all */
  const float* g_alpha_static;  /* [n,N] or NULL */
  const float* g_alpha_transient;
  const float* g_prob;
  float* g_rgb;                 /* [n,N,3,2] out */
  float* g_density;             /* [n,N,2]   out */
  float* g_uncert;              /* [n,N]     out */
  const float* g_rgb_ray;       /* [n,3] or NULL: cotangent of rgb_ray, ADDED to g_out_ray[:,0:3] */
  const float* g_uncert_ray;    /* [n]   or NULL: cotangent of uncert_ray, ADDED to g_out_ray[:,13] */
  const float* g_rgb_ray2;      /* [n,3] or NULL: two more cotangents of rgb_ray (one per consumer of the colours), added likewise */
  const float* g_rgb_ray3;
  const float* g_density_add;   /* [n,N,2] or NULL: a second cotangent of the densities, ADDED to g_density */
} tp_composite_bwd_args;
int tp_composite_bwd(const tp_composite_bwd_args* args, tp_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * One-call evaluation render: tp_raygen -> tp_mlp_fwd -> tp_composite_fwd on one stream
 * ref: Graph.render with mode != 'train', model/nerf_adapt_st_gan.py:547-631 (latent rows chosen by the caller, :589-605)
 * ------------------------------------------------------------------------------------------ */
typedef struct tp_render_eval_args {
  tp_raygen_args raygen;   /* inputs, sizes and modes as for tp_raygen (N > 0); its output pointers are ignored */
  const void* packed;      /* tp_mlp_pack output for `precision` */
  const float* lat_trans;  /* [B,16] */
  const float* lat_light;  /* [B,48] */
  int precision;           /* TP_MLP_FP32 or TP_MLP_F16X3 */
  int* status;             /* as in tp_mlp_fwd_args (may be NULL) */
  float min_uncert;
  void* workspace;         /* tp_render_eval_workspace_bytes(B, R, N) */
  float* out_ray;          /* [B*R,14] out, layout of tp_composite_args.out_ray */
  float* alpha_static;     /* [B*R,N] out or NULL */
  float* alpha_transient;  /* [B*R,N] out or NULL */
  int packed_ray_bias;     /* non-zero: `packed` is the TP_PACK_F16X3 | TP_PACK_RAYBIAS stream (precision TP_MLP_F16X3, N % 128 == 0) and
                              the MLP runs in its ray-bias form (tp_mlp_fwd_args.ray_bias; the scratch is part of `workspace`) */
} tp_render_eval_args;
size_t tp_render_eval_workspace_bytes(int B, int R, int N);   /* (includes the ray-bias scratch) */
int tp_render_eval(const tp_render_eval_args* args, tp_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * K5  patch gather for the photometric / PatchGAN inputs
 * ref: model/nerf_adapt_st_gan.py:444-461,516-545,726-745 (8 grid_sample calls per step)
 * ------------------------------------------------------------------------------------------ */
typedef struct tp_patch_gather_args {
  const float* coords;     /* [B,P,2] (x,y) in [-1,1], P = p*p */
  const float* image;      /* [B,3,H,W] */
  const float* image_syn;  /* [B,3,H,W] */
  const float* nocs;       /* [B,3,H,W] */
  const float* normal;     /* [B,3,H,W] */
  const float* obj_mask;   /* [B,H,W]   (binarised > 0 in-kernel) */
  const float* mask_syn;   /* [B,H,W]   */
  int B, P, H, W;
  float* out;              /* [B,14,P]: image3 image_syn3 nocs3*mask_syn normal3*mask_syn mask mask_syn */
  /* optional (disc_rgb != NULL): the PatchGAN's input stacks of the same pixels in the same launch, what tp_disc_inputs forms from
   * `out` and the rendered colours disc_rgb [B,P,3]: disc_real / disc_fake [B, disc_geo ? 9 : 3, P] (model/nerf_adapt_st_gan.py:478-497) */
  const float* disc_rgb; float* disc_real; float* disc_fake; int disc_geo; int pad_;
} tp_patch_gather_args;
int tp_patch_gather(const tp_patch_gather_args* args, tp_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * K6  evaluation metrics after the render: squared-error and SSIM sums (PSNR / SSIM of evaluate_full)
 * ref: model/nerf_adapt_st_gan.py:340-362, external/pohsun_ssim/pytorch_ssim/__init__.py:7-37
 *      (the optional 480x640 resize of :346-350 is fused: bilinear align_corners=False / nearest)
 * ------------------------------------------------------------------------------------------ */
typedef struct tp_eval_metrics_args {
  const float* rgb_static; /* [B,h*w,3] the render (channels last, as Graph.render returns it) */
  const float* image;      /* [B,3,h,w] */
  const float* obj_mask;   /* [B,h,w]   multiplies the image (NOT binarised: the reference uses it as is) */
  int B, h, w;
  int out_h, out_w;        /* metric resolution; == (h, w) for no resize */
  void* workspace;         /* tp_eval_metrics_workspace_bytes(B, out_h, out_w) */
  double* out;             /* [B,2]: sum over 3*out_h*out_w of squared error, of the SSIM map */
} tp_eval_metrics_args;
int64_t tp_eval_metrics_workspace_bytes(int B, int out_h, int out_w);
int tp_eval_metrics(const tp_eval_metrics_args* args, tp_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * K7  spectral-norm weight normalisation for all convolutions of the PatchGAN at once
 * ref: torch.nn.utils.spectral_norm as applied in layers/discriminator.py:45-115 (one power iteration per forward in
 *      training mode: v <- normalize(W^T u), u <- normalize(W v), sigma = u.(W v), W_sn = W / sigma; u, v in place)
 * ------------------------------------------------------------------------------------------ */
#define TP_SN_MAX_WEIGHTS 8
#define TP_SN_MAX_SLABS 8        /* rows <= 512 */
#define TP_SN_MAX_SETS 3         /* power iterations tp_sn_fwd_sets runs in a row */
typedef struct tp_sn_weight {
  const float* weight;     /* [rows,cols] = weight_orig.view(out, -1); fwd only */
  float* u;                /* [rows]  updated in place when training */
  float* v;                /* [cols]  updated in place when training */
  float* weight_sn;        /* [rows,cols] fwd: out; bwd: in */
  float* sigma;            /* [1]     fwd: out; bwd: in */
  const float* grad_sn;    /* [rows,cols] bwd: gradient wrt weight_sn */
  float* grad;             /* [rows,cols] bwd: out, gradient wrt weight */
  float* work;             /* tp_sn_work_floats(rows, cols) floats of scratch */
  int rows, cols;
  float* u_out;            /* fwd, optional: copy of u [rows] as it stands AFTER this call's power iteration (the backward of
                              THIS forward needs it; later forwards of the iteration advance u / v in place) */
  float* v_out;            /* fwd, optional: copy of v [cols] likewise */
  int32_t accumulate;      /* bwd: grad += */
  /* bwd, optional: a SECOND normalised instance of the same weight in one optimiser step (the discriminator step's fake pass:
   * the power iteration advances between the passes) -- grad = f(grad_sn, weight_sn, u, v, sigma) + f(grad_sn2, ...) in the same
   * two launches; all five or none */
  const float* grad_sn2; const float* weight_sn2; const float* u2; const float* v2; const float* sigma2;
} tp_sn_weight;
int64_t tp_sn_work_floats(int rows, int cols);
/* three launches (two in eval mode): every workgroup of the consuming kernel normalises v / u for itself */
int tp_sn_fwd(const tp_sn_weight* weights, int n, int training, tp_stream_t stream);
/* n_sets training-mode forwards in a row (entry k * n + i = weight i of set k: own weight_sn / sigma / u_out / v_out; weight, u, v, work
 * shared): two launches per set and ONE normalisation launch for all sets; bit-identical to n_sets tp_sn_fwd calls. */
int tp_sn_fwd_sets(const tp_sn_weight* weights, int n, int n_sets, tp_stream_t stream);
int tp_sn_bwd(const tp_sn_weight* weights, int n, tp_stream_t stream);
/* tp_sn_bwd for a discriminator step that ENDS here (one rank, no gradient reduction behind it): the loss total + step gate ride in the
 * first launch (what tp_weighted_sum_flags does: total = sum_k terms[k][0] * weights[k] in ascending k, bad[word_finite] |= 1 if it is not
 * finite, snapshot[0..n_bad) = bad[...]), the RMSprop step of every weight (tp_rmsprop_step's arithmetic; withheld if a snapshot word is
 * set) in the second, on the gradient element the thread has just formed -- 2 launches instead of 4.  The gradients are still written.
 * ref: model/nerf_adapt_st_gan.py:129-171 (disc_trainstep: summarize_loss, backward, optim_disc.step()), model/base.py:145-157 */
typedef struct tp_sn_step_tail {
  const float* terms[4]; float weights[4]; int n_terms; int word_finite;
  float* total;                                  /* [1] */
  int32_t* bad; int32_t* snapshot; int n_bad; int pad_;
  float* param[TP_SN_MAX_WEIGHTS];               /* weight_orig of weight i (the tensor tp_sn_weight.grad belongs to) */
  float* square_avg[TP_SN_MAX_WEIGHTS];
  float* step[TP_SN_MAX_WEIGHTS];                /* optional 0-dim float32 counters, += 1 per applied step */
  const float* lr_dev; float lr_host, alpha, one_minus_alpha, eps;      /* (one_minus_alpha = (float)(1.0 - alpha) formed in double, like tp_rmsprop_step) */
} tp_sn_step_tail;
int tp_sn_bwd_step(const tp_sn_weight* weights, int n, const tp_sn_step_tail* tail, tp_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * K8  render-consuming loss terms of the generator step (render, uncert, trans_reg) and their gradients
 * ref: model/nerf_adapt_st_gan.py:712-776 (compute_loss, train_step == 'nerf', nerf.mask_obj)
 * ------------------------------------------------------------------------------------------ */
#define TP_NERF_LOSSES_MAX_BLOCKS 1024
typedef struct tp_nerf_losses_args {
  const float* rgb;        /* [B,P,3]   composite output */
  const float* uncert;     /* [B,P]     composite output (>= min_uncert) */
  const float* density;    /* [B,P,N,2] MLP output; [...,1] = sigma_transient */
  const float* gathered;   /* [B,14,P]  tp_patch_gather output: image = channels 0..2, object mask = channel 12 */
  int B, P, N;
  void* workspace;         /* 4 * TP_NERF_LOSSES_MAX_BLOCKS floats */
  double* sums;            /* [4]: sum m*se/u^2, sum m, sum log u^2, sum sigma_t  (fwd: out, bwd: in) */
  float* losses;           /* fwd, optional [3]: render = s0 / (s1 + 1e-5), uncert = 5 + s2 / (B P) / 2, trans_reg = s3 / (B P N)
                              in fp32 from the fp32-rounded sums, the reference's operation order */
  uint32_t* ticket;        /* fwd: ONE zero-filled device word owned by the calling stream (the arrival counter of the last-block
                              reduction; the launch leaves it zero).  Launches that may overlap -- different streams of one device --
                              need different words; launches on one stream may share one.  Unused by the backward. */
} tp_nerf_losses_args;
int tp_nerf_losses_fwd(const tp_nerf_losses_args* args, tp_stream_t stream);
/* g_render / g_unc / g_trans: the upstream gradients of the three terms, one device scalar each (NULL = 0) */
int tp_nerf_losses_bwd(const tp_nerf_losses_args* args, const float* g_render, const float* g_unc, const float* g_trans,
                       float* g_rgb, float* g_uncert, float* g_density, tp_stream_t stream);
/* the same launch + the generator step's loss total and step gate (tp_weighted_sum_flags' arguments and arithmetic) as a side job of its
 * first thread: one launch less on the render's backward chain (model/base.py:145-157 summarize_loss) */
int tp_nerf_losses_bwd_total(const tp_nerf_losses_args* args, const float* g_render, const float* g_unc, const float* g_trans,
                             float* g_rgb, float* g_uncert, float* g_density, const float* const* terms, const float* weights, int n,
                             float* out, const int32_t* mlp_status, int32_t* bad, int n_bad, int word_status, int word_finite,
                             int32_t* snapshot, uint64_t* step_counter, tp_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * K9  InstanceNorm2d (affine = False) + LeakyReLU of the PatchGAN ladder, one launch per derivative order
 * ref: layers/discriminator.py:94-115 (IN(c) + LeakyReLU(0.2) after each stride-2 SN-conv);
 *      the double backward serves the R1 penalty, model/nerf_adapt_st_gan.py:794-807 (compute_grad2).
 * x, y, xhat, gy, gx, ggx: [n_inst, hw] (n_inst = images x channels, hw = H*W), rstd: [n_inst].
 * ------------------------------------------------------------------------------------------ */
int tp_inorm_lrelu_fwd(const float* x, int64_t n_inst, int hw, float eps, float slope, float* y, float* xhat, float* rstd,
                       tp_stream_t stream);
/* addend (optional, [n_inst, hw]): a second cotangent of x that is added to the result (the R1 path's, K16) */
int tp_inorm_lrelu_bwd(const float* xhat, const float* rstd, const float* gy, int64_t n_inst, int hw, float slope,
                       const float* addend, float* gx, tp_stream_t stream);
/* cotangent ggx of gx -> gradients wrt gy and wrt x (through xhat and rstd) */
int tp_inorm_lrelu_bwd_bwd(const float* xhat, const float* rstd, const float* gy, const float* ggx, int64_t n_inst, int hw,
                           float slope, float* g_gy, float* g_x, tp_stream_t stream);

/* ---- pairs: TWO independent problems of one kernel in ONE launch (workgroups [0, n_a) work on the first, the rest on the second).  The
 * discriminator step (reference model/nerf_adapt_st_gan.py:129-171) passes the real and the fake patch stack through the same ladder and
 * tail one after the other; both passes' launches are latency-sized at these shapes (10-15 us for a few hundred kFLOP..MFLOP), so a pair
 * takes the time of one.  The two problems must not share an output, a split-K workspace, tile counters or a ticket. */
typedef struct tp_inorm_bwd_args {
  const float* xhat; const float* rstd; const float* gy;
  int64_t n_inst; int32_t hw; float slope;
  const float* addend;     /* or NULL */
  float* gx;
} tp_inorm_bwd_args;
int tp_inorm_lrelu_bwd_pair(const tp_inorm_bwd_args* a, const tp_inorm_bwd_args* b, tp_stream_t stream);


/* ------------------------------------------------------------------------------------------
 * K10  RMSprop step of the PatchGAN parameters in one launch (reference optim_disc.step(),
 * model/nerf_adapt_st_gan.py:168, torch.optim.RMSprop: alpha 0.99, eps 1e-8, no momentum / centring / weight decay).
 * lr_dev != NULL: the learning rate is read from device memory (a hipGraph-captured step); else lr_host.
 * ------------------------------------------------------------------------------------------ */
#define TP_RMSPROP_MAX_TENSORS 16
typedef struct tp_rmsprop_tensor {
  float* param;            /* [numel] updated in place */
  const float* grad;       /* [numel] */
  float* square_avg;       /* [numel] updated in place */
  int64_t numel;
  float* step;             /* optional 0-dim float device tensor (torch's state "step"): += 1 by the same launch when the step is applied */
} tp_rmsprop_tensor;
/* gate: n_gate int32 device words (or 0): if any is non-zero the launch changes nothing (a flagged step, see tp_step_flags) */
/* hyper-parameters as doubles: 1 - alpha is formed in double and then rounded, like torch's scalar arithmetic */
int tp_rmsprop_step(const tp_rmsprop_tensor* tensors /* host array */, int n, const float* lr_dev, double lr_host, double alpha,
                    double eps, const int32_t* gate, int n_gate, tp_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * K11  The stride-2 4x4 convolutions of the PatchGAN ladder (reference layers/discriminator.py:94-115:
 *      Conv2d(c_in, c_out, (4,4), (2,2), (1,1), bias=False) under spectral_norm, called through
 *      model/nerf_adapt_st_gan.py:129-171 and differentiated twice by compute_grad2, :794-807).  Three implicit-GEMM
 *      kernels on the fp32 matrix cores, closed under differentiation (a convolution is bilinear in input and weight):
 *        tp_conv4s2_fwd    out = y  [N,Co,H/2,W/2] = conv(x, w)
 *        tp_conv4s2_dgrad  out = gx [N,C,H,W]      = conv_transpose(gy, w)     (every element written)
 *        tp_conv4s2_wgrad  out = gw [Co,C,4,4]     = sum over images and positions of gy (x) window(x)
 *      Dense fp32 NCHW / OIHW tensors; H, W powers of two >= 8.  `workspace` holds split-K partial sums
 *      (tp_conv4s2_workspace floats, may be NULL when that is 0), `counters` one zero-initialised uint32 per output
 *      tile (n_counters of tp_conv4s2_workspace; the kernels leave them zero).  Fixed summation order: deterministic.
 * ------------------------------------------------------------------------------------------ */
typedef struct tp_conv4s2_args {
  const float* x;          /* [N,C,H,W]        fwd, wgrad */
  const float* w;          /* [Co,C,4,4]       fwd, dgrad */
  const float* gy;         /* [N,Co,H/2,W/2]   dgrad, wgrad */
  float* out;
  float* workspace;
  void* counters;
  int32_t N, C, H, W, Co;
  /* dgrad, optional (in_gx != NULL; H = W = 8): the InstanceNorm + LeakyReLU backward of the ladder stage in FRONT of this convolution
   * (tp_inorm_lrelu_bwd with gy = this data gradient: xhat [N,C,8,8], rstd [N*C], addend or NULL, gx = in_gx) inside the same launch,
   * bit-identical to the two launches; skip_out != 0: the data gradient itself is not written (out may then be NULL... it is not read) */
  int32_t skip_out;
  const float* in_xhat; const float* in_rstd; const float* in_addend; float* in_gx;
  float in_slope; int32_t pad_;
  /* tp_conv4s2_fwd_inorm[_pair], optional: the launch also copies x to x_copy (16-byte aligned, N C H W a multiple of 4) */
  float* x_copy;
} tp_conv4s2_args;
#define TP_CONV_FWD 0
#define TP_CONV_DGRAD 1
#define TP_CONV_WGRAD 2
/* floats of workspace the operation `op` (TP_CONV_*) needs for these sizes; *n_counters = number of uint32 counters */
int64_t tp_conv4s2_workspace(const tp_conv4s2_args* args, int op, int64_t* n_counters);
int tp_conv4s2_fwd(const tp_conv4s2_args* args, tp_stream_t stream);
/* The forward convolution with the InstanceNorm2d (affine = False) + LeakyReLU that follows it in the ladder in ONE launch, for 4x4 and
 * 8x8 output maps (reference layers/discriminator.py:94-115): out = y = lrelu(xhat), xhat [N,Co,H/2,W/2], rstd [N*Co] as
 * tp_inorm_lrelu_fwd returns them; the convolution's own output is not materialised.  Workspace / counters:
 * tp_conv4s2_fwd_inorm_workspace (-1: map size not covered -> two launches). */
int64_t tp_conv4s2_fwd_inorm_workspace(const tp_conv4s2_args* args, int64_t* n_counters);
int tp_conv4s2_fwd_inorm(const tp_conv4s2_args* args, float eps, float slope, float* xhat, float* rstd, tp_stream_t stream);
int tp_conv4s2_dgrad(const tp_conv4s2_args* args, tp_stream_t stream);
int tp_conv4s2_wgrad(const tp_conv4s2_args* args, tp_stream_t stream);
int tp_conv4s2_fwd_inorm_pair(const tp_conv4s2_args* a, float* xhat_a, float* rstd_a, const tp_conv4s2_args* b, float* xhat_b, float* rstd_b,
                              float eps, float slope, tp_stream_t stream);            /* same map size in both problems */
int tp_conv4s2_dgrad_pair(const tp_conv4s2_args* a, const tp_conv4s2_args* b, tp_stream_t stream);
int tp_conv4s2_wgrad_pair(const tp_conv4s2_args* a, const tp_conv4s2_args* b, tp_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * K12  3x3 / stride 1 / padding 1 convolution (+ bias, + ReLU) of the frozen perceptual-loss feature network
 *      (reference layers/perceptual_loss.py:8-45, torchvision VGG19 features[:15]; model/nerf_adapt_st_gan.py:758-766).
 *        tp_conv3s1_fwd    out [N,Co,H,W] = relu?(conv(in [N,C,H,W], w [Co,C,3,3]) + bias)
 *        tp_conv3s1_dgrad  out [N,C,H,W]  = conv_transpose(in [N,Co,H,W] * (mask > 0), w)    (mask = the forward output of a
 *                          ReLU layer, or NULL)
 *      H, W powers of two >= 4.  Workspace / counters as for K11 (tp_conv3s1_workspace, op TP_CONV_FWD / TP_CONV_DGRAD).
 * ------------------------------------------------------------------------------------------ */
typedef struct tp_conv3s1_args {
  const float* in;
  const float* w;
  const float* bias;       /* fwd: [Co] or NULL */
  const float* mask;       /* dgrad: [N,Co,H,W] or NULL */
  float* out;
  float* workspace;
  void* counters;
  int32_t N, C, H, W, Co;
  int32_t relu;            /* fwd: apply max(., 0) */
} tp_conv3s1_args;
int64_t tp_conv3s1_workspace(const tp_conv3s1_args* args, int op, int64_t* n_counters);
int tp_conv3s1_fwd(const tp_conv3s1_args* args, tp_stream_t stream);
int tp_conv3s1_dgrad(const tp_conv3s1_args* args, tp_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * K13  Small per-iteration pieces of the training step, one launch each (csrc/train_misc.hip)
 * ------------------------------------------------------------------------------------------ */
/* FlexPatchSampler.__call__ (reference tools/patch_sampler.py:80-114).  u [3,B] uniforms (scale, x shift,
 * y shift), lattice [p] = linspace(-1, 1, p); the annealed lower scale bound from device memory (lo_dev, a captured step)
 * or from the host (lo_host, span_host = (float)(hi - lo) formed in double like the reference's Python arithmetic).
 * coords [B,p,p,2] in grid_sample (x, y) order, scales [B].
 * u == NULL: the uniforms are drawn inside the kernel -- Philox4x32-10, key = seed, counter (b, c_lo, 'patc', c_hi) with c = *counter
 * (a device word: the step counter of a captured training step; NULL = 0), words x, y, z -> scale, x shift, y shift of image b. */
int tp_patch_coords(const float* u, int B, int p, const float* lattice, const float* lo_dev, float lo_host, float span_host,
                    float hi, int random_scale, int random_shift, uint64_t seed, const uint64_t* counter, float* coords, float* scales,
                    tp_stream_t stream);
/* mean binary_cross_entropy_with_logits(x [n], target) for a constant target (compute_gan_loss 'standard',
 * model/nerf_adapt_st_gan.py:809-823): out [1]; backward gx [n] = g[0] (sigmoid(x) - target) / n. */
int tp_bce_logits_fwd(const float* x, int n, float target, float* out, tp_stream_t stream);
int tp_bce_logits_bwd(const float* x, int n, float target, const float* g, float* gx, tp_stream_t stream);
/* The four image batches of the feature loss (model/nerf_adapt_st_gan.py:758-766), ImageNet-normalised
 * (layers/perceptual_loss.py:19-20): out [4B,3,P] = [rgb | rgb m + image (1-m) | image m + image_syn pad | image],
 * pad = (mask_syn == 1 and m == 0); backward: g_rgb [B,P,3] from g_out [4B,3,P] (only the first 2B images count). */
typedef struct tp_feat_inputs_args {
  const float* rgb;        /* [B,P,3] rendered colours */
  const float* gathered;   /* [B,n_channels,P] tp_patch_gather output */
  int32_t B, P, n_channels;
  int32_t c_image, c_image_syn, c_mask, c_mask_syn;   /* first channel of each in `gathered` */
  float mean[3], std[3];
} tp_feat_inputs_args;
int tp_feat_inputs_fwd(const tp_feat_inputs_args* args, float* out, tp_stream_t stream);
int tp_feat_inputs_bwd(const tp_feat_inputs_args* args, const float* g_out, float* g_rgb, tp_stream_t stream);
/* K18 -- the whole feature-loss chain of the generator step in ONE call (17 launches): the four image stacks (tp_feat_inputs_fwd), the
 * frozen feature network F = torchvision VGG19 features[:15] (reference layers/perceptual_loss.py:8-45: 3x3 convolutions 3-64, 64-64,
 * max pool, 64-128, 128-128, max pool, 128-256, 256-256, 256-256, ReLU after each but the last), the two-pair loss
 * model/nerf_adapt_st_gan.py:758-766   loss = mse(F(fake1), F(real1)) + w2 mse(F(fake2), F(real2))   (targets detached)
 * and its gradient wrt the rendered colours: g_rgb = scale * d loss / d rgb.  Pools, ReLU derivatives and un-pooling ride in the
 * convolutions' epilogues; only the 2B fake images are differentiated.  16 x 16 patches (the reference's patch_size).
 * loss[3] = {l1 + w2 l2, l1, l2}.  workspace / counters: sizes from tp_feat_chain_workspace; the counters must be zero before the first
 * call and are left zero; both belong to ONE stream (calls that may overlap on different streams need their own). */
#define TP_FEAT_CHAIN_LAYERS 7
typedef struct tp_feat_chain_args {
  const float* rgb;        /* [B,P,3] rendered colours, P = H * W */
  const float* gathered;   /* [B,n_channels,P] tp_patch_gather output */
  int32_t B, H, W, n_channels;
  int32_t c_image, c_image_syn, c_mask, c_mask_syn;   /* first channel of each in `gathered` */
  float mean[3], std[3];                               /* ImageNet normalisation constants */
  const float* packed;                                 /* the seven [Co,C,3,3] weights as tp_feat_chain_pack wrote them */
  const float* bias[TP_FEAT_CHAIN_LAYERS];             /* [Co] */
  float w2;                /* weight of the second pair (the reference: 5) */
  float scale;             /* cotangent of the loss (the caller's loss weight 10^w; 1 for the plain gradient) */
  float* loss;             /* [3] */
  float* g_rgb;            /* [B,P,3] */
  float* workspace; int64_t workspace_floats;
  int32_t* counters; int64_t n_counters;
} tp_feat_chain_args;
int64_t tp_feat_chain_workspace(int32_t B, int32_t H, int32_t W, int64_t* n_counters);   /* floats; -1: shape not covered */
/* The frozen weights in the kernels' operand order, once per set of weights: packed [tp_feat_chain_packed_floats()] = for every layer
 * [C][9][Co] (forward) followed by, for every layer, [Co][9, taps flipped][C] (data gradient) -- a half-wavefront's 32 produced channels
 * are 128 consecutive bytes.  w: host array of the seven [Co,C,3,3] device tensors. */
int64_t tp_feat_chain_packed_floats(void);
int tp_feat_chain_pack(const float* const* w, float* packed, tp_stream_t stream);
int tp_feat_chain(const tp_feat_chain_args* args, tp_stream_t stream);
/* The discriminator step's inputs (model/nerf_adapt_st_gan.py:478-497, no gradient): real [B,nc,P] = image m + rgb pad,
 * fake [B,nc,P] = rgb, nc = 3 or (geo) 9 with the masked nocs / normal channels 6..11 of `gathered` [B,14,P] appended. */
/* Step gate of a captured training step (the reference asserts every weighted loss term finite on the host,
 * model/base.py:153-154): bad[word_status] |= mlp_status[0] & 1 (mlp_status may be NULL), bad[word_finite] |= !isfinite(total[0]),
 * then snapshot[0..n_bad) = bad[0..n_bad) -- the words the optimiser launch of this step reads as its gate. */
int tp_step_flags(const int32_t* mlp_status, const float* total, int32_t* bad, int n_bad, int word_status, int word_finite,
                  int32_t* snapshot, tp_stream_t stream);
/* torch.optim.Adam step (reference optim_nerf, model/nerf_adapt_st_gan.py:62-68,125; no weight decay, no amsgrad) for up to
 * TP_ADAM_MAX_TENSORS tensors in one launch; `step` = 0-dim float tensor holding the steps taken so far (the block that finishes last
 * launch adds 1 to each afterwards); learning rate from device memory (lr_dev) or the host; gate as for tp_rmsprop_step.
 * ticket: ONE zero-filled device word owned by the calling stream (arrival counter of the block that advances the step counters;
 * the launch leaves it zero); launches that may overlap on different streams of a device need different words. */
#define TP_ADAM_MAX_TENSORS 32
typedef struct tp_adam_tensor {
  float* param;
  const float* grad;
  float* exp_avg;
  float* exp_avg_sq;
  float* step;
  int64_t numel;
} tp_adam_tensor;
int tp_adam_step(const tp_adam_tensor* tensors /* host array */, int n, const float* lr_dev, double lr_host, double beta1, double beta2,
                 double eps, const int32_t* gate, int n_gate, uint32_t* ticket, tp_stream_t stream);
/* The per-iteration host -> device state of a replayed training step (reference Model.train_iteration's move_to_device of the
 * batch, model/nerf_adapt_st_gan.py:212; patch_sampler.iterations :185; discriminator.progress :182) in ONE launch:
 * n_copies contiguous device-to-device copies (16-byte aligned; the batch into the captured step's static inputs),
 * n_scalars host values written to device floats, and n_words int32 gate words copied out to device-visible pinned host memory
 * (words_dst may be NULL). */
#define TP_STEP_INPUTS_MAX_COPIES 24
#define TP_STEP_INPUTS_MAX_SCALARS 8
typedef struct tp_step_copy { void* dst; const void* src; int64_t bytes; } tp_step_copy;
int tp_step_inputs(const tp_step_copy* copies /* host array */, int n_copies, float* const* scalar_dst /* host array */,
                   const float* scalar_val /* host array */, int n_scalars, const int32_t* words_src, int32_t* words_dst, int n_words,
                   tp_stream_t stream);
/* Data parallel (no reference counterpart: options.py:112 asserts a single GPU; BASELINE.json config C4): the gradients of ONE
 * optimiser step into the flat fp32 buffer that one RCCL all-reduce sums -- flat[concatenation of the n tensors] = scale * grads[k]
 * (grads[k] NULL: zeros; scale = 1 / world, so the sum is the average) -- and the step-gate words into the buffer's tail as 0 / 1 floats:
 * tail[j] = (words[j] != 0 || tail[j] != 0) for j < n_words (words may be NULL).  The tail is STICKY: after the all-reduce it is
 * non-zero on every rank if any rank set it, in this or an earlier step, and read as int32 words it is the gate tp_adam_step /
 * tp_rmsprop_step take.  One launch. */
#define TP_GRAD_PACK_MAX_TENSORS 32
int tp_grad_pack(const float* const* grads /* host array */, const int64_t* numel /* host array */, int n, float* flat, float scale,
                 const int32_t* words, int n_words, float* tail, tp_stream_t stream);
/* Diagnostic (no reference counterpart): number of nodes (kernel launches, fills, copies) recorded so far in the hipGraph that `stream`
 * is capturing into, -1 when it is not capturing (the launch counts of a captured training iteration). */
int64_t tp_capture_node_count(tp_stream_t stream);
/* Diagnostic (no reference counterpart): the device's 100 MHz constant clock written to *slot by a one-thread launch in stream order. */
int tp_stamp(uint64_t* slot, tp_stream_t stream);
/* Diagnostic (no reference counterpart): one sleeping wave samples the shader clock against the 100 MHz constant clock over `windows`
 * consecutive windows of window_us microseconds: out[2w] = shader cycles, out[2w+1] = 100-MHz ticks of window w (out: 2 * windows device
 * words).  Launched on a side stream in front of a kernel, it reports the clock the chip holds under that kernel's load. */
int tp_clock_probe(uint64_t* out, int windows, int64_t window_us, tp_stream_t stream);
int tp_disc_inputs(const float* rgb, const float* gathered, int B, int P, int geo, float* real, float* fake, tp_stream_t stream);
/* Cotangent of the fake stack wrt the rendered colours (the nerf step back-propagates D(fake) into the render):
 * g_rgb [B,P,3] = g_fake [B,nc,P] channels 0..2, transposed. */
int tp_fake_patch_bwd(const float* g_fake, int B, int P, int nc, float* g_rgb, tp_stream_t stream);
/* Feature loss of the generator step (model/nerf_adapt_st_gan.py:762-766; layers/perceptual_loss.py:39-45): feat [4n] = network
 * features of [fake1 | fake2 | real1 | real2] (n floats each); out3 = {l1 + w2 l2, l1, l2} with l_i = mean((fake_i - real_i)^2);
 * backward g_feat [4n] for the cotangent g[0] of out3[0] (targets detached: their quarters are zero). */
int tp_feat_pair_loss_fwd(const float* feat, int64_t n, float w2, float* out3, tp_stream_t stream);
int tp_feat_pair_loss_bwd(const float* feat, int64_t n, float w2, const float* g, float* g_feat, tp_stream_t stream);
/* R1 penalty value (model/nerf_adapt_st_gan.py:794-807 and the mean over the batch its caller takes, :149): out[0] = sum(g^2) / B
 * for g [B,m] (n = B m floats); backward out [n] = 2 g cot[0] / B. */
int tp_sumsq_mean_fwd(const float* g, int64_t n, int B, float* out, tp_stream_t stream);
int tp_sumsq_mean_bwd(const float* g, int64_t n, int B, const float* cot, float* out, tp_stream_t stream);
/* K16 (the discriminator step as an explicit schedule): value and weighted gradient of the R1 penalty in one launch,
 * out[0] = sum(g^2) / B, out[1] = w out[0] (what the reference logs), out_g [n] = 2 w g / B (w = the term's loss weight, a host
 * constant);
 * and both GAN-loss terms of the discriminator step (model/nerf_adapt_st_gan.py:139-160) with their weighted cotangents:
 * out2 = {bce(d_real, 1), bce(d_fake, 0)}, g_real [n] = w_real (sigmoid(d_real) - 1) / n, g_fake [n] = w_fake sigmoid(d_fake) / n. */
int tp_sumsq_mean_fwd_bwd(const float* g, int64_t n, int B, float w, float* out, float* out_g, tp_stream_t stream);
/* MaxPool2d(2, 2) of the feature network (layers/perceptual_loss.py:8-18): x [n, H, W] (n = images x channels; H, W even) ->
 * y [n, H/2, W/2], arg [n, H/2, W/2] = position 0..3 of the maximum inside its window (the first one on ties, NaN wins: torch's
 * rule); backward gx [n, H, W] from gy and arg, every element written (no zero fill). */
int tp_maxpool2_fwd(const float* x, int64_t n, int H, int W, float* y, uint8_t* arg, tp_stream_t stream);
int tp_maxpool2_bwd(const float* gy, const uint8_t* arg, int64_t n, int H, int W, float* gx, tp_stream_t stream);
int tp_gan_disc_losses(const float* d_real, const float* d_fake, int n, float w_real, float w_fake, float* out2, float* g_real,
                       float* g_fake, tp_stream_t stream);
/* Rows idx[b] of the two latent tables (model/nerf_adapt_st_gan.py:589-593) in one launch, and the dense table gradients
 * gw [n_rows,C] = sum over b with idx[b] == r of g[b] (ascending b; every element written: no zero-fill needed). */
/* out[0] = sum_k weights[k] * terms[k][0] (ascending k) for up to 16 scalar device tensors; `terms` and `weights` are HOST
 * arrays (the weighted loss total of model/base.py:145-157, for logging and the step gate). */
int tp_weighted_sum(const float* const* terms, const float* weights, int n, float* out, tp_stream_t stream);
/* the same followed by tp_step_flags on the result, in one launch (the captured training step: loss total + step gate) */
int tp_weighted_sum_flags(const float* const* terms, const float* weights, int n, float* out, const int32_t* mlp_status, int32_t* bad,
                          int n_bad, int word_status, int word_finite, int32_t* snapshot, uint64_t* step_counter /* optional: += 1 */,
                          tp_stream_t stream);
/* idx_copy (optional, [B]): the launch also copies idx there -- the backward of a captured step reads that private copy, so that the
 * caller's idx buffer may be refilled for the next iteration while this one's backward is still to run */
int tp_latent_rows_fwd(const float* w_trans, const float* w_light, const int64_t* idx, int B, int C_trans, int C_light, float* out_trans,
                       float* out_light, int64_t* idx_copy, tp_stream_t stream);
int tp_latent_rows_bwd(const float* g_trans, const float* g_light, const int64_t* idx, int B, int n_rows, int C_trans, int C_light,
                       float* gw_trans, float* gw_light, tp_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * K14  Scale-conditioned head of the PatchGAN (reference layers/discriminator.py:30-40,112-115):
 *      a = [z, sin(s 2^l pi), cos(s 2^l pi) (l < L), s];  out = W3 lrelu(W2 lrelu(W1 lrelu(a)))   (1x1 SN-convs, no bias)
 *      one launch each for the forward, the backward (gz, gW1..3) and the R1 double backward (cotangent c_gz of gz ->
 *      d/d g_out and d/d W1..3; model/nerf_adapt_st_gan.py:794-807).  t0 [B,Cin] / t1 / t2 [B,H] are the saved activations
 *      (Cin = C + 2L + 1), e1 / e2 [B,H] the backward's intermediates the double backward re-uses.
 * ------------------------------------------------------------------------------------------ */
typedef struct tp_disc_head_args {
  const float* z;          /* [B,C]   fwd */
  const float* scale;      /* [B]     fwd */
  const float* W1;         /* [H,Cin] */
  const float* W2;         /* [H,H] */
  const float* W3;         /* [H] */
  const float* g_out;      /* [B]     bwd, bwd_bwd */
  const float* c_gz;       /* [B,C]   bwd_bwd */
  float* t0; float* t1; float* t2;   /* fwd: written; bwd, bwd_bwd: read */
  float* e1; float* e2;              /* bwd: written; bwd_bwd: read */
  float* out;              /* fwd: [B];  bwd: gz [B,C];  bwd_bwd: d/d g_out [B] */
  float* gW1; float* gW2; float* gW3;   /* bwd, bwd_bwd: [H,Cin], [H,H], [H]; bwd: all three NULL = data gradient only */
  int32_t B, C, L, H;
  float slope;
  int32_t accumulate_gw;   /* bwd: gW1..3 += (after the double backward wrote its share there) */
} tp_disc_head_args;
int tp_disc_head_fwd(const tp_disc_head_args* args, tp_stream_t stream);
int tp_disc_head_bwd(const tp_disc_head_args* args, tp_stream_t stream);
int tp_disc_head_bwd_bwd(const tp_disc_head_args* args, tp_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * K15  y = x W^T for a handful of rows: the PatchGAN's last ladder convolution covers its whole map (reference
 *      layers/discriminator.py:110-111), i.e. [B, 8192] x [8192, 64] with B = 4 .. 32.  Forward and weight gradient
 *      (gW = gy^T x) and data gradient gx = gy W (kernel for M <= 16, a library GEMM beyond).  x [M,K], w [N,K], y [M,N].
 * ------------------------------------------------------------------------------------------ */
int tp_skinny_linear_fwd(const float* x, const float* w, float* y, int M, int N, int K, tp_stream_t stream);
int tp_skinny_linear_wgrad(const float* gy, const float* x, float* gw, int M, int N, int K, tp_stream_t stream);
/* gx [M,K] = gy [M,N] w [N,K] for M <= 16 rows (larger M: a library GEMM on the host side) */
int tp_skinny_linear_dgrad(const float* gy, const float* w, float* gx, int M, int N, int K, tp_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * K17  the TAIL of the PatchGAN = K15's full-map convolution + K14's scale-conditioned head in ONE launch per derivative order
 *      (reference layers/discriminator.py:30-40,110-115; the R1 double backward of model/nerf_adapt_st_gan.py:794-807).
 *      Up to TP_DISC_TAIL_MAX_ROWS rows per pass; N = C = ndf, Cin = N + 2 L + 1, K a multiple of 4.
 *      fwd:      a [M,K], W0 [N,K], scale [M], W1..3 -> out [M], t0 [M,Cin], t1, t2 [M,H]
 *      bwd:      g_out [M], t0..2, W0..3 -> (all optional) gz [M,N], e1, e2 [M,H], c_a [M,K] = gz W0, gW0 [N,K] = sum_rows gz (x) a
 *                (+ sum over M2 extra rows gy2 [M2,N] (x) a2 [M2,K]: the R1 path's pair of the same weight), gW1..3 (all or none;
 *                accumulate_gw adds to what is there)
 *      bwd_bwd:  a = c [M,K] (cotangent of the first pass' c_a), g_out, t0..2, e1, e2, W0..3 -> gW1..3 (the double backward's
 *                share), out = d/d g_out [M] (optional)
 *      workspace: tp_disc_tail_workspace_bytes(N) bytes; ticket: one zero-filled device word owned by the calling stream (fwd, bwd_bwd).
 * ------------------------------------------------------------------------------------------ */
#define TP_DISC_TAIL_MAX_ROWS 16
typedef struct tp_disc_tail_args {
  const float* a;          /* fwd: ladder output [M,K]; bwd: the same (for gW0; may be NULL without gW0); bwd_bwd: c [M,K] */
  const float* W0;         /* [N,K] */
  const float* scale;      /* [M]   fwd */
  const float* W1; const float* W2; const float* W3;   /* [H,Cin], [H,H], [H] */
  const float* g_out;      /* [M]   bwd, bwd_bwd */
  float* t0; float* t1; float* t2;   /* fwd: written; bwd, bwd_bwd: read */
  float* e1; float* e2;              /* bwd: written (optional); bwd_bwd: read */
  float* out;              /* fwd: [M]; bwd_bwd: d/d g_out [M] (optional) */
  float* gz;               /* bwd: [M,N] (optional) */
  float* c_a;              /* bwd: [M,K] (optional) */
  float* gW0;              /* bwd: [N,K] (optional) */
  const float* gy2;        /* bwd: [M2,N] extra cotangent rows of gW0 */
  const float* a2;         /* bwd: [M2,K] their inputs */
  float* gW1; float* gW2; float* gW3;
  /* bwd, optional: the InstanceNorm2d + LeakyReLU backward of the ladder's LAST stage (K9 tp_inorm_lrelu_bwd) applied to c_a inside the
   * launch: c_z [M,K] = rstd P(c_a s(xhat)) (+ in_addend); an instance = in_P consecutive columns (in_P | 64: the 4x4 map's 16),
   * xhat [M,K], rstd [M K / in_P].  c_a may then be NULL (only c_z is written). */
  const float* in_xhat; const float* in_rstd; const float* in_addend; float* c_z; int32_t in_P;
  void* workspace;         /* fwd, bwd_bwd */
  uint32_t* ticket;        /* fwd, bwd_bwd */
  int32_t M, M2, K, N, L, H;
  float slope;
  int32_t accumulate_gw;
} tp_disc_tail_args;
size_t tp_disc_tail_workspace_bytes(int N);
int tp_disc_tail_fwd(const tp_disc_tail_args* args, tp_stream_t stream);
int tp_disc_tail_bwd(const tp_disc_tail_args* args, tp_stream_t stream);
int tp_disc_tail_bwd_bwd(const tp_disc_tail_args* args, tp_stream_t stream);
int tp_disc_tail_fwd_pair(const tp_disc_tail_args* a, const tp_disc_tail_args* b, tp_stream_t stream);
int tp_disc_tail_bwd_pair(const tp_disc_tail_args* a, const tp_disc_tail_args* b, tp_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* TEXPOSE_AMD_H */
