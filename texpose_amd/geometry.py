"""Camera / ray helpers of the hot path, mirroring the call signatures of the reference's
camera.py (:292-322, :345-350, :415-440) and tools/ray_sampler.py (:13-69) on top of the HIP kernels.

Only what Graph.render needs is here; the reference's Lie / quaternion / NDC / Procrustes utilities
are unused by the adapt_st_gan path and are out of scope.
"""
from __future__ import annotations

import math

import torch

from . import ops


# ----------------------------------------------------------------------------- rigid transforms
def make_pose(R=None, t=None):
    """[R|t] as a [...,3,4] tensor; either part may be omitted (identity / zero)."""
    if R is None and t is None:
        raise ValueError("make_pose needs R and/or t")
    if R is None:
        t = torch.as_tensor(t, dtype=torch.float32)
        R = torch.eye(3, device=t.device).expand(*t.shape[:-1], 3, 3)
    R = torch.as_tensor(R, dtype=torch.float32)
    t = torch.zeros(R.shape[:-1], device=R.device) if t is None else torch.as_tensor(t, dtype=torch.float32)
    if R.shape[-2:] != (3, 3) or R.shape[:-1] != t.shape:
        raise ValueError(f"bad pose parts {tuple(R.shape)} / {tuple(t.shape)}")
    return torch.cat([R, t.unsqueeze(-1)], dim=-1)


def invert_pose(p):
    """Inverse of a rigid [R|t]: [R^T | -R^T t] (what the reference's cam2world applies, camera.py:38-44)."""
    Rt = p[..., :3].transpose(-1, -2)
    return make_pose(Rt, (-Rt @ p[..., 3:]).squeeze(-1))


def compose_poses(*poses):
    """x -> p_n(...p_2(p_1(x)))."""
    out = poses[0]
    for nxt in poses[1:]:
        out = make_pose(nxt[..., :3] @ out[..., :3], (nxt[..., :3] @ out[..., 3:] + nxt[..., 3:]).squeeze(-1))
    return out


def rotation_distance(R1, R2, eps=1e-7):
    """Geodesic angle between rotation matrices; evaluate_full uses it to pick the light latent of the
    nearest training view (reference camera.py:345-350, model/nerf_adapt_st_gan.py:489-494)."""
    d = R1 @ R2.transpose(-2, -1)
    trace = d[..., 0, 0] + d[..., 1, 1] + d[..., 2, 2]             # (left to right, as the reference adds them: golden G18 is bit-exact)
    return torch.acos(((trace - 1) / 2).clamp(-1 + eps, 1 - eps))


# ----------------------------------------------------------------------------- rays
def get_center_and_ray(opt, pose, intr=None, H=None, W=None):
    """Every pixel-centre ray of an H x W image -> center, ray [B,HW,3] (reference camera.py:292-314)."""
    if opt.camera.model != "perspective":
        raise NotImplementedError(opt.camera.model)
    hh, ww = (opt.H, opt.W) if H is None and W is None else (H, W)
    every = torch.arange(hh * ww, device=pose.device).expand(len(pose), -1).contiguous()
    center, ray, _, _, _ = ops.raygen(intr, pose, H=hh, W=ww, ray_idx=every)
    return center, ray


def get_3D_points_from_depth(opt, center, ray, depth, multi_samples=False):
    """c + d z (reference camera.py:317-322).  Graph.render never calls this: the sample positions are
    formed inside the fused MLP kernel.  Kept as a one-line torch expression for callers that want points."""
    if multi_samples:
        center, ray = center.unsqueeze(2), ray.unsqueeze(2)
    return center + ray * depth


def aabb_ray_intersection(aabb_min, aabb_max, ray_o, ray_d):
    """Slab test (reference camera.py:415-433) -> t_near, t_far [B,HW], valid [B,HW] bool."""
    return ops.aabb_intersect(aabb_min, aabb_max, ray_o, ray_d)


def enlarge_diagonal(aabb_min, aabb_max, alpha=0.25):
    """Grow a box by alpha of its diagonal, half on each side (reference camera.py:436-440)."""
    pad = (aabb_max - aabb_min) * (alpha / 2)
    return aabb_min - pad, aabb_max + pad


# ----------------------------------------------------------------------------- detection crops
def crop_camera(cam_K, center, scale, res, raw_hw=(480, 640), stored_map_convention=False):
    """Camera of the ``res`` x ``res`` network input cut around a detection box, as the reference's data layer defines it:
    ``center`` = (row, col) and ``scale`` (side of the square in raw pixels) from ``get_2d_bbox`` (data/lm.py:161-180).
    Returns (intr [3,3], rect (x0, y0, x1, y1)): the intrinsics of ``preprocess_intrinsics`` applied at
    ``center + get_center_offset`` (data/lm.py:373-378,412-451) and the rectangle of crop pixels that have a source pixel
    inside the raw frame -- ``Crop_by_Pad`` (data/lm.py:455-495) resizes the clipped square keeping its aspect and pastes it
    centred into a zero canvas, so everything outside ``rect`` is padding.  Host arithmetic (a few scalars per image).

    ``stored_map_convention=True`` returns the camera under which the ON-THE-FLY slab test reproduces the reference's STORED
    bound maps: those are resampled with cv2's pixel-centre alignment from a map built with +0.5 ray centres
    (camera.py:301-302), which puts their samples 0.5 * (resize - 1) crop pixels off the rays the resized intrinsics
    define.  Use it for ``online_box_range`` only when bit-for-bit continuity with bounds loaded from
    the old npz files matters; the default evaluates the bounds on exactly the rays that are rendered."""
    ht, wd = raw_hw
    up0, le0 = int(center[0] - scale / 2. + 0.5), int(center[1] - scale / 2. + 0.5)
    upper, left = max(0, up0), max(0, le0)
    bottom, right = min(ht, up0 + int(scale)), min(wd, le0 + int(scale))
    h_off = -up0 / 2 if upper == 0 else (-(up0 + int(scale) - ht) / 2 if bottom == ht else 0)
    w_off = -le0 / 2 if left == 0 else (-(le0 + int(scale) - wd) / 2 if right == wd else 0)
    resize = res / scale
    K = torch.as_tensor(cam_K, dtype=torch.float32).clone()
    K[0, 0], K[1, 1] = K[0, 0] * resize, K[1, 1] * resize
    K[0, 2] = (K[0, 2] + 0.5) * resize - 0.5
    K[1, 2] = (K[1, 2] + 0.5) * resize - 0.5
    K[0, 2] = K[0, 2] - ((center[1] + w_off) * resize - res / 2)
    K[1, 2] = K[1, 2] - ((center[0] + h_off) * resize - res / 2)
    if stored_map_convention:
        K[0, 2] = K[0, 2] - (0.5 * resize - 0.5)
        K[1, 2] = K[1, 2] - (0.5 * resize - 0.5)
    crop_ht, crop_wd = float(bottom - upper), float(right - left)
    if crop_ht > crop_wd:
        rh, rw = res, int(res / crop_ht * crop_wd + 0.5)
    elif crop_ht < crop_wd:
        rw, rh = res, int(res / crop_wd * crop_ht + 0.5)
    else:
        rw = rh = int(res)
    r0, c0 = int(res / 2.0 - rh / 2.0 + 0.5), int(res / 2.0 - rw / 2.0 + 0.5)
    return K, (float(c0), float(r0), float(c0 + rw), float(r0 + rh))


def online_box_range(intr, pose, aabb_min, aabb_max, H, W, bg_range=(0.0, 30.0), valid_rect=None):
    """Per-pixel (z_near, z_far) [B,HW] of an object box seen by (intr, pose), computed on the fly by the fused ray-gen
    kernel -- the replacement of the per-frame ``pred_box_*.npz`` maps (compute_box.py:232-283 -> data/lm.py:316-350).
    ``valid_rect`` [B,4] (from ``crop_camera``) reproduces the padding of clipped crops; units follow ``pose`` / the box."""
    B = pose.shape[0]
    every = torch.arange(H * W, device=pose.device).expand(B, -1).contiguous()
    lo, hi = [float(v) for v in torch.as_tensor(aabb_min).flatten()], [float(v) for v in torch.as_tensor(aabb_max).flatten()]
    _, _, near, far, _ = ops.raygen(intr, pose, H=H, W=W, ray_idx=every, aabb=(lo, hi), bg_range=bg_range,
                                    valid_rect=valid_rect)
    return near, far


class RaySampler:
    """Train-mode ray source (reference tools/ray_sampler.py): continuous patch coordinates in [-1,1]
    -> rays, bilinear near/far bounds, bilinear image taps.  All three are static in the reference."""

    def __init__(self, opt=None, intrinsics=None):
        self.intrinsics = intrinsics

    @staticmethod
    def _size(opt, H, W):
        return (opt.H, opt.W) if H is None and W is None else (H, W)

    @staticmethod
    def get_rays(opt, intrinsics, coords, pose, H=None, W=None):
        hh, ww = RaySampler._size(opt, H, W)
        center, ray, _, _, _ = ops.raygen(intrinsics, pose, H=hh, W=ww, coords=coords)
        return center.view(*coords.shape[:3], 3), ray.view(*coords.shape[:3], 3)

    @staticmethod
    def get_bounds(opt, coords, z_near, z_far, H=None, W=None):
        hh, ww = RaySampler._size(opt, H, W)
        B = coords.shape[0]
        # the bounds lookup does not depend on the camera: run the fused kernel with an identity one
        k = torch.eye(3, device=coords.device).expand(B, 3, 3)
        p = torch.eye(3, 4, device=coords.device).expand(B, 3, 4)
        _, _, near, far, _ = ops.raygen(k, p, H=hh, W=ww, coords=coords, z_near=z_near, z_far=z_far)
        return near.view(coords.shape[:3]), far.view(coords.shape[:3])

    @staticmethod
    def get_image(opt, coords, image, H=None, W=None):
        if image.shape[1] != 3:
            raise NotImplementedError("get_image gathers 3-channel images")
        blank = image.new_zeros(image.shape[0], *image.shape[-2:])
        return ops.patch_gather(coords, image, image, image, image, blank, blank)[:, :3]


# ----------------------------------------------------------------------------- patch coordinates
class FlexPatchSampler:
    """Random-scale / random-shift p x p coordinate grids (reference tools/patch_sampler.py:64-114).

    Per image: s ~ U[lo, hi) with lo annealed as min(0.8, max(min_scale, hi * exp(-it * anneal))), then the
    [-1,1] lattice is scaled by s and shifted by U[-1,1) * (1 - s) per axis.  Returns (coords [B,p,p,2]
    in grid_sample (x, y) order, scales [B,1,1,1]).  Host-side torch plumbing: 3 B random numbers.
    """

    def __init__(self, random_shift=True, random_scale=True, min_scale=0.25, max_scale=1., scale_anneal=-1):
        self.random_shift, self.random_scale = random_shift, random_scale
        self.min_scale, self.max_scale, self.scale_anneal = min_scale, max_scale, scale_anneal
        self.iterations = 0
        self.scales_curr = (min_scale, max_scale)
        self.full_indices = False
        self.device_lo = None                # 0-dim device tensor while a hipGraph-captured step owns the sampler
        self.device_counter = None           # int64 [1] device word: draw inside the kernel, keyed by (seed, this counter)

    def _host_range(self):
        lo = self.min_scale
        if self.scale_anneal > 0:
            lo = min(0.8, max(lo, self.max_scale * math.exp(-self.iterations * self.scale_anneal)))
        return lo, self.max_scale

    def scale_range(self):
        if self.device_lo is not None:       # captured training step: the annealed bound lives in device memory and is
            return self.device_lo, self.max_scale      # refreshed OUTSIDE the graph by update_device_bound()
        return self._host_range()

    def update_device_bound(self):
        self.device_lo.fill_(self._host_range()[0])

    def __call__(self, nbatch, patch_size, device="cuda", u=None, defer=False):
        """``u`` ([3,B,1,1,1] uniforms: scale, x-shift, y-shift) replaces the internal draw in parity tests.  ``defer`` (captured training
        step, in-kernel draw only): nothing is launched; the ray-generation launch fills the returned tensors (ops.patch_coords)."""
        lo, hi = self.scales_curr = self.scale_range()
        if u is None and self.device_counter is not None:
            # captured training step: no torch.rand launch (and no generator-state fills before every replay); the kernel draws
            # from Philox(seed, step counter) -- the counter lives on the device and is advanced once per step
            from . import ops
            return ops.patch_coords(None, patch_size, lo, hi, self.random_scale, self.random_shift, nbatch=nbatch,
                                    seed=torch.initial_seed(), counter=self.device_counter, defer=defer)
        if u is None:
            u = torch.rand(3, nbatch, 1, 1, 1, device=device)
        if u.is_cuda:                                        # one launch (K13 tp_patch_coords), same fp32 operation order
            from . import ops
            return ops.patch_coords(u, patch_size, lo, hi, self.random_scale, self.random_shift)
        s = u[0] * (hi - lo) + lo if self.random_scale else torch.zeros(nbatch, 1, 1, 1, device=device) + lo
        lattice = torch.linspace(-1, 1, patch_size, device=device)
        xs = lattice.view(1, 1, patch_size, 1) * s          # varies along the patch width  -> grid x
        ys = lattice.view(1, patch_size, 1, 1) * s          # varies along the patch height -> grid y
        if self.random_shift:
            room = 1 - s
            xs = xs + (u[1] * 2.0 - 1.0) * room
            ys = ys + (u[2] * 2.0 - 1.0) * room
        coords = torch.cat([xs.expand(nbatch, patch_size, patch_size, 1), ys.expand(nbatch, patch_size, patch_size, 1)],
                           dim=-1)
        return coords.contiguous(), s.contiguous()
