"""CPU oracle for the TexPose ray-marching hot path.

TEST INFRASTRUCTURE ONLY.  This module is a from-scratch CPU restatement (plain
PyTorch fp32 / numpy) of the reference algorithm for SURVEY.md section 8 rows
a1..a17.  Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` may import it; the product path (``texpose_amd``) never
does and fails loudly when the HIP library is missing.

Parity pin: every function below is checked in ``tests/test_oracle_golden.py``
against golden vectors captured from the real reference imported in the build
container (``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``).

Citations are ``file:line`` into the upstream reference tree (HanzhiC/TexPose).
All tensors are fp32 unless stated, ray/sample indices are int64.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch

Tensor = torch.Tensor

# ----------------------------------------------------------------------------
# a1  FlexPatchSampler.__call__            tools/patch_sampler.py:80-114
# ----------------------------------------------------------------------------

def patch_min_scale(iterations: int, scale_anneal: float = 0.0002,
                    min_scale: float = 0.25, max_scale: float = 1.0) -> float:
    """Annealed lower bound of the patch scale (tools/patch_sampler.py:86-91)."""
    if scale_anneal > 0:
        lo = max(min_scale, max_scale * math.exp(-iterations * scale_anneal))
        lo = min(0.8, lo)
    else:
        lo = min_scale
    return lo


def patch_coords(patch_size: int, u_scale: Tensor, u_hoff: Tensor, u_woff: Tensor,
                 lo: float, hi: float = 1.0) -> Tuple[Tensor, Tensor]:
    """Continuous patch coordinates in [-1,1] (tools/patch_sampler.py:80-114).

    ``u_*`` are the three uniform [0,1) draws of shape [B] the reference takes
    from ``torch.rand`` (scale, h-offset, w-offset, in that order).
    Returns coords [B,p,p,2] (last dim = (x, y) in grid_sample order) and
    scales [B,1,1,1].
    """
    B = u_scale.shape[0]
    lin = torch.linspace(-1, 1, patch_size)
    # reference: w,h = meshgrid(lin, lin) (ij) -> w varies along dim0, h along dim1
    w = lin[:, None].expand(patch_size, patch_size)
    h = lin[None, :].expand(patch_size, patch_size)
    scales = (u_scale * (hi - lo) + lo).view(B, 1, 1, 1)
    h = h[None, ..., None] * scales
    w = w[None, ..., None] * scales
    max_off = 1 - scales
    h = h + (u_hoff.view(B, 1, 1, 1) * 2.0 - 1.0) * max_off
    w = w + (u_woff.view(B, 1, 1, 1) * 2.0 - 1.0) * max_off
    coords = torch.cat([h, w], dim=-1)
    return coords.contiguous(), scales.contiguous()


# ----------------------------------------------------------------------------
# camera algebra                            camera.py:38-44,250-277
# ----------------------------------------------------------------------------

def pose_inverse(pose: Tensor) -> Tensor:
    """[R|t] -> [R^T | -R^T t]   (camera.py:38-44)."""
    R, t = pose[..., :3], pose[..., 3:]
    Rt = R.transpose(-1, -2)
    return torch.cat([Rt, -Rt @ t], dim=-1)


def _hom(x: Tensor) -> Tensor:
    return torch.cat([x, torch.ones_like(x[..., :1])], dim=-1)


def pixels_to_rays(uv: Tensor, intr: Tensor, pose: Tensor) -> Tuple[Tensor, Tensor]:
    """uv [B,R,2] pixel coords -> (center, ray) [B,R,3] in the object frame.

    camera.py:266-277,308-314: g = K^-1 [u,v,1]; both g and the origin are sent
    through the inverted pose and subtracted (ray has camera-z == 1).
    """
    g = _hom(uv) @ intr.inverse().transpose(-1, -2)
    pinv = pose_inverse(pose).transpose(-1, -2)          # [B,4,3]
    gw = _hom(g) @ pinv
    cw = _hom(torch.zeros_like(g)) @ pinv
    return cw, gw - cw


# ----------------------------------------------------------------------------
# bilinear / nearest sampling restated by hand (torch.nn.functional.grid_sample
# semantics, zero padding).  Used by a2/a3/a16.
# ----------------------------------------------------------------------------

def bilinear_gather(img: Tensor, grid: Tensor) -> Tensor:
    """grid_sample(img, grid, 'bilinear', align_corners=True, zeros padding).

    img [B,C,H,W], grid [B,h,w,2] (x,y) in [-1,1] -> [B,C,h,w].
    Pixel coordinate: ix = (x+1)*(W-1)/2  (tools/ray_sampler.py:21,35-36,56-57).
    """
    B, C, H, W = img.shape
    x, y = grid[..., 0], grid[..., 1]
    # torch's CPU kernel unnormalises as (x+1)*((W-1)/2); the goldens were captured on CPU, and a
    # 1-ulp change of u,v is amplified ~1e3x by the 2^9*pi positional-encoding band.
    ix = (x + 1) * ((W - 1) / 2)
    iy = (y + 1) * ((H - 1) / 2)
    x0, y0 = torch.floor(ix), torch.floor(iy)
    x1, y1 = x0 + 1, y0 + 1
    wx1, wy1 = ix - x0, iy - y0
    wx0, wy0 = 1 - wx1, 1 - wy1
    flat = img.reshape(B, C, H * W)
    out = None
    # torch's vectorised CPU kernel accumulates nw,ne,sw,se as one multiply followed by three fused
    # multiply-adds (verified bit-exact against F.grid_sample); emulate the single rounding in fp64.
    for xx, yy, ww in ((x0, y0, wx0 * wy0), (x1, y0, wx1 * wy0),
                       (x0, y1, wx0 * wy1), (x1, y1, wx1 * wy1)):
        ok = (xx >= 0) & (xx <= W - 1) & (yy >= 0) & (yy <= H - 1)
        idx = (yy.clamp(0, H - 1) * W + xx.clamp(0, W - 1)).long()
        tap = torch.gather(flat, 2, idx.view(B, 1, -1).expand(B, C, -1)).view(B, C, *x.shape[1:])
        tap = tap * ok[:, None]
        if out is None:
            out = tap * ww[:, None]
        else:
            out = (tap.double() * ww[:, None].double() + out.double()).float()
    return out


def nearest_gather(img: Tensor, grid: Tensor) -> Tensor:
    """grid_sample(img, grid, 'nearest') with the DEFAULT align_corners=False.

    ix = (x+1)*W/2 - 0.5, rounded half-to-even, zero outside
    (model/nerf_adapt_st_gan.py:454-455,728-729).
    """
    B, C, H, W = img.shape
    x, y = grid[..., 0], grid[..., 1]
    ix = torch.round((x + 1) * (W / 2) - 0.5)     # torch.round = half-to-even
    iy = torch.round((y + 1) * (H / 2) - 0.5)
    ok = (ix >= 0) & (ix <= W - 1) & (iy >= 0) & (iy <= H - 1)
    idx = (iy.clamp(0, H - 1) * W + ix.clamp(0, W - 1)).long()
    flat = img.reshape(B, C, H * W)
    tap = torch.gather(flat, 2, idx.view(B, 1, -1).expand(B, C, -1)).view(B, C, *x.shape[1:])
    return tap * ok[:, None]


# ----------------------------------------------------------------------------
# a2/a3  RaySampler.get_rays / get_bounds    tools/ray_sampler.py:24-69
# ----------------------------------------------------------------------------

def rays_train(intr: Tensor, coords: Tensor, pose: Tensor, H: int, W: int) -> Tuple[Tensor, Tensor]:
    """Train-mode rays from continuous coords [B,h,w,2] -> center, ray [B,h,w,3].

    u,v are a bilinear lookup of INTEGER pixel index grids, i.e.
    u=(x+1)/2*(W-1), v=(y+1)/2*(H-1) with no half-pixel offset
    (tools/ray_sampler.py:49-57).
    """
    B, h, w, _ = coords.shape
    xs = torch.arange(W, dtype=torch.float32)[None, None, None, :].expand(B, 1, H, W)
    ys = torch.arange(H, dtype=torch.float32)[None, None, :, None].expand(B, 1, H, W)
    u = bilinear_gather(xs, coords)[:, 0]
    v = bilinear_gather(ys, coords)[:, 0]
    uv = torch.stack([u, v], dim=-1).view(B, h * w, 2)
    c, r = pixels_to_rays(uv, intr, pose)
    return c.view(B, h, w, 3), r.view(B, h, w, 3)


def bounds_train(coords: Tensor, z_near: Tensor, z_far: Tensor, H: int, W: int) -> Tuple[Tensor, Tensor]:
    """Bilinear sample of per-pixel near/far maps (tools/ray_sampler.py:24-37)."""
    B = coords.shape[0]
    zn = bilinear_gather(z_near.reshape(B, 1, H, W), coords)[:, 0]
    zf = bilinear_gather(z_far.reshape(B, 1, H, W), coords)[:, 0]
    return zn, zf


# ----------------------------------------------------------------------------
# a4/a5  eval rays                            camera.py:292-314, model/nerf_adapt_st_gan.py:702-710
# ----------------------------------------------------------------------------

def rays_eval(pose: Tensor, intr: Tensor, H: int, W: int) -> Tuple[Tensor, Tensor]:
    """All H*W rays through pixel centres (x+0.5, y+0.5) -> [B,HW,3] x2."""
    B = pose.shape[0]
    ys = torch.arange(H, dtype=torch.float32) + 0.5
    xs = torch.arange(W, dtype=torch.float32) + 0.5
    uv = torch.stack([xs[None, :].expand(H, W), ys[:, None].expand(H, W)], dim=-1).view(1, H * W, 2)
    return pixels_to_rays(uv.expand(B, -1, -1), intr, pose)


def gather_rows(x: Tensor, ray_idx: Tensor) -> Tensor:
    """x [B,HW,C], ray_idx [B,R] int64 -> [B,R,C] (model/nerf_adapt_st_gan.py:702-710)."""
    B, HW, C = x.shape
    assert ray_idx.shape[0] == B
    flat = x.reshape(B * HW, C)
    rows = ray_idx + HW * torch.arange(B)[:, None]
    return flat[rows].view(B, ray_idx.shape[1], C)


# ----------------------------------------------------------------------------
# a6  AABB slab test                         camera.py:415-440, compute_box.py:62-87
# ----------------------------------------------------------------------------

def aabb_slab(aabb_min: Tensor, aabb_max: Tensor, o: Tensor, d: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
    inv = torch.reciprocal(d)
    ta = (aabb_min - o) * inv
    tb = (aabb_max - o) * inv
    t_near = torch.minimum(ta, tb).max(dim=2).values
    t_far = torch.maximum(ta, tb).min(dim=2).values
    valid = (t_far > 0) & (t_far > t_near)
    return t_near, t_far, valid


def enlarge_diagonal(lo: Tensor, hi: Tensor, alpha: float = 0.25) -> Tuple[Tensor, Tensor]:
    diag = hi - lo
    return lo - diag * alpha / 2, hi + diag * alpha / 2


def box_bounds(aabb_min: Tensor, aabb_max: Tensor, o: Tensor, d: Tensor,
               bg_near: float, bg_far: float) -> Tuple[Tensor, Tensor]:
    """Online replacement of the pred_box npz pipeline: slab test, invalid or
    non-positive bounds fall back to the background range
    (compute_box.py:270-272 zeroes invalid rays; data/lm.py:349-350 maps
    non-positive near/far to depth.range*scale)."""
    tn, tf, ok = aabb_slab(aabb_min, aabb_max, o, d)
    tn = torch.where(ok, tn, torch.zeros_like(tn))
    tf = torch.where(ok, tf, torch.zeros_like(tf))
    zn = torch.where(tn > 0, tn, torch.full_like(tn, bg_near))
    zf = torch.where(tf > 0, tf, torch.full_like(tf, bg_far))
    return zn, zf


# ----------------------------------------------------------------------------
# a7  stratified depth samples               model/nerf_adapt_st_gan.py:683-700
# ----------------------------------------------------------------------------

def stratified_depths(near: Tensor, far: Tensor, n: int, rand: Optional[Tensor] = None, param: str = "metric") -> Tensor:
    """near, far [B,R] -> [B,R,n,1];  rand [B,R,n,1] in [0,1) or None (=0.5).  ``param`` = options nerf.depth.param
    (model/nerf_adapt_st_gan.py:699): 'inverse' returns 1 / (sample + 1e-8)."""
    near = near[:, :, None, None]
    far = far[:, :, None, None]
    r = rand if rand is not None else 0.5
    r = r + torch.arange(n)[None, None, :, None].float()
    z = r / n * (far - near) + near
    return {"metric": z, "inverse": 1 / (z + 1e-8)}[param]


def rays_to_ndc(center: Tensor, ray: Tensor, intr: Tensor, near: float = 1.0) -> Tuple[Tensor, Tensor]:
    """camera.convert_NDC (camera.py:325-342): rays [B,R,3] moved onto the plane z = near and re-expressed in normalised device
    coordinates of a camera looking down +z; intr [B,3,3]."""
    center = center + (near - center[..., 2:]) / ray[..., 2:] * ray
    cx, cy, cz = center.unbind(-1)
    rx, ry, rz = ray.unbind(-1)
    sx = (intr[:, 0, 0] / intr[:, 0, 2])[:, None]
    sy = (intr[:, 1, 1] / intr[:, 1, 2])[:, None]
    c = torch.stack([sx * (cx / cz), sy * (cy / cz), 1 - 2 * near / cz], dim=-1)
    d = torch.stack([sx * (rx / rz - cx / cz), sy * (ry / rz - cy / cz), 2 * near / cz], dim=-1)
    return c, d


# ----------------------------------------------------------------------------
# Philox4x32-10 (counter-based RNG used by the HIP ray-gen kernel for the
# stratified jitter in perf runs).  Integer arithmetic -> bit exact.
# Published algorithm: Salmon et al., "Parallel random numbers: as easy as
# 1, 2, 3" (SC'11).  Not a reference-code path: the reference draws torch.rand
# (model/nerf_adapt_st_gan.py:690) whose CPU/GPU streams already differ.
# ----------------------------------------------------------------------------

_PHILOX_M0, _PHILOX_M1 = 0xD2511F53, 0xCD9E8D57
_PHILOX_W0, _PHILOX_W1 = 0x9E3779B9, 0xBB67AE85


def philox4x32(counter: np.ndarray, key: Tuple[int, int]) -> np.ndarray:
    """counter uint32 [...,4], key (k0,k1) -> uint32 [...,4] after 10 rounds."""
    c = counter.astype(np.uint64)
    k0, k1 = np.uint64(key[0] & 0xFFFFFFFF), np.uint64(key[1] & 0xFFFFFFFF)
    mask = np.uint64(0xFFFFFFFF)
    c0, c1, c2, c3 = c[..., 0], c[..., 1], c[..., 2], c[..., 3]
    for _ in range(10):
        p0 = np.uint64(_PHILOX_M0) * c0
        p1 = np.uint64(_PHILOX_M1) * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & mask
        hi1, lo1 = p1 >> np.uint64(32), p1 & mask
        c0, c1, c2, c3 = (hi1 ^ c1 ^ k0) & mask, lo1, (hi0 ^ c3 ^ k1) & mask, lo0
        k0 = (k0 + np.uint64(_PHILOX_W0)) & mask
        k1 = (k1 + np.uint64(_PHILOX_W1)) & mask
    return np.stack([c0, c1, c2, c3], axis=-1).astype(np.uint32)


def philox_uniform(n_elems: int, seed: int, offset: int = 0) -> np.ndarray:
    """Uniform [0,1) float32 stream used by tp_raygen / tp_sample_depth: element e uses
    word (e & 3) of philox(counter=(lo32(e>>2), lo32(offset), hi32(e>>2), hi32(offset)),
    key=(seed_lo, seed_hi)), mapped as (w >> 8) * 2^-24."""
    e = np.arange(n_elems, dtype=np.uint64)
    ctr = np.zeros((n_elems, 4), dtype=np.uint32)
    cnt = e >> np.uint64(2)
    ctr[:, 0] = (cnt & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    ctr[:, 1] = np.uint32(offset & 0xFFFFFFFF)
    ctr[:, 2] = (cnt >> np.uint64(32)).astype(np.uint32)
    ctr[:, 3] = np.uint32((offset >> 32) & 0xFFFFFFFF)
    out = philox4x32(ctr, (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF))
    w = out[np.arange(n_elems), (e & np.uint64(3)).astype(np.int64)]
    return ((w >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)).astype(np.float32)


# ----------------------------------------------------------------------------
# a10  positional encoding                   layers/nerf_static_transient_light.py:217-234
# ----------------------------------------------------------------------------

def c2f_weight(L: int, progress: float, c2f_range, start=None) -> Tensor:
    """Coarse-to-fine weight of frequency band l < L (layers/nerf_static_transient_light.py:225-231): with
    alpha = (progress - range[0]) / (range[1] - range[0]) * L and k = l - (start or 0): (1 - cos(pi clamp(alpha - k, 0, 1))) / 2."""
    lo, hi = float(c2f_range[0]), float(c2f_range[1])
    alpha = (torch.tensor(float(progress), dtype=torch.float32) - lo) / (hi - lo) * L
    k = torch.arange(L, dtype=torch.float32) - (0 if start is None else start)
    return (1 - (alpha - k).clamp(min=0, max=1).mul(np.pi).cos()) / 2


def posenc(x: Tensor, L: int, weight: Optional[Tensor] = None) -> Tensor:
    """[...,C] -> [...,2*C*L], index = c*2L + s*L + l  (s: 0 sin, 1 cos).
    ``weight`` [L]: the c2f weights (`c2f_weight`), applied per band l to the sin and the cos entry (:232-233); None when
    c2f.range is None, as in the shipped yaml (options/nerf_lm_adapt_gan.yaml:4-6)."""
    freq = (2 ** torch.arange(L, dtype=torch.float32)) * np.pi
    spec = x[..., None] * freq
    enc = torch.stack([spec.sin(), spec.cos()], dim=-2)
    if weight is not None:
        enc = enc * weight
    return enc.reshape(*x.shape[:-1], -1)


# ----------------------------------------------------------------------------
# a11  MLP                                    layers/nerf_static_transient_light.py:16-61,76-145
# ----------------------------------------------------------------------------

TRUNK_DIMS = [(63, 256), (256, 256), (256, 256), (256, 256), (319, 256), (256, 256), (256, 256), (256, 257)]
RGB_DIMS = [(334, 256), (256, 256), (256, 256), (256, 3)]
TRANS_DIMS = [(272, 256), (256, 256), (256, 256), (256, 5)]


def make_params(seed: int, width: int = 256, n_lat_trans: int = 16, n_lat_light: int = 48,
                L_3D: int = 10, L_view: int = 4, bias_scale: float = 0.05) -> Dict[str, Tensor]:
    """Deterministic (numpy RandomState) Xavier-style weights keyed like the
    reference state dict.  Both the golden generator and the tests
    regenerate weights from this recipe so only outputs need to be stored.
    Gains follow tensorflow_init_weights (layers/...light.py:63-74); biases are
    small non-zero values so that bias handling is exercised."""
    rs = np.random.RandomState(seed)
    d3 = 3 + 6 * L_3D
    dv = 3 + 6 * L_view
    W = width
    trunk = [(d3, W), (W, W), (W, W), (W, W), (W + d3, W), (W, W), (W, W), (W, W + 1)]
    rgb = [(W + dv + 3 + n_lat_light, W), (W, W), (W, W), (W, 3)]
    trans = [(W + n_lat_trans, W), (W, W), (W, W), (W, 5)]
    p: Dict[str, Tensor] = {}

    def lin(prefix, dims, last_gain):
        for li, (k_in, k_out) in enumerate(dims):
            gain = math.sqrt(2.0) if li < len(dims) - 1 else last_gain
            bound = gain * math.sqrt(6.0 / (k_in + k_out))
            p[f"{prefix}.{li}.weight"] = torch.from_numpy(
                rs.uniform(-bound, bound, size=(k_out, k_in)).astype(np.float32))
            p[f"{prefix}.{li}.bias"] = torch.from_numpy(
                rs.uniform(-bias_scale, bias_scale, size=(k_out,)).astype(np.float32))

    lin("mlp_feat", trunk, math.sqrt(2.0))
    lin("mlp_rgb", rgb, 1.0)
    lin("mlp_trans", trans, 1.0)
    return p


def mlp_forward(p: Dict[str, Tensor], points: Tensor, ray_unit: Tensor,
                lat_trans: Tensor, lat_light: Tensor, L_3D: int = 10, L_view: int = 4,
                skip: Sequence[int] = (4,), taps: Optional[Dict[str, Tensor]] = None, c2f=None,
                density_noise: Optional[Tensor] = None) -> Tuple[Tensor, Tensor, Tensor]:
    """points, ray_unit [B,R,N,3]; lat_trans [B,Lt]; lat_light [B,Ll]
    -> rgb [B,R,N,3,2], density [B,R,N,2], uncert [B,R,N,1].
    ``taps`` (tests only): filled with the detached PRE-activations of the hidden head layers, ``mlp_rgb.{0,1,2}`` /
    ``mlp_trans.{0,1,2}`` -> [B,R,N,256] (the ReLU gates whose flips bound gradient parity).

    layers/nerf_static_transient_light.py:76-145.  The trunk runs without
    autograd in the reference (:87-100); here the caller decides by detaching.
    ``c2f`` = dict(progress, range, start): coarse-to-fine weights on BOTH encodings (:81,104 pass c2f=True).
    ``density_noise`` [B,R,N]: added to the static density's pre-activation (train mode with nerf.density_noise_reg, :96-97:
    the caller passes randn * reg).
    """
    w3 = wv = None
    if c2f is not None and c2f.get("range") is not None:
        w3 = c2f_weight(L_3D, c2f["progress"], c2f["range"], c2f.get("start"))
        wv = c2f_weight(L_view, c2f["progress"], c2f["range"], c2f.get("start"))
    B, R, N, _ = points.shape
    n_feat = len([k for k in p if k.startswith("mlp_feat.") and k.endswith(".weight")])
    n_rgb = len([k for k in p if k.startswith("mlp_rgb.") and k.endswith(".weight")])
    n_tr = len([k for k in p if k.startswith("mlp_trans.") and k.endswith(".weight")])
    # (two-argument calls when c2f is off: tests substitute an fp64 `posenc(x, L)`)
    enc = torch.cat([points, posenc(points, L_3D) if w3 is None else posenc(points, L_3D, w3)], dim=-1)
    h = enc
    with torch.no_grad():
        for li in range(n_feat):
            if li in skip:
                h = torch.cat([h, enc], dim=-1)
            h = torch.nn.functional.linear(h, p[f"mlp_feat.{li}.weight"], p[f"mlp_feat.{li}.bias"])
            if li == n_feat - 1:
                raw = h[..., 0] if density_noise is None else h[..., 0] + density_noise
                sigma_s = torch.nn.functional.softplus(raw)
                h = h[..., 1:]
            h = torch.relu(h)
    feat = h
    venc = torch.cat([ray_unit, posenc(ray_unit, L_view) if wv is None else posenc(ray_unit, L_view, wv)], dim=-1)
    light = lat_light[:, None, None, :].expand(B, R, N, lat_light.shape[-1])
    g = torch.cat([feat, venc, points, light], dim=-1)
    for li in range(n_rgb):
        g = torch.nn.functional.linear(g, p[f"mlp_rgb.{li}.weight"], p[f"mlp_rgb.{li}.bias"])
        if li != n_rgb - 1:
            if taps is not None:
                taps[f"mlp_rgb.{li}"] = g.detach()
            g = torch.relu(g)
    rgb_s = torch.sigmoid(g)
    tr = lat_trans[:, None, None, :].expand(B, R, N, lat_trans.shape[-1])
    t = torch.cat([feat, tr], dim=-1)
    for li in range(n_tr):
        t = torch.nn.functional.linear(t, p[f"mlp_trans.{li}.weight"], p[f"mlp_trans.{li}.bias"])
        if li != n_tr - 1:
            if taps is not None:
                taps[f"mlp_trans.{li}"] = t.detach()
            t = torch.relu(t)
    rgb_t = torch.sigmoid(t[..., :3])
    sigma_t = torch.nn.functional.softplus(t[..., 3])
    uncert = torch.nn.functional.softplus(t[..., 4:5])
    rgb = torch.stack([rgb_s, rgb_t], dim=-1)
    density = torch.stack([sigma_s, sigma_t], dim=-1)
    return rgb, density, uncert


# a9 + a12  forward_samples                  camera.py:317-322, layers/...light.py:147-166
def forward_samples(p, center: Tensor, ray: Tensor, depth_samples: Tensor,
                    lat_trans: Tensor, lat_light: Tensor, **kw):
    pts = center[:, :, None] + ray[:, :, None] * depth_samples
    unit = torch.nn.functional.normalize(ray, dim=-1)[:, :, None, :].expand_as(pts)
    return mlp_forward(p, pts, unit, lat_trans, lat_light, **kw)


# ----------------------------------------------------------------------------
# a13  composite                              layers/nerf_static_transient_light.py:168-212
# ----------------------------------------------------------------------------

def composite(ray: Tensor, rgb_samples: Tensor, density_samples: Tensor, depth_samples: Tensor,
              uncert_samples: Tensor, min_uncert: float = 0.05):
    """-> 11-tuple (rgb, rgb_static, rgb_transient, depth, opacity, opacity_static,
    opacity_transient, prob, uncert, alpha_static, alpha_transient)."""
    length = ray.norm(dim=-1, keepdim=True)                                   # [B,R,1]
    z = depth_samples[..., 0]                                                  # [B,R,N]
    dz = torch.cat([z[..., 1:] - z[..., :-1], torch.full_like(z[..., :1], 1e10)], dim=2)
    dist = dz * length
    tau_s = density_samples[..., 0] * dist
    tau_t = density_samples[..., 1] * dist
    tau = tau_s + tau_t

    def excl_T(x):
        return torch.exp(-torch.cat([torch.zeros_like(x[..., :1]), x[..., :-1]], dim=2).cumsum(dim=2))

    a_s, a_t, a = 1 - torch.exp(-tau_s), 1 - torch.exp(-tau_t), 1 - torch.exp(-tau)
    T, T_s, T_t = excl_T(tau), excl_T(tau_s), excl_T(tau_t)
    c_s, c_t = rgb_samples[..., 0], rgb_samples[..., 1]
    w_s, w_t = (T * a_s)[..., None], (T * a_t)[..., None]
    prob = (T * a)[..., None]
    ws_own, wt_own = (T_s * a_s)[..., None], (T_t * a_t)[..., None]
    rgb = (c_s * w_s + c_t * w_t).sum(dim=2)
    rgb_static = (ws_own * c_s).sum(dim=2)
    rgb_transient = (wt_own * c_t).sum(dim=2)
    opacity = prob.sum(dim=2)
    opacity_static = ws_own.sum(dim=2)
    opacity_transient = wt_own.sum(dim=2)
    uncert = (uncert_samples * w_t).sum(dim=2) + min_uncert
    depth = (depth_samples * ws_own).sum(dim=2)
    return (rgb, rgb_static, rgb_transient, depth, opacity, opacity_static, opacity_transient,
            prob, uncert, a_s, a_t)


RENDER_KEYS = ("rgb", "rgb_static", "rgb_transient", "opacity", "opacity_static", "opacity_transient",
               "uncert", "depth", "alpha_static", "alpha_transient", "density")


# ----------------------------------------------------------------------------
# a14/a15  render / render_by_slices          model/nerf_adapt_st_gan.py:547-680
# ----------------------------------------------------------------------------

def render(p, emb_trans: Tensor, emb_light: Tensor, pose: Tensor, intr: Tensor, ray_idx: Tensor,
           depth_range: Tuple[Tensor, Tensor], sample_idx, mode: str, H: int, W: int, n_samples: int,
           rand: Optional[Tensor] = None, transient: str = "zero", min_uncert: float = 0.05,
           ndc: bool = False, depth_param: str = "metric", **kw) -> Dict[str, Tensor]:
    """depth_range = (z_near [B,HW,1], z_far [B,HW,1]).  ``rand`` replaces the
    internal torch.rand draw of sample_depth ([B,R,N,1]); None = unstratified.  ``ndc`` / ``depth_param``: options camera.ndc
    (reference :581-583, after the bounds) and nerf.depth.param (:699)."""
    if mode == "train":
        B, h, w, _ = ray_idx.shape
        center, ray = rays_train(intr, ray_idx, pose, H, W)
        zn, zf = bounds_train(ray_idx, depth_range[0], depth_range[1], H, W)
        center, ray = center.view(B, h * w, 3), ray.view(B, h * w, 3)
        zn, zf = zn.view(B, h * w), zf.view(B, h * w)
    else:
        B = pose.shape[0]
        center, ray = rays_eval(pose, intr, H, W)
        center, ray = gather_rows(center, ray_idx), gather_rows(ray, ray_idx)
        zn = gather_rows(depth_range[0], ray_idx).squeeze(-1)
        zf = gather_rows(depth_range[1], ray_idx).squeeze(-1)
    if ndc:
        center, ray = rays_to_ndc(center, ray, intr)
    z = stratified_depths(zn, zf, n_samples, rand, depth_param)
    if mode == "train":
        lt, ll = emb_trans[sample_idx], emb_light[sample_idx]
    elif mode == "val":
        lt, ll = emb_trans[0][None], emb_light[0][None]
    else:
        if transient == "zero":
            lt = torch.zeros(B, emb_trans.shape[1])
        elif transient == "sample":
            lt = emb_trans[sample_idx][None]
        else:
            raise NotImplementedError
        ll = emb_light[sample_idx][None]
    rgb_s, den_s, unc_s = forward_samples(p, center, ray, z, lt, ll, **kw)
    out = composite(ray, rgb_s, den_s, z, unc_s, min_uncert)
    return dict(rgb=out[0], rgb_static=out[1], rgb_transient=out[2], opacity=out[4], opacity_static=out[5],
                opacity_transient=out[6], uncert=out[8], depth=out[3], alpha_static=out[9],
                alpha_transient=out[10], density=den_s)


def render_by_slices(p, emb_trans, emb_light, pose, intr, depth_range, object_mask, sample_idx, mode,
                     H, W, n_samples, chunk: int = 2048, rand_fn=None, min_uncert: float = 0.05, **kw):
    """val: every pixel in chunks; eval: object-mask pixels only, scattered into
    default-filled maps (model/nerf_adapt_st_gan.py:633-680).  ``rand_fn(c, R)``
    returns the jitter tensor for the chunk starting at c (or None)."""
    HW = H * W
    if mode == "val":
        parts = {k: [] for k in RENDER_KEYS}
        for c in range(0, HW, chunk):
            idx = torch.arange(c, min(c + chunk, HW))[None]
            r = render(p, emb_trans, emb_light, pose, intr, idx, depth_range, sample_idx, mode, H, W, n_samples,
                       rand=None if rand_fn is None else rand_fn(c, idx.shape[1]), min_uncert=min_uncert, **kw)
            for k in RENDER_KEYS:
                parts[k].append(r[k])
        return {k: torch.cat(v, dim=1) for k, v in parts.items()}
    obj = (object_mask.reshape(HW) > 0).nonzero(as_tuple=True)[0]
    out = {}
    for k in RENDER_KEYS:
        if k == "uncert":
            out[k] = torch.ones(1, HW, 1) * min_uncert
        elif k == "density":
            out[k] = torch.ones(1, HW, n_samples, 2)
        elif "rgb" in k:
            out[k] = torch.zeros(1, HW, 3)
        elif "alpha" in k:
            out[k] = torch.ones(1, HW, n_samples)
        else:
            out[k] = torch.zeros(1, HW, 1)
    for c in range(0, len(obj), chunk):
        idx = obj[c:c + chunk][None]
        r = render(p, emb_trans, emb_light, pose, intr, idx, depth_range, sample_idx, mode, H, W, n_samples,
                   rand=None if rand_fn is None else rand_fn(c, idx.shape[1]), min_uncert=min_uncert, **kw)
        for k in RENDER_KEYS:
            out[k][:, idx[0]] = r[k][0]
    return out


# ----------------------------------------------------------------------------
# a16  patch gathers                          model/nerf_adapt_st_gan.py:444-461,516-545,726-745
# ----------------------------------------------------------------------------

def patch_gather(coords: Tensor, image: Tensor, image_syn: Tensor, nocs: Tensor, normal: Tensor,
                 obj_mask: Tensor, mask_syn: Tensor) -> Dict[str, Tensor]:
    """coords [B,p,p,2]; image* / nocs / normal [B,3,H,W]; masks [B,H,W].
    Images bilinear align_corners=True, binarised masks nearest align_corners=False."""
    B, _, H, W = image.shape
    m = (obj_mask > 0).float().view(B, 1, H, W)
    ms = (mask_syn > 0).float().view(B, 1, H, W)
    out = dict(image=bilinear_gather(image, coords), image_syn=bilinear_gather(image_syn, coords),
               nocs=bilinear_gather(nocs, coords), normal=bilinear_gather(normal, coords),
               mask=nearest_gather(m, coords), mask_syn=nearest_gather(ms, coords))
    out["nocs_sample"] = out["nocs"] * out["mask_syn"]
    out["normal_sample"] = out["normal"] * out["mask_syn"]
    return out


def disc_patches(rgb: Tensor, g: Dict[str, Tensor]) -> Tuple[Tensor, Tensor]:
    """rgb [B,p*p,3] -> (patch_real, patch_fake) [B,9,p,p] (model/nerf_adapt_st_gan.py:516-545)."""
    B, _, p, _ = g["image"].shape
    rgb_img = rgb.view(B, p, p, 3).permute(0, 3, 1, 2)
    pad = torch.logical_and(g["mask_syn"] == 1, g["mask"] == 0).float()
    real = g["image"] * g["mask"] + rgb_img * pad
    geo = [g["nocs_sample"], g["normal_sample"]]
    return torch.cat([real] + geo, dim=1), torch.cat([rgb_img] + geo, dim=1)


# ----------------------------------------------------------------------------
# a17  render-consuming losses                model/nerf_adapt_st_gan.py:747-776, model/base.py:145-157
# ----------------------------------------------------------------------------

def nerf_losses(rgb: Tensor, uncert: Tensor, density: Tensor, g: Dict[str, Tensor]) -> Dict[str, Tensor]:
    """render / uncert / trans_reg terms (feat and gan_nerf need VGG / discriminator)."""
    B, _, p, _ = g["image"].shape
    rgb_img = rgb.view(B, p, p, 3).permute(0, 3, 1, 2)
    unc_img = uncert.view(B, p, p, 1).permute(0, 3, 1, 2)
    m = g["mask"]
    return dict(render=(m * ((g["image"] - rgb_img) ** 2 / unc_img ** 2)).sum() / (m.sum() + 1e-5),
                uncert=5 + torch.log(uncert ** 2).mean() / 2,
                trans_reg=density[..., -1].mean())


# ---------------------------------------------------------------------------------------------- f4: eval metrics
def ssim_window(window_size: int = 11, sigma: float = 1.5) -> Tensor:
    """Normalised 1-D Gaussian of the reference's SSIM (external/pohsun_ssim/pytorch_ssim/__init__.py:7-9)."""
    g = torch.tensor([math.exp(-(x - window_size // 2) ** 2 / float(2 * sigma ** 2)) for x in range(window_size)])
    return g / g.sum()


def ssim_map(img1: Tensor, img2: Tensor, window_size: int = 11) -> Tensor:
    """Per-pixel SSIM [B,C,H,W]: depthwise 11x11 Gaussian (outer product of the 1-D window), zero padding,
    C1 = 0.01^2, C2 = 0.03^2 (pytorch_ssim/__init__.py:17-37)."""
    C = img1.shape[1]
    w1 = ssim_window(window_size).unsqueeze(1)
    win = w1.mm(w1.t()).float()[None, None].expand(C, 1, window_size, window_size).contiguous()
    conv = lambda x: torch.nn.functional.conv2d(x, win, padding=window_size // 2, groups=C)
    mu1, mu2 = conv(img1), conv(img2)
    mu1_sq, mu2_sq, mu12 = mu1.pow(2), mu2.pow(2), mu1 * mu2
    s1, s2, s12 = conv(img1 * img1) - mu1_sq, conv(img2 * img2) - mu2_sq, conv(img1 * img2) - mu12
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    return ((2 * mu12 + C1) * (2 * s12 + C2)) / ((mu1_sq + mu2_sq + C1) * (s1 + s2 + C2))


def eval_metrics(rgb_static: Tensor, image: Tensor, obj_mask: Tensor, H: int, W: int,
                 out_hw: Optional[Tuple[int, int]] = None) -> Dict[str, Tensor]:
    """PSNR / SSIM of evaluate_full (model/nerf_adapt_st_gan.py:340-362): the static render [B,HW,3] against the
    MASKED image; when the data is not the 128x128 crop both are first resized to ``out_hw`` (480x640 in the
    reference): bilinear align_corners=False for the colours, nearest for the mask."""
    B = image.shape[0]
    rgb_map = rgb_static.view(B, H, W, 3).permute(0, 3, 1, 2)
    mask_map = obj_mask.view(B, H, W, 1).permute(0, 3, 1, 2)
    if out_hw is not None:
        F = torch.nn.functional
        image = F.interpolate(image, size=list(out_hw), mode="bilinear", align_corners=False)
        rgb_map = F.interpolate(rgb_map, size=list(out_hw), mode="bilinear", align_corners=False)
        mask_map = F.interpolate(mask_map, size=list(out_hw), mode="nearest")
    image_masked = image * mask_map
    mse = ((rgb_map.contiguous() - image_masked) ** 2).mean()
    return dict(mse=mse, psnr=-10 * mse.log10(), ssim=ssim_map(rgb_map, image_masked).mean())


def summarize(losses: Dict[str, Tensor], weights: Dict[str, Optional[float]]) -> Tensor:
    total = 0.0
    for k, v in losses.items():
        if weights.get(k) is not None:
            total = total + 10 ** float(weights[k]) * v
    return total


# ----------------------------------------------------------------------------
# synthetic "Duck-like" scene shared by tests, smoke and bench.
# ----------------------------------------------------------------------------

LINEMOD_K = [[572.4114, 0.0, 325.2611], [0.0, 573.57043, 242.04899], [0.0, 0.0, 1.0]]


def rotation_from_axis_angle(w: np.ndarray) -> np.ndarray:
    th = np.linalg.norm(w)
    if th < 1e-12:
        return np.eye(3)
    k = w / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + math.sin(th) * K + (1 - math.cos(th)) * (K @ K)


def synthetic_scene(H: int, W: int, B: int = 1, seed: int = 0, depth_scale: float = 10.0,
                    half_extent: float = 0.5, bg=(0.0, 30.0)):
    """LineMOD intrinsics scaled by H/480 (data/lmsyn2real.py:329-338), seeded pose at
    0.8 m * depth_scale, inflated +-half_extent cube, z_near/z_far from the slab test."""
    rs = np.random.RandomState(seed)
    s = H / 480.0
    K = np.array(LINEMOD_K, dtype=np.float64)
    K[:2] *= s
    intr = torch.from_numpy(np.tile(K[None], (B, 1, 1)).astype(np.float32))
    poses = []
    for _ in range(B):
        Rm = rotation_from_axis_angle(rs.uniform(-1, 1, size=3) * 1.2)
        t = np.array([0.02, -0.03, 0.8]) * depth_scale
        poses.append(np.concatenate([Rm, t[:, None]], axis=1))
    pose = torch.from_numpy(np.stack(poses).astype(np.float32))
    lo = torch.full((1, 1, 3), -half_extent)
    hi = torch.full((1, 1, 3), half_extent)
    lo, hi = enlarge_diagonal(lo, hi, 0.25)
    c, r = rays_eval(pose, intr, H, W)
    zn, zf = box_bounds(lo, hi, c, r, bg[0], bg[1])
    return dict(intr=intr, pose=pose, z_near=zn, z_far=zf, aabb_min=lo, aabb_max=hi)


# ----------------------------------------------------------------------------
# Deterministic state for a spectral-norm PatchGAN (used to pin the stock Discriminator module, SURVEY 8f-1,
# against the reference's layers/discriminator.py through golden g12): weights by SORTED key from a
# RandomState, u / v from one power-iteration step started at ones (so sigma is well-conditioned).
# ----------------------------------------------------------------------------

def seed_spectral_module(module: torch.nn.Module, seed: int, scale: float = 0.05) -> None:
    rs = np.random.RandomState(seed)
    sd = module.state_dict()
    with torch.no_grad():
        for k in sorted(sd):
            if k.endswith("weight_orig"):
                sd[k].copy_(torch.from_numpy(rs.normal(scale=scale, size=tuple(sd[k].shape)).astype(np.float32)))
        for k in sorted(sd):
            if k.endswith("weight_orig"):
                w2 = sd[k].reshape(sd[k].shape[0], -1)
                v = torch.nn.functional.normalize(w2.t() @ torch.ones(w2.shape[0]), dim=0)
                u = torch.nn.functional.normalize(w2 @ v, dim=0)
                sd[k[:-5] + "_u"].copy_(u)
                sd[k[:-5] + "_v"].copy_(v)
            elif sd[k].ndim == 0:
                sd[k].fill_(0.25)


# ----------------------------------------------------------------------------
# f2  checkpoint wire format (util.py:172-263): test helpers shared by the golden generator and the tests
# ----------------------------------------------------------------------------

def _key_rng(key: str, salt: int) -> np.random.RandomState:
    import zlib
    return np.random.RandomState((zlib.crc32(key.encode()) ^ (salt * 2654435761)) & 0x7FFFFFFF)


def seeded_state(state: Dict[str, Tensor], salt: int, scale: float = 0.05) -> Dict[str, Tensor]:
    """Every floating-point tensor of a state dict as a function of its KEY (and ``salt``): the reference (generator) and
    the mirror (tests) fill their graphs with identical contents without shipping a blob."""
    out = {}
    for k, v in state.items():
        if v.dtype.is_floating_point:
            t = torch.from_numpy(_key_rng(k, salt).standard_normal(tuple(v.shape)).astype(np.float32)) * scale
            if k.endswith(("weight_u", "weight_v")):
                t = torch.nn.functional.normalize(t, dim=0, eps=1e-12)
            out[k] = t.to(v.dtype)
        else:
            out[k] = v.clone()
    return out


def seeded_grads(optim: torch.optim.Optimizer, salt: int, scale: float = 1e-3) -> None:
    """Key-free seeded gradients for every trainable parameter of an optimiser (index within its group as the key)."""
    for gi, g in enumerate(optim.param_groups):
        for pi, p in enumerate(g["params"]):
            if p.requires_grad:
                rs = _key_rng("g%d.p%d" % (gi, pi), salt)
                p.grad = torch.from_numpy(rs.standard_normal(tuple(p.shape)).astype(np.float32)).to(p.device) * scale
            else:
                p.grad = None


def _tensor_summary(t: Tensor):
    d = t.detach().double().cpu()
    return dict(shape=list(t.shape), dtype=str(t.dtype).replace("torch.", ""), sum=float(d.sum()), abssum=float(d.abs().sum()))


def state_summary(state: Dict[str, Tensor]):
    return {k: _tensor_summary(v) for k, v in state.items()}


def optim_summary(sd: dict):
    groups = [{k: (list(v) if isinstance(v, (list, tuple)) else (float(v) if isinstance(v, (int, float)) and not isinstance(v, bool) else v))
               for k, v in g.items()} for g in sd["param_groups"]]
    for g in groups:
        for k, v in list(g.items()):
            if torch.is_tensor(v):
                g[k] = float(v)
    state = {str(i): {n: (_tensor_summary(t) if torch.is_tensor(t) else t) for n, t in st.items()} for i, st in sd["state"].items()}
    return dict(param_groups=groups, state=state)


def checkpoint_manifest(blob: dict):
    """What a checkpoint file of the reference layout contains, in comparable form."""
    m = dict(top_level=sorted(blob.keys()), epoch=blob.get("epoch"), iter=blob.get("iter"), graph=state_summary(blob["graph"]))
    for k, v in blob.items():
        if k.startswith("optim"):
            m[k] = optim_summary(v)
        elif k.startswith("sched"):
            m[k] = {n: (x if isinstance(x, (int, float, bool, str, type(None))) else [float(y) for y in x] if isinstance(x, (list, tuple)) else str(type(x)))
                    for n, x in v.items()}
    return m


# ----------------------------------------------------------------------------
# f3  stored box-bound maps -> per-crop depth range        data/lm.py:316-350,412-495
# ----------------------------------------------------------------------------
# The reference stores the slab-test result of EVERY frame as a [2,480,640] map in millimetres (compute_box.py:262-283)
# and, at load time, crops it around the detection box, resizes the crop to the network resolution with
# cv2.resize(INTER_LINEAR), pads it into a square, converts mm -> depth.scale units and replaces non-positive entries
# by the background range.  cv2 (opencv-python, no version pinned by the reference) is not available offline: its
# float32 INTER_LINEAR resize is restated here from OpenCV's documented algorithm (pixel-centre alignment, border
# replicate, horizontal then vertical two-tap passes in float).

def resize_linear(img: np.ndarray, out_w: int, out_h: int) -> np.ndarray:
    """cv2.resize(img, (out_w, out_h), interpolation=cv2.INTER_LINEAR) for float32 [h, w] or [h, w, c]."""
    src = np.asarray(img, dtype=np.float32)
    squeeze = src.ndim == 2
    if squeeze:
        src = src[:, :, None]
    h, w, _ = src.shape

    def taps(n_out, n_in):
        scale = n_in / float(n_out)
        f = ((np.arange(n_out, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
        i0 = np.floor(f).astype(np.int64)
        frac = (f - i0.astype(np.float32)).astype(np.float32)
        lo = i0 < 0
        frac[lo], i0[lo] = 0.0, 0
        hi = i0 >= n_in - 1
        frac[hi], i0[hi] = 0.0, n_in - 1
        i1 = np.minimum(i0 + 1, n_in - 1)
        return i0, i1, frac

    x0, x1, fx = taps(out_w, w)
    y0, y1, fy = taps(out_h, h)
    rows = src[:, x0, :] * (np.float32(1.0) - fx)[None, :, None] + src[:, x1, :] * fx[None, :, None]      # horizontal pass
    out = rows[y0] * (np.float32(1.0) - fy)[:, None, None] + rows[y1] * fy[:, None, None]                 # vertical pass
    out = out.astype(np.float32)
    return out[:, :, 0] if squeeze else out


def crop_by_pad(img: np.ndarray, center, scale: int, res: int) -> np.ndarray:
    """Dataset.Crop_by_Pad(img, center, scale, res, channel<=3, resize=True) (data/lm.py:455-495): crop the square of side
    ``scale`` around ``center`` = (row, col) clipped to the image, resize the longer side to ``res`` keeping the aspect,
    paste centred into a zero [res, res, c] canvas."""
    ht, wd = img.shape[0], img.shape[1]
    up0, le0 = int(center[0] - scale / 2. + 0.5), int(center[1] - scale / 2. + 0.5)
    upper, left = max(0, up0), max(0, le0)
    bottom, right = min(ht, up0 + int(scale)), min(wd, le0 + int(scale))
    crop_ht, crop_wd = float(bottom - upper), float(right - left)
    if crop_ht > crop_wd:
        resize_ht, resize_wd = res, int(res / crop_ht * crop_wd + 0.5)
    elif crop_ht < crop_wd:
        resize_wd, resize_ht = res, int(res / crop_wd * crop_ht + 0.5)
    else:
        resize_wd = resize_ht = int(res)
    tmp = resize_linear(img[upper:bottom, left:right], resize_wd, resize_ht)
    if tmp.ndim < 3:
        tmp = tmp[:, :, None]
    out = np.zeros((res, res, img.shape[2] if img.ndim == 3 else 1))
    r0, c0 = int(res / 2.0 - resize_ht / 2.0 + 0.5), int(res / 2.0 - resize_wd / 2.0 + 0.5)
    out[r0:r0 + resize_ht, c0:c0 + resize_wd, :] = tmp
    return out


def crop_center_offset(center, scale: int, ht: int, wd: int) -> np.ndarray:
    """Dataset.get_center_offset (data/lm.py:431-451): shift of the effective crop centre when the square is clipped."""
    up0, le0 = int(center[0] - scale / 2. + 0.5), int(center[1] - scale / 2. + 0.5)
    upper, left = max(0, up0), max(0, le0)
    bottom, right = min(ht, up0 + int(scale)), min(wd, le0 + int(scale))
    h_off = -up0 / 2 if upper == 0 else (-(up0 + int(scale) - ht) / 2 if bottom == ht else 0)
    w_off = -le0 / 2 if left == 0 else (-(le0 + int(scale) - wd) / 2 if right == wd else 0)
    return np.array([h_off, w_off])


def crop_intrinsics(K: Tensor, resize: float, crop_center, res: int) -> Tensor:
    """Dataset.preprocess_intrinsics (data/lm.py:412-428): intrinsics of the resized, cropped image."""
    K = K.clone()
    K[0, 0], K[1, 1] = K[0, 0] * resize, K[1, 1] * resize
    K[0, 2] = (K[0, 2] + 0.5) * resize - 0.5
    K[1, 2] = (K[1, 2] + 0.5) * resize - 0.5
    top_left = np.asarray(crop_center, dtype=np.float64) * resize - res / 2
    K[0, 2] = K[0, 2] - top_left[1]
    K[1, 2] = K[1, 2] - top_left[0]
    return K


def range_from_box_map(box_map_mm: np.ndarray, center, scale: int, res: int, depth_scale: float = 10.0,
                       bg_range_m=(0.0, 3.0), mask: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    """Dataset.get_range, range_source == 'box' (data/lm.py:316-350): [2,H0,W0] map in mm -> (z_near, z_far) [res*res] in
    depth.scale units, non-positive entries replaced by the background range."""
    m = crop_by_pad(np.asarray(box_map_mm, dtype=np.float32).transpose(1, 2, 0), center, scale, res).astype(np.float32)
    r = torch.from_numpy(m)
    if mask is not None:
        r = r * mask[..., None]
    r = r.permute(2, 0, 1).reshape(2, res * res)
    r = (r / 1000) * depth_scale
    lo = torch.full((res * res,), float(bg_range_m[0] * depth_scale))
    hi = torch.full((res * res,), float(bg_range_m[1] * depth_scale))
    return torch.where(r[0] > 0, r[0], lo), torch.where(r[1] > 0, r[1], hi)
