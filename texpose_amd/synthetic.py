"""Synthetic "Duck-like" data: there is no dataset offline, so benches and tests use seeded recipes with
the data layer's output contract (reference data/lm.py:112-159): a training batch of random crops, an
evaluation scene (LineMOD intrinsics, seeded pose, object box) whose per-pixel depth bounds come from the HIP slab
test, and Xavier-style network weights.  The recipes are pure numpy so that the CPU oracle (which keeps its own copy
of them) regenerates the identical data; tests/test_host_logic_cpu.py checks the two copies against each other."""
from __future__ import annotations

import math

import numpy as np
import torch

from .options import AttrDict


def _rotation(w):
    th = float(np.linalg.norm(w))
    if th < 1e-12:
        return np.eye(3)
    k = w / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + math.sin(th) * K + (1 - math.cos(th)) * (K @ K)


def training_batch(B: int, H: int = 128, W: int = 128, n_train: int = 189, seed: int = 0, device="cuda:0",
                   depth_scale: float = 10.0) -> AttrDict:
    """image/image_syn/nocs_pred/normal_pred [B,3,H,W], obj_mask/mask_syn [B,H,W], intr [B,3,3], pose/pose_init
    [B,3,4] (t in dm), z_near/z_far [B,HW] around an object at 0.8 m, idx [B]."""
    rs = np.random.RandomState(seed)
    f = lambda *s: torch.from_numpy(rs.uniform(size=s).astype(np.float32))
    yy, xx = np.mgrid[0:H, 0:W]
    disk = ((yy - H / 2 + 0.5) ** 2 + (xx - W / 2 + 0.5) ** 2 < (0.38 * H) ** 2).astype(np.float32)
    disk2 = ((yy - H / 2 - 1.5) ** 2 + (xx - W / 2 + 2.5) ** 2 < (0.40 * H) ** 2).astype(np.float32)
    K = np.array([[700.0 * H / 128, 0, W / 2.0], [0, 700.0 * H / 128, H / 2.0], [0, 0, 1]], dtype=np.float32)
    poses = []
    for _ in range(B):
        Rm = _rotation(rs.uniform(-1, 1, size=3) * 1.2)
        t = np.array([0.0, 0.0, 0.8]) * depth_scale
        poses.append(np.concatenate([Rm, t[:, None]], axis=1).astype(np.float32))
    pose = torch.from_numpy(np.stack(poses))
    z0 = 0.8 * depth_scale
    var = AttrDict(
        idx=torch.from_numpy(rs.randint(0, n_train, size=B).astype(np.int64)),
        image=f(B, 3, H, W), image_syn=f(B, 3, H, W), nocs_pred=f(B, 3, H, W), normal_pred=f(B, 3, H, W) * 2 - 1,
        obj_mask=torch.from_numpy(np.tile(disk[None], (B, 1, 1))), mask_syn=torch.from_numpy(np.tile(disk2[None], (B, 1, 1))),
        intr=torch.from_numpy(np.tile(K[None], (B, 1, 1))), pose=pose, pose_init=pose.clone(),
        z_near=torch.full((B, H * W), z0 - 0.9), z_far=torch.full((B, H * W), z0 + 0.9),
        frame_index=torch.arange(B))
    for k, v in list(var.items()):
        var[k] = v.to(device)
    return var


LINEMOD_K = [[572.4114, 0.0, 325.2611], [0.0, 573.57043, 242.04899], [0.0, 0.0, 1.0]]      # compute_box.py:166-168


def eval_scene(H: int, W: int, B: int = 1, seed: int = 0, depth_scale: float = 10.0, half_extent: float = 0.5):
    """intr [B,3,3] (LineMOD K scaled by H/480, data/lmsyn2real.py:329-338), pose [B,3,4] (seeded rotation, object at
    0.8 m x depth_scale), aabb_min / aabb_max [1,1,3] (+-half_extent cube inflated by a quarter of its diagonal,
    compute_box.py:245-252).  CPU tensors; the bounds come from ``scene_bounds``."""
    from .geometry import enlarge_diagonal
    rs = np.random.RandomState(seed)
    K = np.array(LINEMOD_K, dtype=np.float64)
    K[:2] *= H / 480.0
    intr = torch.from_numpy(np.tile(K[None], (B, 1, 1)).astype(np.float32))
    poses = []
    for _ in range(B):
        Rm = _rotation(rs.uniform(-1, 1, size=3) * 1.2)
        t = np.array([0.02, -0.03, 0.8]) * depth_scale
        poses.append(np.concatenate([Rm, t[:, None]], axis=1))
    pose = torch.from_numpy(np.stack(poses).astype(np.float32))
    lo, hi = enlarge_diagonal(torch.full((1, 1, 3), -half_extent), torch.full((1, 1, 3), half_extent), 0.25)
    return dict(intr=intr, pose=pose, aabb_min=lo, aabb_max=hi)


def scene_bounds(scene, H: int, W: int, device, bg=(0.0, 30.0)):
    """Per-pixel (z_near, z_far) [B,HW] of ``scene`` through the HIP ray-gen + slab test (tp_raygen, bounds from the
    box; misses fall back to ``bg`` as data/lm.py:349-350 does)."""
    from . import ops
    B = scene["pose"].shape[0]
    idx = torch.arange(H * W, device=device)[None].expand(B, -1).contiguous()
    lo, hi = scene["aabb_min"].reshape(3).tolist(), scene["aabb_max"].reshape(3).tolist()
    _, _, near, far, _ = ops.raygen(scene["intr"].to(device), scene["pose"].to(device), H=H, W=W, ray_idx=idx,
                                    aabb=(lo, hi), bg_range=bg)
    return near, far


def network_weights(seed: int, width: int = 256, n_lat_trans: int = 16, n_lat_light: int = 48, L_3D: int = 10,
                    L_view: int = 4, bias_scale: float = 0.05):
    """Deterministic Xavier-style weights keyed like the reference state dict (gains of
    tensorflow_init_weights, layers/nerf_static_transient_light.py:63-74); small non-zero biases."""
    rs = np.random.RandomState(seed)
    d3, dv, Wd = 3 + 6 * L_3D, 3 + 6 * L_view, width
    trunk = [(d3, Wd), (Wd, Wd), (Wd, Wd), (Wd, Wd), (Wd + d3, Wd), (Wd, Wd), (Wd, Wd), (Wd, Wd + 1)]
    rgb = [(Wd + dv + 3 + n_lat_light, Wd), (Wd, Wd), (Wd, Wd), (Wd, 3)]
    trans = [(Wd + n_lat_trans, Wd), (Wd, Wd), (Wd, Wd), (Wd, 5)]
    p = {}
    for prefix, dims, last_gain in (("mlp_feat", trunk, math.sqrt(2.0)), ("mlp_rgb", rgb, 1.0), ("mlp_trans", trans, 1.0)):
        for li, (k_in, k_out) in enumerate(dims):
            gain = math.sqrt(2.0) if li < len(dims) - 1 else last_gain
            bound = gain * math.sqrt(6.0 / (k_in + k_out))
            p[f"{prefix}.{li}.weight"] = torch.from_numpy(rs.uniform(-bound, bound, size=(k_out, k_in)).astype(np.float32))
            p[f"{prefix}.{li}.bias"] = torch.from_numpy(rs.uniform(-bias_scale, bias_scale, size=(k_out,)).astype(np.float32))
    return p
