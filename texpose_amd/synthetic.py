"""Synthetic "Duck-like" training batch: there is no dataset offline, so benches and tests use
seeded random crops with the data layer's output contract (reference data/lm.py:112-159)."""
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
