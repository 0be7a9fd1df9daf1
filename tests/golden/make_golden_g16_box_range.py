#!/usr/bin/env python3
"""G16: what the reference's data layer makes of a stored box-bound map.

Run in the build container only:   python tests/golden/make_golden_g16_box_range.py

A synthetic frame (LineMOD intrinsics, seeded pose in mm, object box in mm) goes through the reference's own code:
camera.get_center_and_ray + camera.aabb_ray_intersection build the [2,480,640] map exactly as compute_box.py:262-283
stores it; data/lm.py's Dataset.get_center_offset / preprocess_intrinsics / Crop_by_Pad (static methods of the real
module) and the unit conversion + background fallback of get_range (:343-350, restated line by line below because the
method itself needs the dataset on disk) turn it into the (z_near, z_far) [128*128] the network is fed.
cv2 is not installed: cv2.resize is bound to oracle.resize_linear (OpenCV's float32 INTER_LINEAR, restated).
Two cases: a crop inside the image and one clipped by the image border (non-zero centre offset)."""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG                                            # noqa: E402


def main():
    opt, camera, M, NeRF, RaySampler, FlexPatchSampler = MG._load_reference()
    from oracle import texpose_oracle as O
    cv2 = sys.modules["cv2"]
    cv2.resize = lambda img, size, interpolation=None: O.resize_linear(img, size[0], size[1])
    sys.modules["plyfile"] = types.ModuleType("plyfile")
    import data.lm as L
    D = L.Dataset
    res, depth_scale, bg = 128, 10.0, (0.0, 3.0)
    out = dict(res=res, depth_scale=depth_scale, bg_lo=bg[0], bg_hi=bg[1])
    rs = np.random.RandomState(16)
    K = torch.tensor(O.LINEMOD_K, dtype=torch.float32)
    lo = torch.tensor([-38.0, -52.0, -45.0]).view(1, 1, 3)
    hi = torch.tensor([41.0, 47.0, 43.0]).view(1, 1, 3)
    lo, hi = camera.enlarge_diagonal(lo, hi, alpha=0.25)                                      # compute_box.py:252
    for case, (t_mm, bbox) in enumerate((((25.0, -30.0, 760.0), (300, 160, 110, 120)),       # (x_ul, y_ul, h, w) inside the image
                                         ((330.0, 150.0, 700.0), (520, 300, 140, 130)))):     # clipped at the right / bottom border
        Rm = O.rotation_from_axis_angle(rs.uniform(-1, 1, size=3) * 1.1).astype(np.float32)
        pose = torch.eye(4)[None]
        pose[..., :3, :3] = torch.from_numpy(Rm)
        pose[..., :3, 3] = torch.tensor(t_mm)
        pose = pose[:, :3]
        ray_o, ray_d = camera.get_center_and_ray(opt, pose, intr=K[None], H=480, W=640)       # == compute_box.py:69-87
        t_near, t_far, valid = camera.aabb_ray_intersection(lo, hi, ray_o, ray_d)
        t_near = torch.where(valid > 0, t_near, torch.zeros_like(t_near)).view(480, 640)
        t_far = torch.where(valid > 0, t_far, torch.zeros_like(t_far)).view(480, 640)
        box_bound = torch.stack([t_near, t_far], 0).numpy()                                   # what compute_box.py saves
        x_ul, y_ul, h, w = bbox
        center = np.array([int(y_ul + h / 2), int(x_ul + w / 2)])                             # get_2d_bbox, box_format None
        scale = int(1.5 * max(h, w))
        resize = res / scale
        off = D.get_center_offset(center, scale, 480, 640)
        intr = D.preprocess_intrinsics(K.clone(), resize, center + off, res=res)
        box_range = box_bound.astype(np.float32).transpose((1, 2, 0))
        box_range = D.Crop_by_Pad(box_range, center, scale, res, channel=2).astype(np.float32)
        box_range = torch.from_numpy(box_range)
        box_range = box_range.permute(2, 0, 1).view(2, res * res)
        box_range = (box_range / 1000) * depth_scale
        dmin = torch.Tensor([bg[0] * depth_scale]).float().expand(res * res)
        dmax = torch.Tensor([bg[1] * depth_scale]).float().expand(res * res)
        z_near = torch.where(box_range[0] > 0, box_range[0], dmin)
        z_far = torch.where(box_range[1] > 0, box_range[1], dmax)
        pre = "c%d_" % case
        out.update({pre + "K": K, pre + "R": torch.from_numpy(Rm), pre + "t_mm": torch.tensor(t_mm), pre + "aabb_min_mm": lo.view(3),
                    pre + "aabb_max_mm": hi.view(3), pre + "center": torch.from_numpy(center.astype(np.int64)),
                    pre + "scale": scale, pre + "center_offset": torch.from_numpy(off.astype(np.float32)), pre + "intr_crop": intr,
                    pre + "z_near": z_near, pre + "z_far": z_far,
                    pre + "hit_fraction_full": float((t_far > 0).float().mean())})
        print("case", case, "scale", scale, "offset", off, "hit pixels in crop:", int((box_range[1] > 0).sum()))
    MG._save("g16_box_range", **out)


if __name__ == "__main__":
    main()
