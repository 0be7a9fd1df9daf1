#!/usr/bin/env python3
"""G14: evaluation metrics from the REAL reference pieces: Graph.MSE_loss, external/pohsun_ssim
pytorch_ssim.ssim and the resize calls of evaluate_full (model/nerf_adapt_st_gan.py:340-362), on small images.

    python tests/golden/make_golden_g14.py          (build container only; needs /root/reference)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG                                           # noqa: E402


def main():
    opt, camera, M, NeRF, RaySampler, FlexPatchSampler = MG._load_reference()
    from external.pohsun_ssim import pytorch_ssim
    import torch.nn.functional as torch_F
    rs = np.random.RandomState(2024)
    T = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float32))
    out = {}
    for name, (B, H, W, out_hw) in dict(native=(2, 24, 20, None), resized=(1, 15, 20, (36, 48)),
                                        big=(1, 40, 56, None)).items():
        image = T(rs.uniform(size=(B, 3, H, W)))
        # smooth-ish render so that SSIM is not ~0: blend of the image and noise
        rgb_static = (0.7 * image + 0.3 * T(rs.uniform(size=(B, 3, H, W)))).permute(0, 2, 3, 1).reshape(B, H * W, 3).contiguous()
        yy, xx = np.mgrid[0:H, 0:W]
        obj_mask = T(((yy - H / 2) ** 2 + (xx - W / 2) ** 2 < (0.4 * H) ** 2).astype(np.float32))[None].repeat(B, 1, 1)
        # evaluate_full lines 341-362, verbatim calls on our tensors
        rgb_map = rgb_static.view(-1, H, W, 3).permute(0, 3, 1, 2)
        mask_map = obj_mask.view(-1, H, W, 1).permute(0, 3, 1, 2)
        img = image
        if out_hw is not None:
            img = torch_F.interpolate(img, size=list(out_hw), mode='bilinear', align_corners=False)
            rgb_map = torch_F.interpolate(rgb_map, size=list(out_hw), mode='bilinear', align_corners=False)
            mask_map = torch_F.interpolate(mask_map, size=list(out_hw), mode='nearest')
        image_masked = img * mask_map
        mse = M.Graph.MSE_loss(None, rgb_map, image_masked)          # bound call in the reference (self.graph.MSE_loss)
        psnr = -10 * mse.log10()
        ssim = pytorch_ssim.ssim(rgb_map, image_masked)
        out.update({f"{name}.rgb_static": rgb_static.numpy(), f"{name}.image": image.numpy(), f"{name}.obj_mask": obj_mask.numpy(),
                    f"{name}.H": H, f"{name}.W": W, f"{name}.out_h": 0 if out_hw is None else out_hw[0],
                    f"{name}.out_w": 0 if out_hw is None else out_hw[1],
                    f"{name}.mse": np.float64(mse.item()), f"{name}.psnr": np.float64(psnr.item()),
                    f"{name}.ssim": np.float64(ssim.item())})
        print(name, float(mse), float(psnr), float(ssim))
    path = os.path.join(HERE, "g14_eval_metrics.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path))


if __name__ == "__main__":
    main()
