#!/usr/bin/env python3
"""G17: BASELINE config C1 at its LITERAL size -- 64x64 crop, 32 samples per ray, batch 1 -- rendered by the REAL reference
on CPU: `Graph.render(mode='train')` over a 64x64 patch (4,096 rays, stratified jitter from torch.rand) and
`Graph.render_by_slices(mode='val')` over all 4,096 pixels (mid-point samples), with the rays / depth samples / latent rows
the reference itself fed to NeRF.forward_samples captured by a wrapper (as in G9b: torch's CPU inverse is not correctly
rounded, so "identical rays" are the reference's own).

Run in the build container only:   python tests/golden/make_golden_g17_c1.py

Stored: scene inputs (intr, pose, bounds, patch coords, sample_idx, weight seed), the captured forward_samples inputs, and the
reference's outputs -- all 14 per-ray values of both renders, density / alpha_static / alpha_transient of the train render."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG                                            # noqa: E402

PER_RAY = ("rgb", "rgb_static", "rgb_transient", "depth", "opacity", "opacity_static", "opacity_transient", "uncert")


def main():
    opt, camera, M, NeRF, RaySampler, FlexPatchSampler = MG._load_reference()
    from oracle import texpose_oracle as O
    T = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float32))
    B, H, W, N, n_train, seed_w = 1, 64, 64, 32, 7, 17
    opt.H, opt.W, opt.batch_size, opt.patch_size = H, W, B, 64
    opt.nerf.sample_intvs = N
    opt.nerf.rand_rays = H * W
    opt.data.image_size = [H, W]
    sc = MG._scene(B, H, W, seed=21)
    g = M.Graph(opt)
    sd = g.nerf.state_dict()
    sd.update(O.make_params(seed_w))
    g.nerf.load_state_dict(sd)
    g.latent_vars_trans = torch.nn.Embedding(n_train, 16)
    g.latent_vars_light = torch.nn.Embedding(n_train, 48)
    ers = np.random.RandomState(78)
    emb_t, emb_l = T(ers.normal(size=(n_train, 16))), T(ers.normal(size=(n_train, 48)))
    with torch.no_grad():
        g.latent_vars_trans.weight.copy_(emb_t)
        g.latent_vars_light.weight.copy_(emb_l)
    seen = []
    orig = g.nerf.forward_samples

    def spy(opt_, center, ray, depth_samples, latent_variable_trans=None, latent_variable_light=None, mode=None):
        seen.append(dict(center=center.detach().clone(), ray=ray.detach().clone(), depth=depth_samples.detach().clone(),
                         lat_t=latent_variable_trans.detach().clone(), lat_l=latent_variable_light.detach().clone()))
        return orig(opt_, center, ray, depth_samples, latent_variable_trans=latent_variable_trans,
                    latent_variable_light=latent_variable_light, mode=mode)

    g.nerf.forward_samples = spy
    dr = (sc["z_near"][:, :, None], sc["z_far"][:, :, None])
    # train: a 64x64 patch at scale 0.9, shifted (FlexPatchSampler's grid with fixed draws)
    lin = torch.linspace(-1, 1, 64)
    yy, xx = torch.meshgrid(lin, lin, indexing="ij")
    coords = (torch.stack([xx, yy], -1) * 0.9 + torch.tensor([0.04, -0.06]))[None].contiguous()
    idx = torch.tensor([3])
    torch.manual_seed(43)
    with torch.no_grad():
        ret = g.render(opt, sc["pose"], intr=sc["intr"], ray_idx=coords, depth_range=dr, sample_idx=idx, mode="train")
    out = dict(H=H, W=W, N=N, n_train=n_train, seed_w=seed_w, intr=sc["intr"], pose=sc["pose"], z_near=sc["z_near"], z_far=sc["z_far"],
               coords=coords, sample_idx=idx, emb_seed=78)
    out.update({"train_in_" + k: v for k, v in seen[-1].items()})
    out.update({"train_" + k: ret[k] for k in PER_RAY + ("density", "alpha_static", "alpha_transient")})
    opt.nerf.sample_stratified = False
    with torch.no_grad():
        val = g.render_by_slices(opt, sc["pose"], intr=sc["intr"], depth_range=dr, object_mask=torch.ones(1, H, W), sample_idx=None,
                                 mode="val")
    assert len(seen) == 2
    out.update({"val_in_" + k: v for k, v in seen[-1].items() if k != "depth"})     # (mid-point depths follow from the bounds)
    out.update({"val_" + k: val[k] for k in PER_RAY})
    MG._save("g17_c1_literal", **out)


if __name__ == "__main__":
    main()
