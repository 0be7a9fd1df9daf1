#!/usr/bin/env python3
"""G20: the two ray-parametrisation options of hot-path rows a4 / a7 / a14 that the reference's shipped yaml leaves off but its code
implements:
  (a) `camera.ndc` (model/nerf_adapt_st_gan.py:581-583 -> camera.py:325-342 convert_NDC): centre / ray of eval and train rays in
      normalised device coordinates;
  (b) `nerf.depth.param = inverse` (model/nerf_adapt_st_gan.py:699): 1 / (sample + 1e-8), unstratified and with the stored draw;
  (c) Graph.render with each option on (train mode, with the head / latent-row gradients of a stored cotangent, for ndc; val mode
      for inverse depths) and with both on.

    python tests/golden/make_golden_g20_ndc_inverse.py         (build container only; needs /root/reference)

Weights are recipes (oracle make_params); inputs and expected outputs are stored.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG                                           # noqa: E402
import make_golden_g13 as G13                                      # noqa: E402  (pack: strided subsample + norm of big tensors)


def main():
    from oracle import texpose_oracle as O
    opt, camera, M, NeRF, RaySampler, FlexPatchSampler = MG._load_reference()
    torch.set_num_threads(4)
    T = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float32))
    rs = np.random.RandomState(2001)
    out = {}
    # ------------------------------------------------------------------ (a) convert_NDC on eval and train rays
    B, H, W, p = 2, 12, 16, 4
    sc = MG._scene(B, H, W, seed=20)
    opt.H, opt.W = H, W
    c, r = camera.get_center_and_ray(opt, sc["pose"], intr=sc["intr"])
    cn, rn = camera.convert_NDC(opt, c, r, intr=sc["intr"])
    torch.manual_seed(5)
    coords = torch.rand(B, p, p, 2) * 1.8 - 0.9
    ct, rt = RaySampler.get_rays(opt, sc["intr"], coords, sc["pose"])
    ctn, rtn = camera.convert_NDC(opt, ct.view(B, p * p, 3), rt.view(B, p * p, 3), intr=sc["intr"])
    out.update({"a.H": H, "a.W": W, "a.intr": sc["intr"], "a.pose": sc["pose"], "a.center": c, "a.ray": r, "a.center_ndc": cn,
                "a.ray_ndc": rn, "a.coords": coords, "a.center_t": ct, "a.ray_t": rt, "a.center_t_ndc": ctn, "a.ray_t_ndc": rtn})
    # ------------------------------------------------------------------ (b) inverse depths
    Bd, R, N = 2, 9, 6
    opt.nerf.sample_intvs = N
    opt.nerf.depth.param = "inverse"
    near = T(rs.uniform(1 / 7.0, 1 / 6.0, size=(Bd, R)))
    far = near + T(rs.uniform(0.01, 0.05, size=(Bd, R)))
    opt.nerf.sample_stratified = False
    z_mid = M.Graph.sample_depth(opt, Bd, (near, far), num_rays=R)
    opt.nerf.sample_stratified = True
    torch.manual_seed(22)
    rand = torch.rand(Bd, R, N, 1)
    torch.manual_seed(22)
    z_str = M.Graph.sample_depth(opt, Bd, (near, far), num_rays=R)
    out.update({"b.near": near, "b.far": far, "b.N": N, "b.z_mid": z_mid, "b.rand": rand, "b.z_strat": z_str})
    opt.nerf.depth.param = "metric"
    # ------------------------------------------------------------------ (c) renders
    B, H, W, p, N = 2, 16, 16, 4, 8
    n_train, seed_w = 5, 7
    opt.H, opt.W, opt.batch_size, opt.patch_size = H, W, B, 16
    opt.nerf.sample_intvs = N
    opt.nerf.rand_rays = 48
    opt.data.image_size = [H, W]
    opt.arch.posenc.L_3D, opt.arch.posenc.L_view = 10, 4
    sc = MG._scene(B, H, W, seed=8)
    g = M.Graph(opt)
    sd = g.nerf.state_dict()
    g.nerf.load_state_dict({**sd, **O.make_params(seed_w)})
    g.latent_vars_trans = torch.nn.Embedding(n_train, 16)
    g.latent_vars_light = torch.nn.Embedding(n_train, 48)
    ers = np.random.RandomState(77)
    with torch.no_grad():
        g.latent_vars_trans.weight.copy_(T(ers.normal(size=(n_train, 16))))
        g.latent_vars_light.weight.copy_(T(ers.normal(size=(n_train, 48))))
    torch.manual_seed(31)
    coords = torch.rand(B, p, p, 2) * 1.6 - 0.8
    idx = torch.tensor([3, 1])
    keep = ("rgb", "rgb_static", "rgb_transient", "opacity", "opacity_static", "opacity_transient", "uncert", "depth", "alpha_static",
            "alpha_transient", "density")
    out.update({"c.H": H, "c.W": W, "c.N": N, "c.seed": seed_w, "c.n_train": n_train, "c.emb_seed": 77, "c.intr": sc["intr"],
                "c.pose": sc["pose"], "c.coords": coords, "c.sample_idx": idx})
    # ndc, train mode: the ranges are fractions of the NDC depth axis (the object sits at metric z ~ 6: t = 1 - 1 / z ~ 0.83)
    zn = T(rs.uniform(0.76, 0.8, size=(B, H * W)))
    zf = zn + T(rs.uniform(0.06, 0.12, size=(B, H * W)))
    opt.camera.ndc = True
    torch.manual_seed(41)
    rand = torch.rand(B, p * p, N, 1)
    torch.manual_seed(41)
    ret = g.render(opt, sc["pose"], intr=sc["intr"], ray_idx=coords, depth_range=(zn[:, :, None], zf[:, :, None]), sample_idx=idx, mode="train")
    crs = np.random.RandomState(56)
    cot = {k: T(crs.normal(size=tuple(ret[k].shape))) for k in ("rgb", "rgb_static", "rgb_transient", "uncert", "depth", "density")}
    sum((ret[k] * cot[k]).sum() for k in cot).backward()
    grads = {"stride": G13.STRIDE}
    for name in ("mlp_rgb", "mlp_trans"):
        for li, layer in enumerate(getattr(g.nerf, name)):
            G13.pack(f"c.ndc.g.{name}.{li}.weight", layer.weight.grad, grads)
            G13.pack(f"c.ndc.g.{name}.{li}.bias", layer.bias.grad, grads)
    grads["c.ndc.g.latent_vars_trans"] = g.latent_vars_trans.weight.grad.clone()
    grads["c.ndc.g.latent_vars_light"] = g.latent_vars_light.weight.grad.clone()
    out.update({"c.ndc.z_near": zn, "c.ndc.z_far": zf, "c.ndc.rand": rand, **{"c.ndc.out_" + k: ret[k] for k in keep},
                **{"c.ndc.cot_" + k: v for k, v in cot.items()}, **grads})
    opt.camera.ndc = False
    # inverse depths, val mode over every pixel (unstratified): ranges are reciprocals of the scene's metric ranges
    opt.nerf.depth.param = "inverse"
    opt.nerf.sample_stratified = False
    sc1 = MG._scene(1, H, W, seed=9)
    zi_n, zi_f = 1 / sc1["z_far"], 1 / sc1["z_near"]
    every = torch.arange(H * W)[None]
    with torch.no_grad():
        val = g.render(opt, sc1["pose"], intr=sc1["intr"], ray_idx=every, depth_range=(zi_n[:, :, None], zi_f[:, :, None]), sample_idx=None, mode="val")
    out.update({"c.inv.intr": sc1["intr"], "c.inv.pose": sc1["pose"], "c.inv.z_near": zi_n, "c.inv.z_far": zi_f,
                **{"c.inv.out_" + k: val[k] for k in keep}})
    # both on, train mode, the stored draw
    opt.camera.ndc = True
    opt.nerf.sample_stratified = True
    zb_n = T(rs.uniform(1.05, 1.1, size=(B, H * W)))             # 1 / (t + 1e-8) in 0.83 ... 0.95
    zb_f = zb_n + T(rs.uniform(0.05, 0.1, size=(B, H * W)))
    torch.manual_seed(43)
    rand_b = torch.rand(B, p * p, N, 1)
    torch.manual_seed(43)
    with torch.no_grad():
        both = g.render(opt, sc["pose"], intr=sc["intr"], ray_idx=coords, depth_range=(zb_n[:, :, None], zb_f[:, :, None]), sample_idx=idx, mode="train")
    out.update({"c.both.z_near": zb_n, "c.both.z_far": zb_f, "c.both.rand": rand_b, **{"c.both.out_" + k: both[k] for k in keep}})
    opt.camera.ndc = False
    opt.nerf.depth.param = "metric"
    MG._save("g20_ndc_inverse", **out)


if __name__ == "__main__":
    main()
