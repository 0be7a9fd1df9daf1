#!/usr/bin/env python3
"""G19: option values of hot-path rows a10 / a11 / f1 that the reference's shipped yaml leaves empty but its code implements:
  (a) coarse-to-fine weights of the positional encodings (layers/nerf_static_transient_light.py:217-234; `c2f.range`, `c2f.start`,
      `NeRF.progress`): forward + head gradients of NeRF.forward at three progress values (before, inside and behind the window);
  (b) `nerf.density_noise_reg` (:96-97): Gaussian noise on the static density's pre-activation in train mode (the noise tensor is the
      first draw of the forward and is stored);
  (c) the discriminator's geometry encodings (layers/discriminator.py:117-141,145-168; `gan.L_nocs`, `gan.L_normal`, `gan.geo_c2f`,
      `Discriminator.progress`): logits, weight gradients, the R1-style input gradient, u / v after the pass.

    python tests/golden/make_golden_g19_options.py         (build container only; needs /root/reference)

Weights are recipes (oracle make_params / seed_spectral_module); inputs and expected outputs are stored.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG                                           # noqa: E402
import make_golden_g9b as G9B                                      # noqa: E402
import make_golden_g13 as G13                                      # noqa: E402  (pack: strided subsample + norm of big tensors)


def main():
    from oracle import texpose_oracle as O
    opt, camera, M, NeRF, RaySampler, FlexPatchSampler = MG._load_reference()
    import layers.discriminator as RD
    torch.set_num_threads(4)
    T = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float32))
    rs = np.random.RandomState(1901)
    out = {}
    # ------------------------------------------------------------------ (a) c2f
    opt.arch.posenc.L_3D, opt.arch.posenc.L_view = 10, 4
    opt.c2f.range, opt.c2f.start = [0.1, 0.5], 2
    nerf = NeRF(opt)
    seed_w = 31
    sd = nerf.state_dict()
    nerf.load_state_dict({**sd, **O.make_params(seed_w)})
    for q in nerf.mlp_feat.parameters():
        q.requires_grad_(False)
    B, R, N = 2, 24, 8
    pts = T(rs.uniform(-1.2, 1.2, size=(B, R, N, 3)))
    unit = torch.nn.functional.normalize(T(rs.normal(size=(B, R, 1, 3))), dim=-1).expand(B, R, N, 3).contiguous()
    lt, ll = T(rs.normal(size=(B, 16))), T(rs.normal(size=(B, 48)))
    cots = [T(rs.normal(size=(B, R, N, 3, 2))), T(rs.normal(size=(B, R, N, 2))), T(rs.normal(size=(B, R, N, 1)))]
    out.update({"a.seed_w": seed_w, "a.range": np.array(opt.c2f.range, np.float32), "a.start": opt.c2f.start, "a.points": pts,
                "a.ray_unit": unit, "a.lat_trans": lt, "a.lat_light": ll})
    x = T(rs.uniform(-3, 3, size=(5, 3)))
    out["a.enc_x"] = x
    for tag, progress in (("p005", 0.05), ("p027", 0.27), ("p100", 1.0)):
        nerf.progress.data.fill_(progress)
        out[f"a.{tag}.progress"] = np.float32(progress)
        out[f"a.{tag}.enc10"] = nerf.positional_encoding(opt, x, L=10, c2f=True)
        out[f"a.{tag}.enc4"] = nerf.positional_encoding(opt, x, L=4, c2f=True)
        ltq, llq = lt.clone().requires_grad_(), ll.clone().requires_grad_()
        for q in nerf.parameters():
            q.grad = None
        taps, handles = G9B._hook_preactivations(nerf)
        rgb, den, unc = nerf.forward(opt, pts, ray_unit=unit, latent_variable_trans=ltq, latent_variable_light=llq, mode="val")
        for h in handles:
            h.remove()
        keep = (~G9B._risky(taps, (B, R, N))).float()                  # flip-free cotangents, as in G9c (b)
        ck = [c * keep.view(B, R, N, *([1] * (c.dim() - 3))) for c in cots]
        sum((o * c).sum() for o, c in zip((rgb, den, unc), ck)).backward()
        out.update({f"a.{tag}.rgb": rgb, f"a.{tag}.density": den, f"a.{tag}.uncert": unc})
        if tag == "p027":                              # gradients at the progress value inside the window (big tensors: subsample + norm)
            out.update({f"a.{tag}.keep": keep, f"a.{tag}.g.lat_t": ltq.grad, f"a.{tag}.g.lat_l": llq.grad})
            for k, c in zip(("rgb", "density", "uncert"), ck):
                out[f"a.{tag}.cot_{k}"] = c
            for name in ("mlp_rgb", "mlp_trans"):
                for li, layer in enumerate(getattr(nerf, name)):
                    G13.pack(f"a.{tag}.g.{name}.{li}.weight", layer.weight.grad, out)
                    G13.pack(f"a.{tag}.g.{name}.{li}.bias", layer.bias.grad, out)
            # the view-encoding columns of mlp_rgb.0 in full: the gradient entries the c2f weights scale
            out[f"a.{tag}.g.mlp_rgb.0.weight.viewenc"] = nerf.mlp_rgb[0].weight.grad[:, 256:256 + 27].clone()
        print("G19a", tag, "kept samples:", int(keep.sum()), "of", keep.numel())
    # ------------------------------------------------------------------ (b) density noise
    opt.c2f.range, opt.c2f.start = None, None
    opt.nerf.density_noise_reg = 0.3
    nerf_b = NeRF(opt)
    nerf_b.load_state_dict({**nerf_b.state_dict(), **O.make_params(seed_w)})
    torch.manual_seed(77)
    noise = torch.randn(B, R, N)
    torch.manual_seed(77)
    rgb, den, unc = nerf_b.forward(opt, pts, ray_unit=unit, latent_variable_trans=lt, latent_variable_light=ll, mode="train")
    rgb_v, den_v, _ = nerf_b.forward(opt, pts, ray_unit=unit, latent_variable_trans=lt, latent_variable_light=ll, mode="val")
    assert not torch.equal(den[..., 0], den_v[..., 0]) and torch.equal(den[..., 1], den_v[..., 1]) and torch.equal(rgb, rgb_v)
    out.update({"b.reg": np.float32(0.3), "b.noise": noise, "b.density_train": den, "b.density_val": den_v, "b.rgb": rgb, "b.uncert": unc})
    opt.nerf.density_noise_reg = None
    # ------------------------------------------------------------------ (c) discriminator geometry encodings
    opt.patch_size = 16
    opt.gan.L_nocs, opt.gan.L_normal, opt.gan.geo_c2f = 2, 2, [0.1, 0.5]
    disc = RD.Discriminator(opt)
    O.seed_spectral_module(disc, 440)
    disc.train()
    Bd = 3
    xin = T(rs.uniform(0, 1, size=(Bd, 9, 16, 16)))
    xin[:, 3:] = xin[:, 3:] * 2 - 1                                          # nocs / normal channels in [-1, 1]
    scale = T(rs.uniform(0.3, 1.0, size=(Bd, 1, 1, 1)))
    out.update({"c.seed_d": 440, "c.L": 2, "stride": G13.STRIDE, "c.range": np.array([0.1, 0.5], np.float32), "c.x": xin, "c.scale": scale})
    for name, buf in disc.state_dict().items():
        if name.endswith(("weight_u", "weight_v")):
            out["c.in." + name] = buf.clone()
    for tag, progress in (("p030", 0.3), ("p100", 1.0)):
        state = {k: v.clone() for k, v in disc.state_dict().items()}
        disc.progress.data.fill_(progress)
        for q in disc.parameters():
            q.grad = None
        xq = xin.clone().requires_grad_()
        logits = disc(opt, xq, scale)
        (gx,) = torch.autograd.grad(logits.sum(), xq, create_graph=True)      # the R1 penalty's first pass (model :794-807)
        reg = gx.pow(2).reshape(Bd, -1).sum(1).mean()
        (torch.nn.functional.binary_cross_entropy_with_logits(logits, torch.ones_like(logits)) + 10.0 * reg).backward()
        out.update({f"c.{tag}.progress": np.float32(progress), f"c.{tag}.logits": logits, f"c.{tag}.gx": gx, f"c.{tag}.reg": reg})
        for name, q in disc.named_parameters():
            if q.grad is not None:
                G13.pack(f"c.{tag}.g.{name}", q.grad, out)
        for name, buf in disc.state_dict().items():
            if name.endswith(("weight_u", "weight_v")):
                out[f"c.{tag}.out.{name}"] = buf.clone()
        disc.load_state_dict(state)
    assert out["c.p030.logits"].shape == (Bd,) and disc.main[0].weight_orig.shape[1] == 9 + 24
    MG._save("g19_options", **out)


if __name__ == "__main__":
    main()
