#!/usr/bin/env python3
"""G13b: the DISCRIMINATOR step of iteration 0 of the REAL reference (model/nerf_adapt_st_gan.py:129-171: D(real) + BCE, the R1
penalty of compute_grad2 :794-807 -- a double backward --, D(fake) + BCE) with everything it consumes stored as INPUTS, so that
a test can run the step WITHOUT a render in the loop (G13 compares the step behind this repo's own render, whose rays differ
from the reference's in the last bit: its bound there is 1e-2; here the inputs are identical and the bound is 1e-4).

    python tests/golden/make_golden_g13b.py         (build container only; needs /root/reference)

Same recipes and seeds as G13 (tests/golden/make_golden_g13.py).  The reference's nerf_trainstep of iteration 0 runs first (it
advances the discriminator's power-iteration vectors once and produces the render); then stored as INPUTS of the step:
  rgb [B,P*P,3] (the reference's render), ray_idx, ray_scales, the discriminator's weight_u / weight_v as the step finds them
  (weight_orig is the seeded recipe, untouched by the nerf step), the batch recipe seed;
and as OUTPUTS: patch_real / patch_fake, the logits, the three loss values (R1 unweighted and weighted), the gradients of all six
weight_orig after the three backward calls (strided subsample + L2 norm for the large ones, as in G13), weight_u / weight_v after
the step's two power iterations.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG                                           # noqa: E402
import make_golden_g13 as G13                                      # noqa: E402


def main():
    from oracle import texpose_oracle as O
    from texpose_amd.synthetic import training_batch
    opt, camera, M, NeRF, RaySampler, FlexPatchSampler = MG._load_reference()
    torch.set_num_threads(4)
    B, H, W, P, N, N_TRAIN = G13.B, G13.H, G13.W, G13.P, G13.N, G13.N_TRAIN
    opt.H, opt.W, opt.batch_size, opt.patch_size = H, W, B, P
    opt.data.image_size = [H, W]
    opt.nerf.sample_intvs = N
    opt.loss_weight.feat = None
    opt.max_iter = 1000
    T = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float32))
    mdl = M.Model.__new__(M.Model)
    g = M.Graph(opt)
    g.nerf.load_state_dict({**g.nerf.state_dict(), **O.make_params(G13.SEED_W)})
    O.seed_spectral_module(g.discriminator, G13.SEED_D)
    g.latent_vars_trans = torch.nn.Embedding(N_TRAIN, 16)
    g.latent_vars_light = torch.nn.Embedding(N_TRAIN, 48)
    ers = np.random.RandomState(G13.SEED_E)
    with torch.no_grad():
        g.latent_vars_trans.weight.copy_(T(ers.normal(size=(N_TRAIN, 16))))
        g.latent_vars_light.weight.copy_(T(ers.normal(size=(N_TRAIN, 48))))
    for q in g.nerf.mlp_feat.parameters():
        q.requires_grad_(False)
    mdl.graph = g
    g.train()
    mdl.setup_optimizer(opt)
    batch = training_batch(B, H, W, n_train=N_TRAIN, seed=G13.SEED_B, device="cpu")
    crs = torch.Generator().manual_seed(5)
    var = MG._AttrDict({k: v.clone() for k, v in batch.items()})
    s = 0.5 + 0.4 * torch.rand(B, 1, 1, 1, generator=crs)
    lin = torch.linspace(-1, 1, P)
    yy, xx = torch.meshgrid(lin, lin, indexing="ij")
    shift = (1 - s) * (torch.rand(B, 1, 1, 2, generator=crs) * 2 - 1)
    var.ray_idx = torch.stack([xx, yy], -1)[None] * s + shift
    var.ray_scales = s.clone()
    torch.manual_seed(100)
    var, gloss = mdl.nerf_trainstep(opt, var)
    out = dict(B=B, H=H, W=W, P=P, N=N, n_train=N_TRAIN, seed_d=G13.SEED_D, seed_b=G13.SEED_B, stride=G13.STRIDE,
               rgb=var.rgb.detach().numpy(), ray_idx=var.ray_idx.detach().numpy(), ray_scales=var.ray_scales.detach().numpy())
    w0 = {}
    for name, buf in g.discriminator.state_dict().items():
        if name.endswith("weight_u") or name.endswith("weight_v"):
            out["in." + name] = buf.detach().clone().numpy()
        if name.endswith("weight_orig"):
            w0[name] = buf.detach().clone()
    # the disc step proper, unrolled exactly as the reference runs it (so that the unweighted R1 value can be stored as well)
    var, dloss = mdl.disc_trainstep(opt, var)
    for name, buf in g.discriminator.state_dict().items():
        if name.endswith("weight_orig"):
            assert not torch.equal(buf, w0[name])                 # RMSprop stepped
        if name.endswith("weight_u") or name.endswith("weight_v"):
            out["out." + name] = buf.detach().clone().numpy()
    out["patch_real"] = var.patch_real.detach().numpy()
    out["patch_fake"] = var.patch_fake.detach().numpy()
    out["d_real"] = var.d_real_disc.detach().numpy()
    out["d_fake"] = var.d_fake_disc.detach().numpy()
    for k in ("gan_disc_real", "gan_disc_fake", "gan_reg_real"):
        out["dloss." + k] = np.float64(dloss[k].item())           # (as the reference leaves them: each scaled by 10^w in place)
    out["w.gan_disc_real"], out["w.gan_disc_fake"], out["w.gan_reg_real"] = (np.float64(10 ** float(opt.loss_weight[k]))
                                                                             for k in ("gan_disc_real", "gan_disc_fake", "gan_reg_real"))
    n = 0
    for name, q in g.discriminator.named_parameters():
        if q.grad is not None:
            G13.pack("grad." + name, q.grad, out)
            n += 1
    assert n == 6
    path = os.path.join(HERE, "g13b_disc_step.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", len(out), "entries;",
          {k: float(out[k]) for k in out if k.startswith("dloss.")})


if __name__ == "__main__":
    main()
