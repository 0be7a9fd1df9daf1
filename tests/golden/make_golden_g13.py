#!/usr/bin/env python3
"""G13: two full GAN training iterations of the REAL reference (nerf_trainstep + disc_trainstep of
model/nerf_adapt_st_gan.py:108-171) on CPU -> gradients and parameter deltas.

    python tests/golden/make_golden_g13.py          (build container only; needs /root/reference)

Inputs are recipes (seeds) both sides can regenerate: synthetic batch = texpose_amd.synthetic.training_batch,
NeRF weights = oracle make_params, discriminator = oracle seed_spectral_module, embeddings = RandomState.
Random draws inside an iteration are pinned: patch coordinates / scales are stored (get_ray_idx is skipped on both
sides), the stratified jitter tensor is the first torch.rand draw of nerf_trainstep and is stored.
The feature loss is off (VGG weights unavailable); everything else is the reference default.
Big tensors (discriminator convs) are stored as a strided subsample + L2 norm.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG                                           # noqa: E402  (stubs + reference loader)

B, H, W, P, N, N_TRAIN = 2, 32, 32, 16, 8, 5
SEED_W, SEED_D, SEED_E, SEED_B = 11, 654, 78, 3
STRIDE = 211                                                        # subsample stride for tensors > 4096 elements


def pack(name, t, out):
    t = t.detach().reshape(-1).double()
    out[name + ".norm"] = np.float64(t.norm().item())
    if t.numel() > 4096:
        out[name + ".sub"] = t[::STRIDE].float().numpy()
    else:
        out[name] = t.float().numpy()


def main():
    from oracle import texpose_oracle as O
    from texpose_amd.synthetic import training_batch
    opt, camera, M, NeRF, RaySampler, FlexPatchSampler = MG._load_reference()
    torch.set_num_threads(4)
    opt.H, opt.W, opt.batch_size, opt.patch_size = H, W, B, P
    opt.data.image_size = [H, W]
    opt.nerf.sample_intvs = N
    opt.loss_weight.feat = None
    opt.max_iter = 1000
    T = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float32))

    mdl = M.Model.__new__(M.Model)
    g = M.Graph(opt)
    sd = g.nerf.state_dict()
    g.nerf.load_state_dict({**sd, **O.make_params(SEED_W)})
    O.seed_spectral_module(g.discriminator, SEED_D)
    g.latent_vars_trans = torch.nn.Embedding(N_TRAIN, 16)
    g.latent_vars_light = torch.nn.Embedding(N_TRAIN, 48)
    ers = np.random.RandomState(SEED_E)
    with torch.no_grad():
        g.latent_vars_trans.weight.copy_(T(ers.normal(size=(N_TRAIN, 16))))
        g.latent_vars_light.weight.copy_(T(ers.normal(size=(N_TRAIN, 48))))
    # the reference freezes the trunk when it restores the pretrain checkpoint (util.py); same effect here
    for q in g.nerf.mlp_feat.parameters():
        q.requires_grad_(False)
    mdl.graph = g
    g.train()
    mdl.setup_optimizer(opt)

    batch = training_batch(B, H, W, n_train=N_TRAIN, seed=SEED_B, device="cpu")
    p0 = {k: v.detach().clone() for k, v in g.state_dict().items()}
    out = dict(B=B, H=H, W=W, P=P, N=N, n_train=N_TRAIN, seed_w=SEED_W, seed_d=SEED_D, seed_e=SEED_E, seed_b=SEED_B,
               stride=STRIDE, lr=opt.optim.lr, lr_disc=opt.optim_disc.lr)
    crs = torch.Generator().manual_seed(5)
    for it in range(2):
        var = MG._AttrDict({k: v.clone() for k, v in batch.items()})
        s = 0.5 + 0.4 * torch.rand(B, 1, 1, 1, generator=crs)
        lin = torch.linspace(-1, 1, P)
        yy, xx = torch.meshgrid(lin, lin, indexing="ij")
        shift = (1 - s) * (torch.rand(B, 1, 1, 2, generator=crs) * 2 - 1)
        var.ray_idx = torch.stack([xx, yy], -1)[None] * s + shift
        var.ray_scales = s.clone()
        torch.manual_seed(100 + it)
        rand = torch.rand(B, P * P, N, 1)
        torch.manual_seed(100 + it)
        var, gloss = mdl.nerf_trainstep(opt, var)
        for name, q in g.named_parameters():                       # generator-step gradients (heads, embeddings)
            if not name.startswith("discriminator") and q.grad is not None:
                pack(f"it{it}.grad.{name}", q.grad, out)
        var, dloss = mdl.disc_trainstep(opt, var)
        out[f"it{it}.ray_idx"] = var.ray_idx.detach().numpy()
        out[f"it{it}.ray_scales"] = var.ray_scales.detach().numpy()
        out[f"it{it}.rand"] = rand.numpy()
        for k in ("render", "uncert", "trans_reg", "gan_nerf", "all"):
            out[f"it{it}.gloss.{k}"] = np.float64(gloss[k].item())
        for k in ("gan_disc_real", "gan_disc_fake", "gan_reg_real"):
            out[f"it{it}.dloss.{k}"] = np.float64(dloss[k].item())
        out[f"it{it}.rgb"] = var.rgb.detach().numpy()
        out[f"it{it}.d_real"] = var.d_real_disc.detach().numpy()
        out[f"it{it}.d_fake"] = var.d_fake_disc.detach().numpy()
        for name, q in g.named_parameters():                       # discriminator-step gradients (real + R1 + fake)
            if name.startswith("discriminator") and q.grad is not None:
                pack(f"it{it}.grad.{name}", q.grad, out)
    assert all(q.grad is None for q in g.nerf.mlp_feat.parameters())
    for k, v in g.state_dict().items():
        if not torch.equal(v, p0[k]):
            pack("delta." + k, v.double() - p0[k].double(), out)
        else:
            out["unchanged." + k] = np.int8(1)
    path = os.path.join(HERE, "g13_train_iterations.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", len(out), "entries")
    print({k: float(out[k]) for k in out if ".gloss." in k or ".dloss." in k})


if __name__ == "__main__":
    main()
