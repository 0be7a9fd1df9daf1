#!/usr/bin/env python3
"""G13c: the GENERATOR step of iteration 0 of the REAL reference (model/nerf_adapt_st_gan.py:108-127 `nerf_trainstep`: render,
:464-514 `nerf_forward` incl. the pass through the frozen discriminator, :747-776 the loss terms, model/base.py:145-157 the weighted
total, backward, Adam) with everything it consumes stored as INPUTS -- the G9b / G13b trick applied to the nerf step, so that a test
pins it at the tight tier (G13 compares the step behind this repo's own ray generation, whose rays differ from the reference's in the
last bit; its bound there is 1e-2 .. 6e-2).

    python tests/golden/make_golden_g13c.py         (build container only; needs /root/reference)

Same recipes and seeds as G13 / G13b.  Stored as INPUTS: the rays, depth samples and latent rows the reference's render fed to
NeRF.forward_samples (captured by wrapping it), the patch coordinates and scales, the discriminator's weight_u / weight_v as the
step finds them, the batch recipe seed.  As OUTPUTS:
  * the render (rgb, uncert, per-sample density), D(fake) of the nerf step, the loss terms and their weighted total;
  * `raw.*`: the gradients `gloss.all.backward()` left in the 16 head tensors and the two embedding tables;
  * `ff.*`: the same gradients "flip-free": every ray that holds a ReLU gate within 64 ulp (of its layer's largest pre-activation)
    of zero -- found from the REFERENCE's own pre-activations by forward hooks -- has its render outputs detached at the render
    boundary (value kept, no gradient path), in a second run of the reference's own nerf_forward / compute_loss / backward from the
    same state.  Another fp32 evaluation order may take the other side of such a gate; with those rays out of the gradient path the
    remaining difference is rounding, and the gradients are compared at 1e-4 instead of the gate-flip bound 5e-3.  `keep_ray`
    is the mask;
  * weight_u / weight_v after the step's one power iteration.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG                                           # noqa: E402
import make_golden_g13 as G13                                      # noqa: E402
import make_golden_g9b as G9B                                      # noqa: E402  (_hook_preactivations, _risky)

RAY_KEYS = ("rgb", "rgb_static", "rgb_transient", "opacity", "opacity_static", "opacity_transient", "uncert", "depth", "density")


def _grads(g):
    out = {}
    for name in ("mlp_rgb", "mlp_trans"):
        for li, layer in enumerate(getattr(g.nerf, name)):
            out[f"g.{name}.{li}.weight"] = layer.weight.grad.detach().clone()
            out[f"g.{name}.{li}.bias"] = layer.bias.grad.detach().clone()
    out["g.latent_vars_trans"] = g.latent_vars_trans.weight.grad.detach().clone()
    out["g.latent_vars_light"] = g.latent_vars_light.weight.grad.detach().clone()
    return out


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

    def fresh_var():
        var = MG._AttrDict({k: v.clone() for k, v in batch.items()})
        var.ray_idx, var.ray_scales = ray_idx.clone(), s.clone()
        return var

    s = 0.5 + 0.4 * torch.rand(B, 1, 1, 1, generator=crs)
    lin = torch.linspace(-1, 1, P)
    yy, xx = torch.meshgrid(lin, lin, indexing="ij")
    shift = (1 - s) * (torch.rand(B, 1, 1, 2, generator=crs) * 2 - 1)
    ray_idx = torch.stack([xx, yy], -1)[None] * s + shift
    state0 = {k: v.detach().clone() for k, v in g.state_dict().items()}
    out = dict(B=B, H=H, W=W, P=P, N=N, n_train=N_TRAIN, seed_w=G13.SEED_W, seed_d=G13.SEED_D, seed_e=G13.SEED_E, seed_b=G13.SEED_B,
               ray_idx=ray_idx.numpy(), ray_scales=s.numpy())
    for name, buf in state0.items():
        if name.endswith("weight_u") or name.endswith("weight_v"):
            out["in." + name] = buf.numpy()
    # ---- the reference's own step, its render inputs and gate pre-activations captured on the way
    seen = []
    orig_fs = g.nerf.forward_samples

    def spy(opt_, center, ray, depth_samples, latent_variable_trans=None, latent_variable_light=None, mode=None):
        seen.append(dict(center=center.detach().clone(), ray=ray.detach().clone(), depth=depth_samples.detach().clone(),
                         lat_t=latent_variable_trans.detach().clone(), lat_l=latent_variable_light.detach().clone()))
        return orig_fs(opt_, center, ray, depth_samples, latent_variable_trans=latent_variable_trans,
                       latent_variable_light=latent_variable_light, mode=mode)

    g.nerf.forward_samples = spy
    taps, handles = G9B._hook_preactivations(g.nerf)
    torch.manual_seed(100)
    var, gloss = mdl.nerf_trainstep(opt, fresh_var())
    g.nerf.forward_samples = orig_fs
    assert len(seen) == 1
    risky_ray = G9B._risky(taps, (B, P * P, N)).any(dim=-1)                        # [B,R]
    for h in handles:
        h.remove()
    keep = (~risky_ray).float()
    print("G13c: %d of %d rays hold a gate-flip candidate" % (int(risky_ray.sum()), risky_ray.numel()))
    assert 0 < int(risky_ray.sum()) < 0.8 * risky_ray.numel()
    out.update({"in." + k: v.numpy() for k, v in seen[0].items()})
    out["sample_idx"] = var.idx.numpy()
    for k in ("rgb", "uncert", "density", "depth", "opacity"):
        out["out." + k] = var[k].detach().numpy()
    out["out.d_fake_nerf"] = var.d_fake_nerf.detach().numpy()
    for k in ("render", "uncert", "trans_reg", "gan_nerf", "all"):
        out["gloss." + k] = np.float64(gloss[k].item())
    for k in ("render", "uncert", "trans_reg", "gan_nerf"):
        out["w." + k] = np.float64(10 ** float(opt.loss_weight[k]))
    out.update({"raw." + k: v.numpy() for k, v in _grads(g).items()})
    for name, buf in g.discriminator.state_dict().items():
        if name.endswith("weight_u") or name.endswith("weight_v"):
            out["out." + name] = buf.detach().clone().numpy()
    out["keep_ray"] = keep.numpy()
    # ---- flip-free: the same state, the same draws; the candidate rays' render outputs carry no gradient
    g.load_state_dict(state0)
    orig_render = g.render

    def masked_render(opt_, pose, **kw):
        ret = orig_render(opt_, pose, **kw)
        for k in RAY_KEYS:
            if ret.get(k) is not None:
                m = keep.view(B, P * P, *([1] * (ret[k].dim() - 2)))
                ret[k] = m * ret[k] + (1 - m) * ret[k].detach()
        return ret

    g.render = masked_render
    mdl.toggle_grad(g.discriminator, False)
    for q in list(g.nerf.parameters()) + [g.latent_vars_trans.weight, g.latent_vars_light.weight]:
        q.grad = None
    torch.manual_seed(100)
    var2 = g.nerf_forward(opt, fresh_var(), mode="train")
    gloss2 = g.compute_loss(opt, var2, mode="train", train_step="nerf")
    gloss2 = mdl.summarize_loss(opt, var2, gloss2)
    assert np.array_equal(var2.rgb.detach().numpy(), out["out.rgb"]) and float(gloss2.all) == float(out["gloss.all"])
    gloss2.all.backward()
    g.render = orig_render
    out.update({"ff." + k: v.numpy() for k, v in _grads(g).items()})
    path = os.path.join(HERE, "g13c_nerf_step.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", len(out), "entries;", {k: float(out[k]) for k in out if k.startswith("gloss.")})
    rel = lambda a, b: float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))
    print("raw vs flip-free gradients (how much the masked rays carry):",
          {k[4:]: "%.2f" % rel(out[k], out["ff." + k[4:]]) for k in out if k.startswith("raw.g.") and k.endswith("weight")})


if __name__ == "__main__":
    main()
