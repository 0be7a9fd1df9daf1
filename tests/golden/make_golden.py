#!/usr/bin/env python3
"""Generate golden vectors by importing the REAL reference in the build container.

Run here (never on the GPU box; /root/reference does not exist there):

    python tests/golden/make_golden.py

It registers stub modules for the third-party packages the reference imports at
module scope but that the hot path never calls (easydict, torchvision, cv2,
pytorch3d, lpips, kornia, ...), imports the reference
modules from /root/reference, drives the hot-path functions with seeded inputs
and writes inputs + expected outputs to tests/golden/*.npz.  Only data is
committed; no reference source travels.

Weights for full-width MLP cases come from ``oracle.texpose_oracle.make_params``
(a numpy RandomState recipe) loaded into the reference's own ``NeRF`` module,
so the fixtures store seeds + outputs rather than 3.6 MB of weights.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("TEXPOSE_REFERENCE", "/root/reference")
sys.path.insert(0, REPO)


# --------------------------------------------------------------------------- stubs
class _AttrDict(dict):
    """Minimal recursive attribute dict standing in for easydict.EasyDict."""

    def __init__(self, d=None, **kw):
        super().__init__()
        d = dict(d or {}, **kw)
        for k, v in d.items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, _AttrDict):
            v = _AttrDict(v)
        super().__setitem__(k, v)

    __setattr__ = __setitem__

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def update(self, *a, **kw):
        for k, v in dict(*a, **kw).items():
            self[k] = v


def _install_stubs():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    mod("easydict", EasyDict=_AttrDict)
    mod("ipdb", set_trace=lambda *a, **k: None)
    mod("termcolor", colored=lambda s, *a, **k: str(s))
    cv2 = mod("cv2", setNumThreads=lambda n: None, INTER_LINEAR=1)
    cv2.ocl = types.SimpleNamespace(setUseOpenCL=lambda b: None)
    mod("imageio")
    mod("visdom")
    tv = mod("torchvision")
    tvt = mod("torchvision.transforms", Normalize=lambda mean, std: (lambda x: x))
    tvf = mod("torchvision.transforms.functional")
    tvm = mod("torchvision.models",
              vgg19=lambda pretrained=True: types.SimpleNamespace(features=torch.nn.Sequential()))
    tv.transforms, tv.models = tvt, tvm
    tvt.functional = tvf
    mod("pytorch3d")
    mod("pytorch3d.ops")
    mod("pytorch3d.ops.knn", knn_gather=None, knn_points=None)
    mod("pytorch3d.structures")
    mod("pytorch3d.structures.pointclouds", Pointclouds=object)
    mod("pytorch3d.loss")
    mod("pytorch3d.loss.chamfer", _validate_chamfer_reduction_inputs=None, _handle_pointcloud_input=None)
    mod("lpips", LPIPS=lambda net="alex": torch.nn.Identity())
    mod("kornia")
    mod("kornia.color", rgb_to_lab=lambda x: x)
    tb = mod("torch.utils.tensorboard", SummaryWriter=object)
    torch.utils.tensorboard = tb
    tools = mod("tools")
    tools.__path__ = [os.path.join(REF, "tools")]       # bypass tools/__init__.py (PyTorch3D renderer)
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self


def _load_reference():
    _install_stubs()
    sys.path.insert(0, REF)
    os.chdir(REF)
    import options                                             # noqa
    opt = options.load_options("options/nerf_lm_adapt_gan.yaml")
    opt.model = "nerf_adapt_st_gan"
    opt.device = "cpu"
    opt.name = None
    import camera                                              # noqa
    import model.nerf_adapt_st_gan as M                        # noqa
    from layers.nerf_static_transient_light import NeRF        # noqa
    from tools.ray_sampler import RaySampler                   # noqa
    from tools.patch_sampler import FlexPatchSampler           # noqa
    return opt, camera, M, NeRF, RaySampler, FlexPatchSampler


def _np(d):
    out = {}
    for k, v in d.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    return out


def _save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    arrays = _np(arrays)
    np.savez_compressed(path, **arrays)
    print("wrote", path, {k: tuple(v.shape) for k, v in arrays.items()})


def _scene(B, H, W, seed):
    from oracle import texpose_oracle as O
    sc = O.synthetic_scene(H, W, B=B, seed=seed)
    # crop-like intrinsics for small images: keep the object in view
    K = np.array(O.LINEMOD_K, dtype=np.float32)
    K[0, 0] = K[1, 1] = 700.0 * H / 128.0
    K[0, 2], K[1, 2] = W / 2.0 - 0.3, H / 2.0 + 0.2
    sc["intr"] = torch.from_numpy(np.tile(K[None], (B, 1, 1)))
    return sc


def main():
    from oracle import texpose_oracle as O
    opt, camera, M, NeRF, RaySampler, FlexPatchSampler = _load_reference()
    torch.set_num_threads(4)
    rs = np.random.RandomState(1234)
    T = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float32))

    # ---------------------------------------------------------------- G0 patch sampler (a1)
    ps = FlexPatchSampler(opt, scale_anneal=0.0002)            # same positional-arg slip as the reference
    ps.iterations = 3000
    torch.manual_seed(11)
    u = torch.rand(3, 4, 1, 1, 1)                              # the three draws, in call order
    torch.manual_seed(11)
    coords, scales = ps(nbatch=4, patch_size=8, device="cpu")
    _save("g0_patch_sampler", iterations=3000, patch_size=8, u_scale=u[0].flatten(), u_hoff=u[1].flatten(),
          u_woff=u[2].flatten(), coords=coords, scales=scales, scales_curr=np.array(ps.scales_curr))

    # ---------------------------------------------------------------- G1 train rays + bounds (a2,a3)
    B, H, W, p = 3, 24, 32, 5
    sc = _scene(B, H, W, seed=5)
    opt.H, opt.W = H, W
    torch.manual_seed(3)
    coords = (torch.rand(B, p, p, 2) * 2.2 - 1.1)             # some coords fall outside [-1,1] (zero padding)
    coords[0, 0, 0] = torch.tensor([-1.0, -1.0])
    coords[0, 0, 1] = torch.tensor([1.0, 1.0])
    zn = T(rs.uniform(5, 7, size=(B, H * W)))
    zf = zn + T(rs.uniform(0.5, 2, size=(B, H * W)))
    center, ray = RaySampler.get_rays(opt, sc["intr"], coords, sc["pose"])
    zns, zfs = RaySampler.get_bounds(opt, coords, zn, zf)
    _save("g1_rays_train", H=H, W=W, intr=sc["intr"], pose=sc["pose"], coords=coords, z_near=zn, z_far=zf,
          center=center, ray=ray, z_near_s=zns, z_far_s=zfs)

    # ---------------------------------------------------------------- G2 eval rays + gather (a4,a5)
    B, H, W = 2, 6, 8
    sc = _scene(B, H, W, seed=6)
    opt.H, opt.W = H, W
    c, r = camera.get_center_and_ray(opt, sc["pose"], intr=sc["intr"])
    ray_idx = torch.from_numpy(rs.randint(0, H * W, size=(B, 7)).astype(np.int64))
    cg = M.Graph.ray_batch_sample(c, ray_idx)
    rg = M.Graph.ray_batch_sample(r, ray_idx)
    _save("g2_rays_eval", H=H, W=W, intr=sc["intr"], pose=sc["pose"], center=c, ray=r, ray_idx=ray_idx,
          center_g=cg, ray_g=rg)

    # ---------------------------------------------------------------- G3 AABB slab (a6)
    o = T(rs.uniform(-3, 3, size=(2, 40, 3)))
    d = T(rs.normal(size=(2, 40, 3)))
    o[0, 0] = T([0.1, 0.2, -0.1]); d[0, 0] = T([0.3, -0.2, 1.0])        # origin inside the box
    o[0, 1] = T([0.0, 0.0, -5.0]); d[0, 1] = T([0.0, 0.0, 1.0])         # axis-parallel hit (inf slabs)
    o[0, 2] = T([5.0, 0.0, -5.0]); d[0, 2] = T([0.0, 0.0, 1.0])         # axis-parallel miss
    o[0, 3] = T([0.0, 0.0, 5.0]); d[0, 3] = T([0.0, 0.1, 1.0])          # box behind the ray
    lo, hi = camera.enlarge_diagonal(T([[[-0.5, -0.4, -0.6]]]), T([[[0.5, 0.6, 0.4]]]))
    tn, tf, ok = camera.aabb_ray_intersection(lo, hi, o, d)
    _save("g3_aabb", aabb_min0=T([[[-0.5, -0.4, -0.6]]]), aabb_max0=T([[[0.5, 0.6, 0.4]]]), aabb_min=lo,
          aabb_max=hi, o=o, d=d, t_near=tn, t_far=tf, valid=ok.to(torch.uint8))

    # ---------------------------------------------------------------- G4 stratified depths (a7)
    B, R, N = 2, 9, 6
    opt.nerf.sample_intvs = N
    near = T(rs.uniform(5, 7, size=(B, R)))
    far = near + T(rs.uniform(0.5, 2, size=(B, R)))
    opt.nerf.sample_stratified = False
    z_mid = M.Graph.sample_depth(opt, B, (near, far), num_rays=R)
    opt.nerf.sample_stratified = True
    torch.manual_seed(21)
    rand = torch.rand(B, R, N, 1)
    torch.manual_seed(21)
    z_str = M.Graph.sample_depth(opt, B, (near, far), num_rays=R)
    _save("g4_sample_depth", near=near, far=far, N=N, z_mid=z_mid, rand=rand, z_strat=z_str)

    # ---------------------------------------------------------------- G5 positional encoding (a10)
    opt.arch.posenc.L_3D, opt.arch.posenc.L_view = 10, 4
    nerf = NeRF(opt)
    x = T(rs.uniform(-5, 5, size=(2, 3, 4, 3)))
    x[0, 0, 0] = T([4.99, -5.0, 0.0])
    x[0, 0, 1] = T([1.0, 0.5, 0.25])
    _save("g5_posenc", x=x, enc10=nerf.positional_encoding(opt, x, L=10, c2f=True),
          enc4=nerf.positional_encoding(opt, x, L=4, c2f=True))

    # ---------------------------------------------------------------- G6 full-width MLP forward (a11)
    def load_params(net, params):
        sd = net.state_dict()
        for k, v in params.items():
            assert sd[k].shape == v.shape, (k, sd[k].shape, v.shape)
        net.load_state_dict({**{k: v for k, v in sd.items() if k not in params}, **params})

    seed_w = 7
    params = O.make_params(seed_w)
    load_params(nerf, params)
    B, R, N = 2, 5, 4
    pts = T(rs.uniform(-1.2, 1.2, size=(B, R, N, 3)))
    unit = torch.nn.functional.normalize(T(rs.normal(size=(B, R, 1, 3))), dim=-1).expand(B, R, N, 3).contiguous()
    lt = T(rs.normal(size=(B, 16)))
    ll = T(rs.normal(size=(B, 48)))
    rgb, den, unc = nerf.forward(opt, pts, ray_unit=unit, latent_variable_trans=lt, latent_variable_light=ll,
                                 mode="train")
    _save("g6_mlp_full", seed=seed_w, points=pts, ray_unit=unit, lat_trans=lt, lat_light=ll,
          rgb=rgb, density=den, uncert=unc)

    # width-32 variant with stored weights (cheap cross-check of layer wiring)
    opt32 = _AttrDict({k: v for k, v in opt.items()})
    opt32.arch = _AttrDict(dict(opt.arch))
    opt32.arch.layers_feat = [None] + [32] * 8
    opt32.arch.layers_rgb = [None, 32, 32, 32, 3]
    opt32.arch.layers_trans = [None, 32, 32, 32, 5]
    nerf32 = NeRF(opt32)
    p32 = O.make_params(9, width=32)
    load_params(nerf32, p32)
    rgb32, den32, unc32 = nerf32.forward(opt32, pts, ray_unit=unit, latent_variable_trans=lt,
                                         latent_variable_light=ll, mode="train")
    _save("g6_mlp_w32", points=pts, ray_unit=unit, lat_trans=lt, lat_light=ll, rgb=rgb32, density=den32,
          uncert=unc32, **{"w." + k: v for k, v in p32.items()})

    # ---------------------------------------------------------------- G7 composite (a13)
    B, R, N = 2, 11, 16
    ray = T(rs.normal(size=(B, R, 3))); ray[..., 2] = 1.0
    rgb_s = T(rs.uniform(0, 1, size=(B, R, N, 3, 2)))
    den_s = T(rs.gamma(0.7, 1.5, size=(B, R, N, 2)))
    den_s[0, 0] = 0.0                                           # fully empty ray
    den_s[0, 1, :, 0] = 50.0                                    # opaque at the first sample
    z = torch.sort(T(rs.uniform(5, 8, size=(B, R, N, 1))), dim=2).values
    unc_s = T(rs.gamma(1.0, 0.5, size=(B, R, N, 1)))
    out = NeRF.composite(opt, ray, rgb_s, den_s, z, unc_s)
    names = ("rgb", "rgb_static", "rgb_transient", "depth", "opacity", "opacity_static", "opacity_transient",
             "prob", "uncert", "alpha_static", "alpha_transient")
    _save("g7_composite", ray=ray, rgb_samples=rgb_s, density_samples=den_s, depth_samples=z, uncert_samples=unc_s,
          min_uncert=opt.nerf.min_uncert, **{"out_" + n: o for n, o in zip(names, out)})

    # ---------------------------------------------------------------- G7b composite backward
    leaves = [t.clone().requires_grad_() for t in (rgb_s, den_s, unc_s)]
    out = NeRF.composite(opt, ray, leaves[0], leaves[1], z, leaves[2])
    cot = [T(rs.normal(size=tuple(o.shape))) for o in out]
    loss = sum((o * c).sum() for o, c in zip(out, cot))
    loss.backward()
    _save("g7b_composite_bwd", **{"cot_" + n: c for n, c in zip(names, cot)},
          g_rgb_samples=leaves[0].grad, g_density_samples=leaves[1].grad, g_uncert_samples=leaves[2].grad)

    # ---------------------------------------------------------------- G8/G9 end-to-end render (a14,a15) + grads
    B, H, W, p, N = 2, 16, 16, 4, 8
    n_train = 5
    opt.H, opt.W, opt.batch_size, opt.patch_size = H, W, B, 16   # Discriminator ctor only accepts 16/32/64/128
    opt.nerf.sample_intvs = N
    opt.nerf.rand_rays = 48
    opt.data.image_size = [H, W]
    sc = _scene(B, H, W, seed=8)
    g = M.Graph(opt)
    load_params(g.nerf, O.make_params(seed_w))
    g.latent_vars_trans = torch.nn.Embedding(n_train, 16)
    g.latent_vars_light = torch.nn.Embedding(n_train, 48)
    ers = np.random.RandomState(77)
    with torch.no_grad():
        g.latent_vars_trans.weight.copy_(T(ers.normal(size=(n_train, 16))))
        g.latent_vars_light.weight.copy_(T(ers.normal(size=(n_train, 48))))
    torch.manual_seed(31)
    coords = torch.rand(B, p, p, 2) * 1.6 - 0.8
    zn, zf = sc["z_near"], sc["z_far"]
    idx = torch.tensor([3, 1])
    torch.manual_seed(41)
    rand = torch.rand(B, p * p, N, 1)
    torch.manual_seed(41)
    ret = g.render(opt, sc["pose"], intr=sc["intr"], ray_idx=coords, depth_range=(zn[:, :, None], zf[:, :, None]),
                   sample_idx=idx, mode="train")
    crs = np.random.RandomState(55)
    cot = {k: T(crs.normal(size=tuple(ret[k].shape))) for k in
           ("rgb", "rgb_static", "rgb_transient", "uncert", "depth", "density")}
    loss = sum((ret[k] * cot[k]).sum() for k in cot)
    loss.backward()
    assert all(q.grad is None for q in g.nerf.mlp_feat.parameters())
    grads = {}
    for name in ("mlp_rgb", "mlp_trans"):
        for li, layer in enumerate(getattr(g.nerf, name)):
            grads[f"g.{name}.{li}.weight"] = layer.weight.grad
            grads[f"g.{name}.{li}.bias"] = layer.bias.grad
    grads["g.latent_vars_trans"] = g.latent_vars_trans.weight.grad
    grads["g.latent_vars_light"] = g.latent_vars_light.weight.grad
    _save("g9_render_train", H=H, W=W, N=N, seed=seed_w, n_train=n_train, emb_seed=77, intr=sc["intr"], pose=sc["pose"],
          coords=coords, z_near=zn, z_far=zf, sample_idx=idx, rand=rand,
          **{"out_" + k: v for k, v in ret.items()}, **{"cot_" + k: v for k, v in cot.items()}, **grads)

    # val (all pixels, row 0) and eval (mask with holes, default fills), unstratified => deterministic
    opt.nerf.sample_stratified = False
    sc1 = _scene(1, H, W, seed=9)
    zn1, zf1 = sc1["z_near"], sc1["z_far"]
    mask = torch.zeros(H, W)
    mask[3:12, 2:13] = 1
    mask[5:7, 6:9] = 0
    with torch.no_grad():
        val = g.render_by_slices(opt, sc1["pose"], intr=sc1["intr"], depth_range=(zn1[:, :, None], zf1[:, :, None]),
                                 object_mask=mask[None], sample_idx=None, mode="val")
        ev = g.render_by_slices(opt, sc1["pose"], intr=sc1["intr"], depth_range=(zn1[:, :, None], zf1[:, :, None]),
                                object_mask=mask[None], sample_idx=torch.tensor(2), mode="eval_noalign")
    keep = ("rgb", "rgb_static", "rgb_transient", "opacity", "opacity_static", "opacity_transient", "uncert",
            "depth", "alpha_static", "alpha_transient", "density")
    _save("g9_render_slices", H=H, W=W, N=N, seed=seed_w, n_train=n_train, emb_seed=77, chunk=48,
          intr=sc1["intr"], pose=sc1["pose"], z_near=zn1, z_far=zf1, mask=mask, eval_sample_idx=2,
          **{"val_" + k: val[k] for k in keep}, **{"eval_" + k: ev[k] for k in keep})
    opt.nerf.sample_stratified = True

    # ---------------------------------------------------------------- G10/G11 patch gathers + losses (a16,a17)
    B, H, W, p = 2, 16, 16, 4
    irs = np.random.RandomState(99)
    var = _AttrDict()
    var.idx = torch.tensor([0, 1])
    var.image = T(irs.uniform(size=(B, 3, H, W)))
    var.image_syn = T(irs.uniform(size=(B, 3, H, W)))
    var.nocs_pred = T(irs.uniform(size=(B, 3, H, W)))
    var.normal_pred = T(irs.uniform(-1, 1, size=(B, 3, H, W)))
    yy, xx = np.mgrid[0:H, 0:W]
    var.obj_mask = T(((yy - 7.5) ** 2 + (xx - 7.5) ** 2 < 30).astype(np.float32))[None].repeat(B, 1, 1)
    var.mask_syn = T(((yy - 8.5) ** 2 + (xx - 6.5) ** 2 < 36).astype(np.float32))[None].repeat(B, 1, 1)
    coords = torch.rand(B, p, p, 2) * 2 - 1
    coords[0, 0, 0] = torch.tensor([1.0, 1.0])                 # nearest tap rounds out of range (quirk 10)
    coords[0, 0, 1] = torch.tensor([-1.0, -1.0])
    coords[0, 0, 2] = torch.tensor([0.5, 0.0])
    var.ray_idx = coords
    var.rgb = T(irs.uniform(size=(B, p * p, 3)))
    var.uncert = T(irs.uniform(0.05, 1.0, size=(B, p * p, 1)))
    var.density = T(irs.gamma(1.0, 1.0, size=(B, p * p, 8, 2)))
    opt.loss_weight.feat = None
    opt.loss_weight.gan_nerf = None
    loss = g.compute_loss(opt, var, mode="train", train_step="nerf")
    var = g.sample_geometry(opt, var, mode="train")
    # patch_real / patch_fake exactly as disc_forward builds them (without running the discriminator)
    rgb_img = var.rgb.view(B, p, p, 3).permute(0, 3, 1, 2).contiguous()
    pad = torch.logical_and(var.mask_syn_sample == 1, var.mask_sample == 0).float()
    patch_real = torch.cat([var.image_sample * var.mask_sample + rgb_img * pad, var.nocs_sample, var.normal_sample], 1)
    patch_fake = torch.cat([rgb_img, var.nocs_sample, var.normal_sample], 1)
    from model import base as Mbase
    mdl = Mbase.Model.__new__(Mbase.Model)
    tot = mdl.summarize_loss(opt, var, _AttrDict(dict(loss)))
    _save("g10_patch_gather", image=var.image, image_syn=var.image_syn, nocs=var.nocs_pred, normal=var.normal_pred,
          obj_mask=var.obj_mask, mask_syn=var.mask_syn, coords=coords, rgb=var.rgb, uncert=var.uncert,
          density=var.density, image_sample=var.image_sample, image_syn_sample=var.image_syn_sample,
          mask_sample=var.mask_sample, mask_syn_sample=var.mask_syn_sample, nocs_sample=var.nocs_sample,
          normal_sample=var.normal_sample, patch_real=patch_real, patch_fake=patch_fake,
          loss_render=loss.render, loss_uncert=loss.uncert, loss_trans_reg=loss.trans_reg, loss_all=tot.all,
          w_render=opt.loss_weight.render, w_uncert=opt.loss_weight.uncert, w_trans_reg=opt.loss_weight.trans_reg)
    # ---------------------------------------------------------------- G12 discriminator fwd + R1 (f1, stock module)
    from layers.discriminator import Discriminator
    opt.patch_size = 16
    disc = Discriminator(opt)
    drs = np.random.RandomState(321)
    O.seed_spectral_module(disc, 321)                          # seeded weights + deterministic spectral-norm u / v
    disc.eval()                                                # eval: no power iteration -> deterministic u / v
    xin = T(drs.uniform(size=(3, 9, 16, 16))).requires_grad_()
    scale = T(drs.uniform(0.25, 1.0, size=(3, 1, 1, 1)))
    d_out = disc(opt, xin, scale)
    reg = M.Graph.compute_grad2(opt, d_out, xin)
    bce1 = M.Graph.compute_gan_loss(opt, d_outs=d_out, target=1)
    _save("g12_discriminator", seed=321, x=xin, scale=scale, d_out=d_out, grad2=reg, bce_real=bce1)
    print("done")


if __name__ == "__main__":
    main()
