#!/usr/bin/env python3
"""Build-container check of the drop-in claim in INTEGRATION.md (needs /root/reference, so it cannot run on the GPU box):
construct the shim `model/nerf_adapt_st_gan_amd.py` would define -- the reference's Graph with the ray-marching path
swapped for texpose_amd's -- and compare its state dict, method table and option handling with the reference Graph.

    python tests/golden/check_shim_here.py
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG                                            # noqa: E402


def main():
    opt, camera, M, NeRF, RaySampler, FlexPatchSampler = MG._load_reference()
    opt.patch_size = 16
    from texpose_amd.graph import Graph as AmdGraph
    from texpose_amd.nerf import NeRF as AmdNeRF
    from texpose_amd.gan_modules import Discriminator as AmdDisc

    class Graph(M.Graph):
        def __init__(self, opt):
            super().__init__(opt)
            self.nerf = AmdNeRF(opt)
        render = AmdGraph.render
        render_by_slices = AmdGraph.render_by_slices
        _slice_rays = staticmethod(AmdGraph._slice_rays)
        _jitter = staticmethod(AmdGraph._jitter)
        _render_by_slices = AmdGraph._render_by_slices
        _range_guarded = AmdGraph._range_guarded
        sample_depth = staticmethod(AmdGraph.sample_depth)
        ray_batch_sample = staticmethod(AmdGraph.ray_batch_sample)
        gather_patches = AmdGraph.gather_patches
        sample_geometry = AmdGraph.sample_geometry
        compute_loss = AmdGraph.compute_loss
        _warn_once = AmdGraph._warn_once                 # (compute_loss says once when it leaves its fused launches)
        evaluate_metrics = AmdGraph.evaluate_metrics

    torch.manual_seed(0)
    ref = M.Graph(opt)
    shim = Graph(opt)
    ks_ref = {k: tuple(v.shape) for k, v in ref.state_dict().items()}
    ks_shim = {k: tuple(v.shape) for k, v in shim.state_dict().items()}
    print("reference Graph state_dict entries:", len(ks_ref), " shim:", len(ks_shim))
    assert ks_ref == ks_shim, set(ks_ref.items()) ^ set(ks_shim.items())
    shim.load_state_dict(ref.state_dict())                          # reference checkpoints load unchanged
    # our stock-compatible discriminator has the reference discriminator's keys as well
    d_ref = {k: tuple(v.shape) for k, v in ref.discriminator.state_dict().items()}
    d_amd = {k: tuple(v.shape) for k, v in AmdDisc(opt).state_dict().items()}
    assert d_ref == d_amd, set(d_ref.items()) ^ set(d_amd.items())
    print("discriminator state_dict identical:", len(d_ref), "entries")
    # the reference's trunk-only restore (util.restore_pretrain_partial_checkpoint) filters by these prefixes
    assert any(k.startswith("nerf.mlp_feat.") for k in ks_shim) and "nerf.progress" in ks_shim
    # signatures the reference engine calls (model/nerf_adapt_st_gan.py:471-545)
    import inspect
    for name in ("render", "render_by_slices", "compute_loss", "sample_geometry"):
        # (sample_geometry is a staticmethod in the reference and a method here; both are only ever called as
        # self.sample_geometry(opt, var, mode), :508,530 -- compare the argument lists without `self`)
        a = [p for p in inspect.signature(getattr(M.Graph, name)).parameters if p != "self"]
        b = [p for p in inspect.signature(getattr(Graph, name)).parameters if p != "self"]
        assert a == b[:len(a)], (name, a, b)
        print("signature ok:", name, a)
    # the other classes / functions of the boundary: same argument names, same order
    import camera as Rcam
    from texpose_amd import geometry as G
    from texpose_amd.nerf import NeRF as ANeRF
    def args(f):
        return [p for p in inspect.signature(f).parameters if p != "self"]
    for ref_f, our_f, name in (
            (NeRF.forward, ANeRF.forward, "NeRF.forward"), (NeRF.forward_samples, ANeRF.forward_samples, "NeRF.forward_samples"),
            (NeRF.composite, ANeRF.composite, "NeRF.composite"), (NeRF.positional_encoding, ANeRF.positional_encoding, "NeRF.positional_encoding"),
            (RaySampler.get_rays, G.RaySampler.get_rays, "RaySampler.get_rays"), (RaySampler.get_bounds, G.RaySampler.get_bounds, "RaySampler.get_bounds"),
            (RaySampler.get_image, G.RaySampler.get_image, "RaySampler.get_image"),
            (FlexPatchSampler.__call__, G.FlexPatchSampler.__call__, "FlexPatchSampler.__call__"),
            (Rcam.get_center_and_ray, G.get_center_and_ray, "camera.get_center_and_ray"),
            (Rcam.aabb_ray_intersection, G.aabb_ray_intersection, "camera.aabb_ray_intersection"),
            (M.Graph.sample_depth, AmdGraph.sample_depth, "Graph.sample_depth"), (M.Graph.ray_batch_sample, AmdGraph.ray_batch_sample, "Graph.ray_batch_sample")):
        a, b = args(ref_f), args(our_f)
        assert a == b[:len(a)], (name, a, b)
        print("signature ok:", name, a)
    print("shim signatures ok")
    run_training_steps(opt, M, Graph)
    print("shim check passed")


def run_training_steps(opt, M, ShimGraph):
    """The reference's OWN engine (Model.train_iteration -> nerf_trainstep / disc_trainstep, model/nerf_adapt_st_gan.py:
    108-202) drives the shim Graph for two iterations, next to the pure reference Graph on the same weights, batch and
    random draws.  There is no GPU here, so the four C-ABI entry points the shim's methods reach (tp_raygen, tp_mlp_fwd
    /bwd, tp_composite, tp_patch_gather) are bound to the CPU oracle for THIS CHECK ONLY: what is exercised is the wiring
    through the reference's method-resolution order (nerf_forward -> render -> compute_loss -> sample_geometry ...)."""
    import copy
    import types
    import numpy as np
    from oracle import texpose_oracle as O
    from texpose_amd import autograd_ops, ops
    from texpose_amd.nerf import NeRF as AmdNeRF

    B, H, W, N = 2, 32, 32, 4
    opt = copy.deepcopy(opt)
    opt.H, opt.W, opt.batch_size, opt.patch_size = H, W, B, 16
    opt.nerf.sample_intvs, opt.data.image_size = N, [H, W]
    opt.loss_weight.feat = None                                   # (VGG weights unavailable offline)
    opt.max_epoch, opt.max_iter = 10, 1000
    for k in ("scalar", "vis", "val", "ckpt"):
        opt.freq[k] = 10 ** 9

    # oracle-backed stand-ins for the HIP entry points (CPU tensors)
    def raygen(intr, pose, *, H, W, n_samples=0, coords=None, ray_idx=None, z_near=None, z_far=None, rand=None, **kw):
        c, r = O.rays_train(intr, coords, pose, H, W)
        zn, zf = O.bounds_train(coords, z_near.view(len(pose), -1, 1), z_far.view(len(pose), -1, 1), H, W)
        Bn = len(pose)
        c, r, zn, zf = c.reshape(Bn, -1, 3), r.reshape(Bn, -1, 3), zn.reshape(Bn, -1), zf.reshape(Bn, -1)
        if rand is None and kw.get("jitter") == ops.JITTER_PHILOX:          # the kernel's in-kernel stream; here: the draw the
            rand = torch.rand(Bn, c.shape[1], n_samples, 1)                  # reference makes at this point (:690-692)
        return c, r, zn, zf, O.stratified_depths(zn, zf, n_samples, rand)[..., 0]

    def forward_samples(self, opt_, center, ray, depth_samples, latent_variable_trans=None, latent_variable_light=None, mode=None):
        p = {k: v for k, v in self.named_parameters() if k.startswith("mlp_")}
        return O.forward_samples(p, center, ray, depth_samples, latent_variable_trans, latent_variable_light)

    def composite(opt_, ray, rgb_samples, density_samples, depth_samples, uncert_samples=None, per_sample=True, want_prob=True,
                  fan_out=None):                      # (the oracle stand-in hands out no aliases: every consumer reads rgb / density)
        return O.composite(ray, rgb_samples, density_samples, depth_samples, uncert_samples, opt_.nerf.min_uncert)

    def patch_gather(coords, image, image_syn, nocs, normal, obj_mask, mask_syn):
        g = O.patch_gather(coords, image, image_syn, nocs, normal, obj_mask, mask_syn)
        return torch.cat([g["image"], g["image_syn"], g["nocs_sample"], g["normal_sample"], g["mask"], g["mask_syn"]], dim=1)

    ops.raygen, ops.patch_gather = raygen, patch_gather
    AmdNeRF.forward_samples, AmdNeRF.composite = forward_samples, staticmethod(composite)

    def model(graph_cls, seed):
        torch.manual_seed(seed)
        m = M.Model.__new__(M.Model)
        m.graph = graph_cls(opt)
        m.graph.latent_vars_trans = torch.nn.Embedding(6, opt.nerf.N_latent_trans)
        m.graph.latent_vars_light = torch.nn.Embedding(6, opt.nerf.N_latent_light)
        m.train_data = list(range(6))
        M.Model.setup_optimizer(m, opt)
        m.it, m.ep = 0, 0
        now = __import__("time").time()
        m.timer = types.SimpleNamespace(start=now, it_start=now, it_end=now, it_mean=None)
        return m

    ref, shim = model(M.Graph, 5), model(ShimGraph, 6)
    state = O.seeded_state(ref.graph.state_dict(), salt=3)
    ref.graph.load_state_dict(state)
    shim.graph.load_state_dict(state)                              # same keys: the reference state dict loads into the shim
    rs = np.random.RandomState(4)
    f = lambda *sh: torch.from_numpy(rs.uniform(size=sh).astype(np.float32))
    yy, xx = np.mgrid[0:H, 0:W]
    disk = torch.from_numpy((((yy - H / 2) ** 2 + (xx - W / 2) ** 2) < (0.4 * H) ** 2).astype(np.float32))
    sc = O.synthetic_scene(H, W, B=B, seed=2)
    K = sc["intr"].clone()
    K[:, 0, 0] = K[:, 1, 1] = 700.0 * H / 128.0
    K[:, 0, 2], K[:, 1, 2] = W / 2.0, H / 2.0
    base = dict(idx=torch.tensor([1, 4]), image=f(B, 3, H, W), image_syn=f(B, 3, H, W), nocs_pred=f(B, 3, H, W),
                normal_pred=f(B, 3, H, W) * 2 - 1, obj_mask=disk[None].repeat(B, 1, 1), mask_syn=disk[None].repeat(B, 1, 1),
                intr=K, pose=sc["pose"], pose_init=sc["pose"], z_near=sc["z_near"], z_far=sc["z_far"])
    EasyDict = sys.modules["easydict"].EasyDict
    losses = {}
    for name, m in (("reference", ref), ("shim", shim)):
        torch.manual_seed(77)                                       # patch draws + stratified jitter + nothing else
        out = []
        for it in range(2):
            var = EasyDict({k: v.clone() for k, v in base.items()})
            gloss, dloss = M.Model.train_iteration(m, opt, var, loader=[0])
            out.append({**{"g." + k: float(v) for k, v in gloss.items()}, **{"d." + k: float(v) for k, v in dloss.items()}})
        losses[name] = out
    for it in range(2):
        for k, v in losses["reference"][it].items():
            w = losses["shim"][it][k]
            assert abs(v - w) <= 2e-4 * abs(v) + 1e-6, (it, k, v, w)
    moved = 0
    for (k, a), (_, b) in zip(ref.graph.state_dict().items(), shim.graph.state_dict().items()):
        assert torch.allclose(a, b, rtol=1e-3, atol=2e-5), (k, float((a - b).abs().max()))
        moved += int(not torch.equal(a, state[k]))
    assert moved > 20 and shim.it == 2 and shim.graph.patch_sampler.iterations == 1
    print("two reference-engine training iterations through the shim == through the reference Graph:",
          {k: round(v, 5) for k, v in losses["shim"][1].items()})


if __name__ == "__main__":
    main()
