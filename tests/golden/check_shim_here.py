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
    print("shim check passed")


if __name__ == "__main__":
    main()
