#!/usr/bin/env python3
"""f16x3 on TRAINED weights: every other test and the bench use Xavier-random weights with zero
biases, so "NeRF activations are O(1..100), far below the 6e4 range guard" was an assertion.  This tool trains the two heads
and the latents with the product trainer on the synthetic GAN loop (the trunk is frozen in this stage, as in the reference:
model/nerf_adapt_st_gan.py:236-239), then renders an image with the resulting network through both MLP kernels and reports

    rays / s, how many images raised the range flag, the largest hidden activation the f16x3 kernel saw
    (tp_mlp_fwd_args.act_max), and f16x3-vs-exact-fp32 rel-L2 / max-rel of the per-ray outputs,

also with the trunk FEATURE scaled x4 / x16 (last trunk layer's feature rows and biases multiplied: what a trunk with larger
weights hands to the heads).  Used by bench.py (`trained_weights` object of the line) and tests/test_gpu_configs.py.

    python tools/trained_weights.py [--iters 500] [--H 480 --W 640 --samples 128]
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tools"))
import numpy as np                                                              # noqa: E402
import torch                                                                    # noqa: E402


def train_heads(device, iters=500, batch=4, graphed=True, seed=0):
    """`iters` iterations of the full GAN loop (C3 shapes) on synthetic crops; returns (graph, trainer, last losses)."""
    import train_dp
    from texpose_amd.options import AttrDict
    from texpose_amd.synthetic import training_batch
    torch.manual_seed(seed)
    opt, graph, cls = train_dp.build(device, batch, full=True, train_precision="f16x3", graphed=graphed)
    trainer = cls(opt, graph, n_train=189)
    batches = [training_batch(batch, 128, 128, seed=s, device=device) for s in range(4)]
    loss = None
    for i in range(iters):
        _, loss = trainer.train_iteration(AttrDict(dict(batches[i % 4])))
    torch.cuda.synchronize(device)
    return graph, trainer, {k: float(v) for k, v in loss.items() if torch.is_tensor(v)}


def eval_graph_from(trained, device, H, W, n_samples, trunk_feature_scale=1.0):
    """A fresh evaluation Graph at H x W x n_samples carrying the trained network (heads, latents; trunk as constructed in
    `trained`), the trunk feature optionally scaled."""
    from texpose_amd.graph import Graph
    from texpose_amd.options import default_options
    opt = default_options(H=H, W=W, device=str(device))
    opt.nerf.sample_intvs, opt.nerf.sample_stratified, opt.batch_size = n_samples, False, 1
    g = Graph(opt).to(device)
    g.nerf.load_state_dict(trained.nerf.state_dict())
    g.attach_latents(189, opt)
    with torch.no_grad():
        g.latent_vars_trans.weight.copy_(trained.latent_vars_trans.weight)
        g.latent_vars_light.weight.copy_(trained.latent_vars_light.weight)
        if trunk_feature_scale != 1.0:
            g.nerf.mlp_feat[7].weight[1:] *= trunk_feature_scale
            g.nerf.mlp_feat[7].bias[1:] *= trunk_feature_scale
    g.eval()
    return g, opt


def compare_kernels(trained, device, H=480, W=640, n_samples=128, scales=(1.0, 4.0, 16.0), images=2):
    """Per trunk-feature scale: f16x3 render (timed, range flag and activation maximum read back) vs the exact-fp32 kernel
    on the same rays / depths."""
    from texpose_amd import ops, synthetic
    sc = synthetic.eval_scene(H, W, B=1, seed=3)
    near, far = synthetic.scene_bounds(sc, H, W, device)
    pose, intr = sc["pose"].to(device), sc["intr"].to(device)
    dr = (near[:, :, None], far[:, :, None])
    mask = torch.ones(1, H, W, device=device)
    out = []
    ops.track_activation_max(True)
    try:
        for s in scales:
            g, opt = eval_graph_from(trained, device, H, W, n_samples, s)
            opt.arch.mlp_range_check = "off"                      # read the flag by hand: count, do not fall back

            def render():
                with torch.no_grad():
                    return g._render_by_slices(opt, pose, intr=intr, depth_range=dr, object_mask=mask, sample_idx=None, mode="val")

            g.nerf.precision = "f16x3"
            render()
            ops.take_mlp_status(device)
            ops.take_activation_max(device)
            flagged = 0
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            for _ in range(images):
                r16 = render()
                flagged += int(ops.take_mlp_status(device) & 1)
            torch.cuda.synchronize(device)
            dt = (time.perf_counter() - t0) / images
            act_max = ops.take_activation_max(device)
            g.nerf.precision = "fp32"
            r32 = render()
            err = {}
            for k in ("rgb", "rgb_static", "depth", "uncert"):
                a, b = r16[k].double(), r32[k].double()
                err[k] = dict(rel_l2=float((a - b).norm() / b.norm()), max_rel=float(((a - b).abs() / (b.abs() + 1e-6)).max()))
            dens = dict(rel_l2=float((r16.density.double() - r32.density.double()).norm() / r32.density.double().norm()))
            out.append(dict(trunk_feature_scale=s, rays_per_s=H * W / dt, ms_per_image=dt * 1e3, images=images,
                            range_flagged_images=flagged, max_hidden_activation=act_max, fp16_range_limit=6.0e4,
                            f16x3_vs_fp32=err, density=dens))
            del r16, r32, g
            torch.cuda.empty_cache()
    finally:
        ops.track_activation_max(False)
    return out


def run(device, iters=500, H=480, W=640, n_samples=128):
    t0 = time.perf_counter()
    graph, trainer, losses = train_heads(device, iters)
    t_train = time.perf_counter() - t0
    w = {k: float(p.detach().abs().max()) for k, p in graph.nerf.named_parameters() if k.endswith("weight") and k.startswith(("mlp_rgb", "mlp_trans"))}
    res = dict(trained="%d iterations of the full GAN loop (B=4, hipGraph trainer, synthetic crops; trunk frozen as in the "
                       "reference's adapt stage), %.1f s" % (iters, t_train),
               skipped_steps=trainer.skipped_steps, recording_forward_after=graph.nerf.train_precision,
               final_losses=losses, max_abs_head_weight=max(w.values()),
               renders=compare_kernels(graph, device, H, W, n_samples),
               workload="%dx%d x %d samples, mid-point depths (identical for both kernels)" % (H, W, n_samples))
    return res


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=500)
    ap.add_argument("--H", type=int, default=480)
    ap.add_argument("--W", type=int, default=640)
    ap.add_argument("--samples", type=int, default=128)
    a = ap.parse_args()
    print(json.dumps(run(torch.device("cuda:0"), a.iters, a.H, a.W, a.samples)))
