#!/usr/bin/env python3
"""G18: the evaluation path's LIGHT-LATENT pick (eval branch) by the REAL reference on CPU:
`Graph.nerf_forward(mode='eval_noalign')` (model/nerf_adapt_st_gan.py:485-502) computes the rotation distance of the test
pose to every anchor (training) pose (camera.py:345-350), takes the `opt.render.N_candidate` nearest anchors with `torch.topk`,
draws one of them with `torch.randperm` (global CPU generator) and renders the object-mask pixels with THAT row of
`latent_vars_light` (transient latent = zeros, render.transient = 'zero').

Run in the build container only:   python tests/golden/make_golden_g18_eval_latent.py

Stored per case (k = N_candidate in {1, 3}, several seeds): the test pose, the anchor poses, the manual seed, and what the
reference computed -- R_dist, the candidate indices, the picked row (captured from the latent the renderer was handed) -- plus the
per-ray outputs of the render of one case (mid-point samples, mask with holes), so that the GPU test checks the pick AND that
the picked row is the one that reaches the kernels."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG                                            # noqa: E402

PER_RAY = ("rgb", "rgb_static", "rgb_transient", "depth", "opacity", "opacity_static", "opacity_transient", "uncert")


def rot(rs, angle_scale):
    """Rotation by a random axis-angle of norm ~ angle_scale (Rodrigues, float64 -> float32)."""
    v = rs.normal(size=3)
    v = v / np.linalg.norm(v) * angle_scale * (0.5 + rs.uniform())
    th = np.linalg.norm(v)
    k = v / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K


def main():
    opt, camera, M, NeRF, RaySampler, FlexPatchSampler = MG._load_reference()
    from oracle import texpose_oracle as O
    T = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float32))
    B, H, W, N, n_train, seed_w = 1, 16, 16, 8, 9, 17
    opt.H, opt.W, opt.batch_size = H, W, B
    opt.nerf.sample_intvs = N
    opt.nerf.rand_rays = 48
    opt.nerf.sample_stratified = False
    opt.data.image_size = [H, W]
    sc = MG._scene(B, H, W, seed=23)
    g = M.Graph(opt)
    sd = g.nerf.state_dict()
    sd.update(O.make_params(seed_w))
    g.nerf.load_state_dict(sd)
    g.latent_vars_trans = torch.nn.Embedding(n_train, 16)
    g.latent_vars_light = torch.nn.Embedding(n_train, 48)
    ers = np.random.RandomState(79)
    emb_t, emb_l = T(ers.normal(size=(n_train, 16))), T(ers.normal(size=(n_train, 48)))
    with torch.no_grad():
        g.latent_vars_trans.weight.copy_(emb_t)
        g.latent_vars_light.weight.copy_(emb_l)
    seen = []
    orig = g.nerf.forward_samples

    def spy(opt_, center, ray, depth_samples, latent_variable_trans=None, latent_variable_light=None, mode=None):
        seen.append(dict(lat_t=latent_variable_trans.detach().clone(), lat_l=latent_variable_light.detach().clone()))
        return orig(opt_, center, ray, depth_samples, latent_variable_trans=latent_variable_trans,
                    latent_variable_light=latent_variable_light, mode=mode)

    g.nerf.forward_samples = spy
    mask = torch.zeros(H, W)
    mask[3:12, 2:13] = 1
    mask[5:7, 6:9] = 0
    rs = np.random.RandomState(5)
    R0 = sc["pose"][0, :, :3].double().numpy()
    # anchors: the test rotation perturbed by 0.05 .. 1.5 rad (two of them nearly tied), translation as the test pose
    scales = [0.9, 0.05, 1.5, 0.30, 0.0501, 0.6, 0.12, 1.1, 0.2]
    anchors = torch.stack([torch.cat([T(rot(rs, s) @ R0), sc["pose"][0, :, 3:]], dim=1) for s in scales])      # [n_train,3,4]
    out = dict(H=H, W=W, N=N, n_train=n_train, seed_w=seed_w, emb_seed=79, intr=sc["intr"], pose=sc["pose"], z_near=sc["z_near"],
               z_far=sc["z_far"], mask=mask, pose_anchor=anchors)
    cases = []
    for k, seed in ((1, 0), (3, 0), (3, 1), (3, 2), (3, 5), (2, 7)):
        opt.render.N_candidate = k
        var = MG._AttrDict()
        var.idx = torch.tensor([0])
        var.pose, var.pose_init, var.intr = sc["pose"], sc["pose"], sc["intr"]
        var.z_near, var.z_far = sc["z_near"], sc["z_far"]
        var.obj_mask = mask[None]
        var.pose_anchor = anchors
        R_dist = camera.rotation_distance(var.pose[..., :3, :3], var.pose_anchor[..., :3, :3]).unsqueeze(-1)
        cand = torch.topk(R_dist, k=k, dim=0, largest=False, sorted=True)[1]
        del seen[:]
        torch.manual_seed(seed)
        with torch.no_grad():
            var = g.nerf_forward(opt, var, mode="eval_noalign")
        lat = seen[0]["lat_l"].reshape(-1)
        picked = int((emb_l - lat[None]).abs().sum(-1).argmin())
        assert torch.equal(emb_l[picked], lat) and float(seen[0]["lat_t"].abs().sum()) == 0.0
        assert picked in cand[:, 0].tolist()
        cases.append((k, seed, picked))
        tag = "k%d_s%d" % (k, seed)
        out[tag + "_cand"] = cand[:, 0]
        out[tag + "_picked"] = np.int64(picked)
        if (k, seed) == (3, 1):
            out.update({"render_" + kk: var[kk] for kk in PER_RAY})
            out["render_case"] = np.array([k, seed])
    out["R_dist"] = R_dist[:, 0]
    out["cases"] = np.array(cases)
    print("cases (k, seed, picked):", cases, "R_dist", [round(float(x), 4) for x in R_dist[:, 0]])
    assert len({c[2] for c in cases}) >= 3, "the seeds should exercise different draws"
    MG._save("g18_eval_latent", **out)


if __name__ == "__main__":
    main()
