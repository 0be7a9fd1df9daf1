"""CPU: pin the oracle (oracle/texpose_oracle.py) to golden vectors captured from the
real reference (tests/golden/make_golden.py).  Same torch, same op order => tight tolerances."""
import numpy as np
import torch

from oracle import texpose_oracle as O
from conftest import load_golden

TIGHT = dict(rtol=1e-5, atol=1e-6)


def close(a, b, **kw):
    kw = kw or TIGHT
    torch.testing.assert_close(a, b, **kw)


def test_patch_sampler_g0():
    g = load_golden("g0_patch_sampler")
    lo = O.patch_min_scale(g["iterations"])
    assert abs(lo - float(g["scales_curr"][0])) < 1e-12
    coords, scales = O.patch_coords(g["patch_size"], g["u_scale"], g["u_hoff"], g["u_woff"], lo)
    close(coords, g["coords"])
    close(scales, g["scales"])


def test_rays_train_g1():
    g = load_golden("g1_rays_train")
    c, r = O.rays_train(g["intr"], g["coords"], g["pose"], g["H"], g["W"])
    close(c, g["center"])
    close(r, g["ray"], rtol=1e-5, atol=2e-6)
    zn, zf = O.bounds_train(g["coords"], g["z_near"], g["z_far"], g["H"], g["W"])
    close(zn, g["z_near_s"])
    close(zf, g["z_far_s"])


def test_rays_eval_g2():
    g = load_golden("g2_rays_eval")
    c, r = O.rays_eval(g["pose"], g["intr"], g["H"], g["W"])
    close(c, g["center"])
    close(r, g["ray"], rtol=1e-5, atol=2e-6)
    close(O.gather_rows(c, g["ray_idx"]), g["center_g"])
    close(O.gather_rows(r, g["ray_idx"]), g["ray_g"], rtol=1e-5, atol=2e-6)


def test_aabb_g3():
    g = load_golden("g3_aabb")
    lo, hi = O.enlarge_diagonal(g["aabb_min0"], g["aabb_max0"])
    close(lo, g["aabb_min"])
    close(hi, g["aabb_max"])
    tn, tf, ok = O.aabb_slab(lo, hi, g["o"], g["d"])
    assert torch.equal(ok.to(torch.uint8), g["valid"])
    torch.testing.assert_close(tn, g["t_near"], rtol=1e-6, atol=1e-6, equal_nan=True)
    torch.testing.assert_close(tf, g["t_far"], rtol=1e-6, atol=1e-6, equal_nan=True)
    assert g["valid"][0, 0] == 1 and g["valid"][0, 1] == 1 and g["valid"][0, 2] == 0 and g["valid"][0, 3] == 0


def test_sample_depth_g4():
    g = load_golden("g4_sample_depth")
    close(O.stratified_depths(g["near"], g["far"], g["N"]), g["z_mid"])
    close(O.stratified_depths(g["near"], g["far"], g["N"], g["rand"]), g["z_strat"])


def test_posenc_g5():
    g = load_golden("g5_posenc")
    close(O.posenc(g["x"], 10), g["enc10"])
    close(O.posenc(g["x"], 4), g["enc4"])
    # layout index = c*2L + s*L + l
    x = g["x"]
    L = 10
    e = O.posenc(x, L)
    c, s, l = 1, 1, 3
    close(e[..., c * 2 * L + s * L + l], torch.cos(x[..., c] * (2.0 ** l * np.pi)))


def test_mlp_full_g6():
    g = load_golden("g6_mlp_full")
    p = O.make_params(g["seed"])
    rgb, den, unc = O.mlp_forward(p, g["points"], g["ray_unit"], g["lat_trans"], g["lat_light"])
    close(rgb, g["rgb"], rtol=1e-5, atol=1e-6)
    close(den, g["density"], rtol=1e-5, atol=1e-6)
    close(unc, g["uncert"], rtol=1e-5, atol=1e-6)


def test_mlp_w32_g6():
    g = load_golden("g6_mlp_w32")
    p = {k[2:]: v for k, v in g.items() if k.startswith("w.")}
    ref = O.make_params(9, width=32)
    for k in p:
        assert torch.equal(p[k], ref[k]), k        # the recipe is reproducible
    rgb, den, unc = O.mlp_forward(p, g["points"], g["ray_unit"], g["lat_trans"], g["lat_light"])
    close(rgb, g["rgb"])
    close(den, g["density"])
    close(unc, g["uncert"])


COMP = ("rgb", "rgb_static", "rgb_transient", "depth", "opacity", "opacity_static", "opacity_transient",
        "prob", "uncert", "alpha_static", "alpha_transient")


def test_composite_g7():
    g = load_golden("g7_composite")
    out = O.composite(g["ray"], g["rgb_samples"], g["density_samples"], g["depth_samples"], g["uncert_samples"],
                      g["min_uncert"])
    for n, o in zip(COMP, out):
        close(o, g["out_" + n])
    # last interval is 1e10 => every opacity is 1 wherever sigma > 0
    assert abs(float(out[4][1, 3]) - 1) < 1e-5


def test_composite_bwd_g7b():
    g = load_golden("g7_composite")
    b = load_golden("g7b_composite_bwd")
    leaves = [g[k].clone().requires_grad_() for k in ("rgb_samples", "density_samples", "uncert_samples")]
    out = O.composite(g["ray"], leaves[0], leaves[1], g["depth_samples"], leaves[2], g["min_uncert"])
    sum((o * b["cot_" + n]).sum() for n, o in zip(COMP, out)).backward()
    close(leaves[0].grad, b["g_rgb_samples"], rtol=1e-4, atol=1e-5)
    close(leaves[1].grad, b["g_density_samples"], rtol=1e-4, atol=1e-4)
    close(leaves[2].grad, b["g_uncert_samples"], rtol=1e-4, atol=1e-5)


def _embeddings(n_train, seed):
    ers = np.random.RandomState(seed)
    et = torch.from_numpy(ers.normal(size=(n_train, 16)).astype(np.float32))
    el = torch.from_numpy(ers.normal(size=(n_train, 48)).astype(np.float32))
    return et, el


def test_render_train_g9():
    g = load_golden("g9_render_train")
    p = {k: v.clone() for k, v in O.make_params(g["seed"]).items()}
    for k, v in p.items():
        if not k.startswith("mlp_feat"):
            v.requires_grad_()
    et, el = _embeddings(g["n_train"], g["emb_seed"])
    et.requires_grad_()
    el.requires_grad_()
    ret = O.render(p, et, el, g["pose"], g["intr"], g["coords"], (g["z_near"][:, :, None], g["z_far"][:, :, None]),
                   g["sample_idx"], "train", g["H"], g["W"], g["N"], rand=g["rand"])
    for k in O.RENDER_KEYS:
        close(ret[k], g["out_" + k], rtol=2e-5, atol=2e-6)
    cot = {k[4:]: v for k, v in g.items() if k.startswith("cot_")}
    sum((ret[k] * cot[k]).sum() for k in cot).backward()
    for k, v in p.items():
        if k.startswith("mlp_feat"):
            assert v.grad is None
        else:
            close(v.grad, g["g." + k], rtol=1e-3, atol=1e-4)
    close(et.grad, g["g.latent_vars_trans"], rtol=1e-3, atol=1e-4)
    close(el.grad, g["g.latent_vars_light"], rtol=1e-3, atol=1e-4)
    assert torch.count_nonzero(el.grad.abs().sum(dim=1)) == 2      # only the B rows of var.idx


def test_render_slices_g9():
    g = load_golden("g9_render_slices")
    p = O.make_params(g["seed"])
    et, el = _embeddings(g["n_train"], g["emb_seed"])
    dr = (g["z_near"][:, :, None], g["z_far"][:, :, None])
    with torch.no_grad():
        val = O.render_by_slices(p, et, el, g["pose"], g["intr"], dr, g["mask"][None], None, "val",
                                 g["H"], g["W"], g["N"], chunk=g["chunk"])
        ev = O.render_by_slices(p, et, el, g["pose"], g["intr"], dr, g["mask"][None], g["eval_sample_idx"],
                                "eval_noalign", g["H"], g["W"], g["N"], chunk=g["chunk"])
    for k in O.RENDER_KEYS:
        close(val[k], g["val_" + k], rtol=2e-5, atol=2e-6)
        close(ev[k], g["eval_" + k], rtol=2e-5, atol=2e-6)
    off = (g["mask"].reshape(-1) == 0)
    assert torch.all(ev["uncert"][0, off] == 0.05) and torch.all(ev["alpha_static"][0, off] == 1)


def test_patch_gather_and_losses_g10():
    g = load_golden("g10_patch_gather")
    s = O.patch_gather(g["coords"], g["image"], g["image_syn"], g["nocs"], g["normal"], g["obj_mask"], g["mask_syn"])
    close(s["image"], g["image_sample"])
    close(s["image_syn"], g["image_syn_sample"])
    assert torch.equal(s["mask"], g["mask_sample"])
    assert torch.equal(s["mask_syn"], g["mask_syn_sample"])
    close(s["nocs_sample"], g["nocs_sample"])
    close(s["normal_sample"], g["normal_sample"])
    real, fake = O.disc_patches(g["rgb"], s)
    close(real, g["patch_real"])
    close(fake, g["patch_fake"])
    L = O.nerf_losses(g["rgb"], g["uncert"], g["density"], s)
    close(L["render"], torch.as_tensor(g["loss_render"]))
    close(L["uncert"], torch.as_tensor(g["loss_uncert"]))
    close(L["trans_reg"], torch.as_tensor(g["loss_trans_reg"]))
    tot = O.summarize(L, dict(render=g["w_render"], uncert=g["w_uncert"], trans_reg=g["w_trans_reg"]))
    close(tot, torch.as_tensor(g["loss_all"]))
    assert s["mask"][0, 0, 0, 0] == 0          # x=y=+1 rounds to tap 16 -> out of range (quirk 10)


def test_train_iteration_g13_oracle():
    """First iteration of G13 (real reference nerf_trainstep + disc_trainstep) re-done with the CPU oracle for the
    render / gathers / losses and stock torch for the discriminator: losses, head gradients, discriminator gradients
    (R1 double backward included)."""
    from texpose_amd.gan_modules import Discriminator
    from texpose_amd.options import default_options
    from texpose_amd.synthetic import training_batch
    g = load_golden("g13_train_iterations")
    B, H, W, P, N, n_train, stride = (g[k] for k in ("B", "H", "W", "P", "N", "n_train", "stride"))
    p = {k: v.clone() for k, v in O.make_params(g["seed_w"]).items()}
    for k, v in p.items():
        if not k.startswith("mlp_feat"):
            v.requires_grad_()
    et, el = _embeddings(n_train, g["seed_e"])
    et.requires_grad_()
    el.requires_grad_()
    var = training_batch(B, H, W, n_train=n_train, seed=g["seed_b"], device="cpu")
    coords, scales = g["it0.ray_idx"], g["it0.ray_scales"]
    ret = O.render(p, et, el, var.pose_init, var.intr, coords, (var.z_near[:, :, None], var.z_far[:, :, None]),
                   var.idx, "train", H, W, N, rand=g["it0.rand"])
    close(ret["rgb"], g["it0.rgb"], rtol=2e-5, atol=2e-6)
    smp = O.patch_gather(coords, var.image, var.image_syn, var.nocs_pred, var.normal_pred, var.obj_mask, var.mask_syn)
    L = O.nerf_losses(ret["rgb"], ret["uncert"], ret["density"], smp)
    opt = default_options(H=H, W=W, device="cpu")
    opt.patch_size = P
    disc = Discriminator(opt)
    O.seed_spectral_module(disc, g["seed_d"])
    disc.train()
    for q in disc.parameters():
        q.requires_grad_(False)
    real, fake = O.disc_patches(ret["rgb"], smp)
    bce = torch.nn.functional.binary_cross_entropy_with_logits
    d_fake_nerf = disc(opt, fake, scales)
    L["gan_nerf"] = bce(d_fake_nerf, torch.ones_like(d_fake_nerf))
    for k in ("render", "uncert", "trans_reg", "gan_nerf"):
        close(L[k], torch.as_tensor(g["it0.gloss." + k], dtype=torch.float32), rtol=1e-4, atol=1e-6)
    w = dict(render=0, uncert=0, trans_reg=-2, gan_nerf=-1)
    tot = O.summarize(L, w)
    close(tot, torch.as_tensor(g["it0.gloss.all"], dtype=torch.float32), rtol=1e-4, atol=1e-6)
    tot.backward()

    def rel(name, t):
        t = t.reshape(-1).double()
        ref = g[name].double() if name in g else g[name + ".sub"].double()
        if name not in g:
            t = t[::stride]
        return float((t - ref).norm() / ref.norm())

    for k, v in p.items():
        if k.startswith("mlp_feat"):
            assert v.grad is None
        else:
            assert rel("it0.grad.nerf." + k, v.grad) < 2e-3, k
    assert rel("it0.grad.latent_vars_light.weight", el.grad) < 2e-3 and rel("it0.grad.latent_vars_trans.weight", et.grad) < 2e-3
    # discriminator step (stock torch; the spectral-norm state has advanced by the generator-step forward, as in the
    # reference where the same module instance serves both steps)
    for q in disc.parameters():
        q.requires_grad_(True)
    real, fake = real.detach().requires_grad_(), fake.detach().requires_grad_()
    d_real, d_fake = disc(opt, real, scales), disc(opt, fake, scales)
    close(d_real, g["it0.d_real"], rtol=1e-4, atol=1e-5)
    l_real, l_fake = bce(d_real, torch.ones_like(d_real)), bce(d_fake, torch.zeros_like(d_fake))
    close(l_real, torch.as_tensor(g["it0.dloss.gan_disc_real"], dtype=torch.float32), rtol=1e-4, atol=1e-6)
    close(l_fake, torch.as_tensor(g["it0.dloss.gan_disc_fake"], dtype=torch.float32), rtol=1e-4, atol=1e-6)
    l_real.backward(retain_graph=True)
    gx = torch.autograd.grad(d_real.sum(), real, create_graph=True)[0]
    reg = gx.pow(2).reshape(B, -1).sum(1).mean()
    close(10.0 * reg, torch.as_tensor(g["it0.dloss.gan_reg_real"], dtype=torch.float32), rtol=1e-4, atol=1e-6)
    (10.0 * reg).backward()
    l_fake.backward()
    for name, q in disc.named_parameters():
        if q.grad is not None:
            assert rel("it0.grad.discriminator." + name, q.grad) < 1e-3, name


def test_eval_metrics_g14():
    g = load_golden("g14_eval_metrics")
    for name in ("native", "resized", "big"):
        H, W, oh, ow = (g[f"{name}.{k}"] for k in ("H", "W", "out_h", "out_w"))
        r = O.eval_metrics(g[f"{name}.rgb_static"], g[f"{name}.image"], g[f"{name}.obj_mask"], H, W,
                           out_hw=(oh, ow) if oh else None)
        assert abs(float(r["mse"]) - g[f"{name}.mse"]) < 1e-6 * g[f"{name}.mse"] + 1e-9
        assert abs(float(r["psnr"]) - g[f"{name}.psnr"]) < 1e-5
        assert abs(float(r["ssim"]) - g[f"{name}.ssim"]) < 1e-6


def test_philox_known_answer():
    # Random123 known-answer vectors for philox4x32-10
    z = O.philox4x32(np.zeros((1, 4), dtype=np.uint32), (0, 0))[0]
    assert [hex(int(v)) for v in z] == ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8"]
    f = np.full((1, 4), 0xFFFFFFFF, dtype=np.uint32)
    z = O.philox4x32(f, (0xFFFFFFFF, 0xFFFFFFFF))[0]
    assert [hex(int(v)) for v in z] == ["0x408f276d", "0x41c83b0e", "0xa20bc7c6", "0x6d5451fd"]
    u = O.philox_uniform(4096, seed=123, offset=5)
    assert u.dtype == np.float32 and u.min() >= 0 and u.max() < 1 and abs(u.mean() - 0.5) < 0.02


def test_box_range_restatement_matches_reference_g16():
    """the oracle's restatement of the stored-map pipeline (slab-test map in mm as compute_box.py:262-283
    writes it -> Crop_by_Pad -> mm to depth.scale units -> background fallback, data/lm.py:316-350,412-495) against the
    golden G16 the reference's own Dataset.Crop_by_Pad / preprocess_intrinsics / get_center_offset produced (cv2.resize
    restated: OpenCV float32 INTER_LINEAR), and the product's host-side crop camera against the same golden."""
    from conftest import load_golden
    from texpose_amd.geometry import crop_camera
    g = load_golden("g16_box_range")
    res = g["res"]
    for c in (0, 1):
        pre = "c%d_" % c
        K, R, t = g[pre + "K"], g[pre + "R"], g[pre + "t_mm"]
        pose_mm = torch.cat([R, t[:, None]], 1)[None]
        o, d = O.rays_eval(pose_mm, K[None], 480, 640)
        tn, tf, ok = O.aabb_slab(g[pre + "aabb_min_mm"].view(1, 1, 3), g[pre + "aabb_max_mm"].view(1, 1, 3), o, d)
        tn = torch.where(ok, tn, torch.zeros_like(tn)).view(480, 640)
        tf = torch.where(ok, tf, torch.zeros_like(tf)).view(480, 640)
        center, scale = g[pre + "center"].numpy(), g[pre + "scale"]
        zn, zf = O.range_from_box_map(torch.stack([tn, tf]).numpy(), center, scale, res, g["depth_scale"], (g["bg_lo"], g["bg_hi"]))
        assert torch.equal(zn, g[pre + "z_near"]) and torch.equal(zf, g[pre + "z_far"])
        off = O.crop_center_offset(center, scale, 480, 640)
        assert np.array_equal(off.astype(np.float32), g[pre + "center_offset"].numpy())
        assert torch.equal(O.crop_intrinsics(K, res / scale, center + off, res), g[pre + "intr_crop"])
        Kc, rect = crop_camera(K, center, scale, res)
        torch.testing.assert_close(Kc, g[pre + "intr_crop"], rtol=0, atol=2e-5)
        # the rectangle is exactly where the golden's padding ends: outside it every pixel has the background range
        x0, y0, x1, y1 = [int(v) for v in rect]
        far_img = g[pre + "z_far"].view(res, res)
        outside = torch.ones(res, res, dtype=torch.bool)
        outside[y0:y1, x0:x1] = False
        assert bool((far_img[outside] == g["bg_hi"] * g["depth_scale"]).all())
        assert c == 0 or outside.any()


def test_c1_literal_size_g17():
    """BASELINE config C1 at its literal size (64x64 crop, 32 samples per ray, batch 1; golden G17 rendered by the real
    reference): the oracle from intrinsics / poses reproduces the reference's rays bit for bit and both renders (train: a
    64x64 patch with the reference's own stratified depths; val: every pixel, mid-point samples) at rtol 2e-5."""
    g = load_golden("g17_c1_literal")
    H, W, N = g["H"], g["W"], g["N"]
    p = O.make_params(g["seed_w"])
    et, el = _embeddings(g["n_train"], g["emb_seed"])
    c, r = O.rays_train(g["intr"], g["coords"], g["pose"], H, W)
    assert torch.equal(c.reshape(1, -1, 3), g["train_in_center"]) and torch.equal(r.reshape(1, -1, 3), g["train_in_ray"])
    assert torch.equal(et[g["sample_idx"]], g["train_in_lat_t"]) and torch.equal(el[g["sample_idx"]], g["train_in_lat_l"])
    with torch.no_grad():
        rgb_s, den_s, unc_s = O.forward_samples(p, g["train_in_center"], g["train_in_ray"], g["train_in_depth"],
                                                g["train_in_lat_t"], g["train_in_lat_l"])
        out = O.composite(g["train_in_ray"], rgb_s, den_s, g["train_in_depth"], unc_s, 0.05)
    names = ("rgb", "rgb_static", "rgb_transient", "depth", "opacity", "opacity_static", "opacity_transient")
    for name, o in zip(names, out[:7]):
        close(o, g["train_" + name], rtol=2e-5, atol=2e-6)
    close(out[8], g["train_uncert"], rtol=2e-5, atol=2e-6)
    close(den_s, g["train_density"], rtol=2e-5, atol=2e-6)
    close(out[9], g["train_alpha_static"], rtol=2e-5, atol=2e-6)
    close(out[10], g["train_alpha_transient"], rtol=2e-5, atol=2e-6)
    dr = (g["z_near"][:, :, None], g["z_far"][:, :, None])
    with torch.no_grad():
        val = O.render_by_slices(p, et, el, g["pose"], g["intr"], dr, torch.ones(1, H, W), None, "val", H, W, N, chunk=H * W)
    for name in names + ("uncert",):
        close(val[name], g["val_" + name], rtol=2e-5, atol=2e-6)


def _rel_l2(a, b):
    a, b = a.detach().double(), b.detach().double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def test_flip_free_gradients_g9c():
    """G9c: gradients the REFERENCE's autograd produced on cotangents from which every gate-flip candidate (a sample -- or, at the
    render level, a ray -- with a head pre-activation inside a 64-ulp band around zero, found from the reference's own
    pre-activations by forward hooks) was removed.  Without flips the oracle must reproduce them tightly on every layer and on
    the latent rows: <= 1e-5 (G9's unmasked gradients only hold rtol 1e-3)."""
    g9, gb, gc = load_golden("g9_render_train"), load_golden("g9b_reference_rays"), load_golden("g9c_flipfree_grads")
    base = O.make_params(g9["seed"])

    def leaves():
        p = {k: v.clone().requires_grad_(not k.startswith("mlp_feat")) for k, v in base.items()}
        return p

    # (a) render level on the reference's own rays
    p = leaves()
    lt, ll = gb["train_lat_t"].clone().requires_grad_(), gb["train_lat_l"].clone().requires_grad_()
    rgb_s, den_s, unc_s = O.forward_samples(p, gb["train_center"], gb["train_ray"], gb["train_depth"], lt, ll)
    comp = O.composite(gb["train_ray"], rgb_s, den_s, gb["train_depth"], unc_s)
    ret = dict(rgb=comp[0], rgb_static=comp[1], rgb_transient=comp[2], depth=comp[3], uncert=comp[8])
    (sum((ret[k] * gc["a_cot_" + k]).sum() for k in ret) + (den_s * gc["a_cot_density"]).sum()).backward()
    errs = {k: _rel_l2(v.grad, gc["a_g." + k]) for k, v in p.items() if v.requires_grad}
    errs["lat_t"], errs["lat_l"] = _rel_l2(lt.grad, gc["a_g.lat_t"]), _rel_l2(ll.grad, gc["a_g.lat_l"])
    assert max(errs.values()) < 1e-5, errs
    assert 0 < float(gc["a_keep_ray"].sum()) < gc["a_keep_ray"].numel()
    # (b) MLP level, per-sample cotangents
    p = leaves()
    lt, ll = gc["b_lat_t"].clone().requires_grad_(), gc["b_lat_l"].clone().requires_grad_()
    outs = O.forward_samples(p, gc["b_center"], gc["b_ray"], gc["b_depth"], lt, ll)
    for o, k in zip(outs, ("rgb", "density", "uncert")):
        close(o, gc["b_out_" + k], rtol=2e-5, atol=2e-6)
    sum((o * gc["b_cot_" + k]).sum() for o, k in zip(outs, ("rgb", "density", "uncert"))).backward()
    errs = {k: _rel_l2(v.grad, gc["b_g." + k]) for k, v in p.items() if v.requires_grad}
    errs["lat_t"], errs["lat_l"] = _rel_l2(lt.grad, gc["b_g.lat_t"]), _rel_l2(ll.grad, gc["b_g.lat_l"])
    assert max(errs.values()) < 1e-5, errs


def test_nerf_step_g13c_oracle():
    """G13c: the REFERENCE's nerf_trainstep of iteration 0 (model/nerf_adapt_st_gan.py:108-127, 464-514, 747-776) with the reference's
    OWN rays, depth samples, latent rows, patch coordinates, scales and power-iteration state as inputs, re-done with the CPU oracle
    (MLP, composite, gathers, losses) and the discriminator mirror on CPU: render, D(fake), the four loss terms and their total at
    1e-5, the gradients of all 16 head tensors and both latent tables -- raw (gate flips: 5e-3) and flip-free (1e-4: the rays holding a
    gate within 64 ulp of zero carry no gradient, make_golden_g13c.py)."""
    from texpose_amd.gan_modules import Discriminator
    from texpose_amd.options import default_options
    from texpose_amd.synthetic import training_batch
    g = load_golden("g13c_nerf_step")
    B, H, W, P, N, n_train = (int(g[k]) for k in ("B", "H", "W", "P", "N", "n_train"))
    base = O.make_params(g["seed_w"])
    var = training_batch(B, H, W, n_train=n_train, seed=g["seed_b"], device="cpu")
    coords, scales, idx = g["ray_idx"], g["ray_scales"], g["sample_idx"].long()
    opt = default_options(H=H, W=W, device="cpu")
    opt.patch_size = P
    smp = O.patch_gather(coords, var.image, var.image_syn, var.nocs_pred, var.normal_pred, var.obj_mask, var.mask_syn)
    bce = torch.nn.functional.binary_cross_entropy_with_logits
    keep = g["keep_ray"]
    assert 0 < float(keep.sum()) < keep.numel()
    for tier, tol in (("raw", 5e-3), ("ff", 1e-4)):
        p = {k: v.clone().requires_grad_(not k.startswith("mlp_feat")) for k, v in base.items()}
        lt, ll = g["in.lat_t"].clone().requires_grad_(), g["in.lat_l"].clone().requires_grad_()
        rgb_s, den_s, unc_s = O.forward_samples(p, g["in.center"], g["in.ray"], g["in.depth"], lt, ll)
        comp = O.composite(g["in.ray"], rgb_s, den_s, g["in.depth"], unc_s)
        ret = dict(rgb=comp[0], uncert=comp[8], density=den_s)
        close(ret["rgb"], g["out.rgb"], rtol=1e-5, atol=1e-6)
        close(ret["uncert"], g["out.uncert"], rtol=1e-5, atol=1e-6)
        assert _rel_l2(den_s, g["out.density"]) < 1e-5
        if tier == "ff":
            for k in ret:
                m = keep.view(B, P * P, *([1] * (ret[k].dim() - 2)))
                ret[k] = m * ret[k] + (1 - m) * ret[k].detach()
        L = O.nerf_losses(ret["rgb"], ret["uncert"], ret["density"], smp)
        disc = Discriminator(opt)
        O.seed_spectral_module(disc, g["seed_d"])
        sd = disc.state_dict()
        sd.update({k: g["in." + k] for k in sd if "in." + k in g})
        disc.load_state_dict(sd)
        disc.train()
        for q in disc.parameters():
            q.requires_grad_(False)
        _, fake = O.disc_patches(ret["rgb"], smp)
        d_fake = disc(opt, fake, scales)
        close(d_fake, g["out.d_fake_nerf"], rtol=1e-4, atol=1e-6)
        for k, v in disc.state_dict().items():
            if "out." + k in g:
                assert _rel_l2(v, g["out." + k]) < 1e-5, k                    # one power iteration, as the reference's forward
        L["gan_nerf"] = bce(d_fake, torch.ones_like(d_fake))
        for k in ("render", "uncert", "trans_reg", "gan_nerf"):
            close(L[k], torch.as_tensor(g["gloss." + k], dtype=torch.float32), rtol=1e-5, atol=1e-6)
        tot = O.summarize(L, dict(render=0, uncert=0, trans_reg=-2, gan_nerf=-1))
        close(tot, torch.as_tensor(g["gloss.all"], dtype=torch.float32), rtol=1e-5, atol=1e-6)
        tot.backward()
        errs = {k: _rel_l2(v.grad, g["%s.g.%s" % (tier, k)]) for k, v in p.items() if v.requires_grad}
        # (the reference holds dense table gradients: row i = the sum of the latent-row gradients of the images with idx == i -- this
        # batch draws the same row twice)
        errs["lat_t"] = _rel_l2(torch.zeros(n_train, 16).index_add_(0, idx, lt.grad), g[tier + ".g.latent_vars_trans"])
        errs["lat_l"] = _rel_l2(torch.zeros(n_train, 48).index_add_(0, idx, ll.grad), g[tier + ".g.latent_vars_light"])
        assert len(errs) == 18 and max(errs.values()) < tol, (tier, errs)


def test_option_values_g19_oracle():
    """G19 (a), (b): the coarse-to-fine encoding weights (layers/nerf_static_transient_light.py:217-234) at three progress values --
    encodings bit-close, MLP outputs, flip-free head gradients -- and the density noise of train mode (:96-97) on the reference's own
    noise draw, with the CPU oracle."""
    g = load_golden("g19_options")
    rng, start = g["a.range"].tolist(), int(g["a.start"])
    base = O.make_params(int(g["a.seed_w"]))
    for tag in ("p005", "p027", "p100"):
        c2f = dict(progress=float(g[f"a.{tag}.progress"]), range=rng, start=start)
        for L in (10, 4):
            w = O.c2f_weight(L, c2f["progress"], rng, start)
            close(O.posenc(g["a.enc_x"], L, w), g[f"a.{tag}.enc{L}"], rtol=1e-6, atol=1e-6)
        if tag == "p005":
            # before the window only the bands below `start` carry weight (k = l - start < alpha)
            assert float(O.c2f_weight(10, c2f["progress"], rng, start)[start:].abs().sum()) == 0
        if tag == "p100":
            assert torch.equal(O.c2f_weight(4, c2f["progress"], rng, start), torch.ones(4))        # behind it: every band on
        p = {k: v.clone().requires_grad_(not k.startswith("mlp_feat")) for k, v in base.items()}
        lt, ll = g["a.lat_trans"].clone().requires_grad_(), g["a.lat_light"].clone().requires_grad_()
        outs = O.mlp_forward(p, g["a.points"], g["a.ray_unit"], lt, ll, c2f=c2f)
        for o, k in zip(outs, ("rgb", "density", "uncert")):
            close(o, g[f"a.{tag}.{k}"], rtol=2e-5, atol=2e-6)
        if tag != "p027":
            continue
        sum((o * g[f"a.{tag}.cot_{k}"]).sum() for o, k in zip(outs, ("rgb", "density", "uncert"))).backward()
        stride = int(g["stride"])
        for k, v in p.items():
            if not v.requires_grad:
                continue
            key = f"a.{tag}.g.{k}"
            t = v.grad.reshape(-1).double()
            ref = g[key].double() if key in g else g[key + ".sub"].double()
            t = t if key in g else t[::stride]
            assert float((t - ref).norm() / ref.norm()) < 1e-5, k
            assert abs(float(v.grad.double().norm()) - float(g[key + ".norm"])) <= 1e-5 * float(g[key + ".norm"]), k
        assert _rel_l2(p["mlp_rgb.0.weight"].grad[:, 256:283], g[f"a.{tag}.g.mlp_rgb.0.weight.viewenc"]) < 1e-5
        assert _rel_l2(lt.grad, g[f"a.{tag}.g.lat_t"]) < 1e-5 and _rel_l2(ll.grad, g[f"a.{tag}.g.lat_l"]) < 1e-5
    # (b)
    outs = O.mlp_forward(base, g["a.points"], g["a.ray_unit"], g["a.lat_trans"], g["a.lat_light"],
                         density_noise=g["b.noise"] * float(g["b.reg"]))
    close(outs[1], g["b.density_train"], rtol=2e-5, atol=2e-6)
    close(outs[0], g["b.rgb"], rtol=2e-5, atol=2e-6)
    plain = O.mlp_forward(base, g["a.points"], g["a.ray_unit"], g["a.lat_trans"], g["a.lat_light"])
    close(plain[1], g["b.density_val"], rtol=2e-5, atol=2e-6)


def test_ndc_and_inverse_depth_g20_oracle():
    """G20 (tests/golden/make_golden_g20_ndc_inverse.py): the options `camera.ndc` (camera.py:325-342) and `nerf.depth.param =
    inverse` (model/nerf_adapt_st_gan.py:699) of the reference, off in its shipped yaml: the oracle's restatements against the
    reference's values -- the ray transform on eval and train rays, the depth samples, and Graph.render with each and with both."""
    g = load_golden("g20_ndc_inverse")
    # (a) the transform is a handful of elementwise operations: same operations, same values
    c, d = O.rays_to_ndc(g["a.center"], g["a.ray"], g["a.intr"])
    assert torch.equal(c, g["a.center_ndc"]) and torch.equal(d, g["a.ray_ndc"])
    B = g["a.coords"].shape[0]
    c, d = O.rays_to_ndc(g["a.center_t"].view(B, -1, 3), g["a.ray_t"].view(B, -1, 3), g["a.intr"])
    assert torch.equal(c, g["a.center_t_ndc"]) and torch.equal(d, g["a.ray_t_ndc"])
    assert float((g["a.center_ndc"][..., 2] + 1).abs().max()) < 1e-5            # every centre on the near plane: z_ndc = -1
    # (b)
    N = int(g["b.N"])
    assert torch.equal(O.stratified_depths(g["b.near"], g["b.far"], N, None, "inverse"), g["b.z_mid"])
    assert torch.equal(O.stratified_depths(g["b.near"], g["b.far"], N, g["b.rand"], "inverse"), g["b.z_strat"])
    assert float(g["b.z_mid"].min()) > 4 and bool((g["b.z_mid"][:, :, 1:] < g["b.z_mid"][:, :, :-1]).all())      # metric, far to near
    # (c)
    H, W, N = int(g["c.H"]), int(g["c.W"]), int(g["c.N"])
    p = {k: v.clone().requires_grad_(not k.startswith("mlp_feat")) for k, v in O.make_params(int(g["c.seed"])).items()}
    et, el = _embeddings(int(g["c.n_train"]), int(g["c.emb_seed"]))
    et.requires_grad_()
    el.requires_grad_()
    ret = O.render(p, et, el, g["c.pose"], g["c.intr"], g["c.coords"], (g["c.ndc.z_near"][:, :, None], g["c.ndc.z_far"][:, :, None]),
                   g["c.sample_idx"], "train", H, W, N, rand=g["c.ndc.rand"], ndc=True)
    for k in O.RENDER_KEYS:
        close(ret[k], g["c.ndc.out_" + k], rtol=2e-5, atol=2e-6)
    cot = {k[len("c.ndc.cot_"):]: v for k, v in g.items() if k.startswith("c.ndc.cot_")}
    sum((ret[k] * cot[k]).sum() for k in cot).backward()
    stride = int(g["stride"])
    for k, v in p.items():
        if not v.requires_grad:
            continue
        key = "c.ndc.g." + k
        t = v.grad.reshape(-1).double()
        ref = g[key].double() if key in g else g[key + ".sub"].double()
        t = t if key in g else t[::stride]
        assert float((t - ref).norm() / ref.norm()) < 1e-3, k
        assert abs(float(v.grad.double().norm()) - float(g[key + ".norm"])) <= 1e-3 * float(g[key + ".norm"]), k
    close(et.grad, g["c.ndc.g.latent_vars_trans"], rtol=1e-3, atol=1e-4)
    close(el.grad, g["c.ndc.g.latent_vars_light"], rtol=1e-3, atol=1e-4)
    with torch.no_grad():
        every = torch.arange(H * W)[None]
        val = O.render(p, et, el, g["c.inv.pose"], g["c.inv.intr"], every, (g["c.inv.z_near"][:, :, None], g["c.inv.z_far"][:, :, None]),
                       None, "val", H, W, N, depth_param="inverse")
        both = O.render(p, et, el, g["c.pose"], g["c.intr"], g["c.coords"], (g["c.both.z_near"][:, :, None], g["c.both.z_far"][:, :, None]),
                        g["c.sample_idx"], "train", H, W, N, rand=g["c.both.rand"], ndc=True, depth_param="inverse")
    for k in O.RENDER_KEYS:
        close(val[k], g["c.inv.out_" + k], rtol=2e-5, atol=2e-6)
        close(both[k], g["c.both.out_" + k], rtol=2e-5, atol=2e-6)
