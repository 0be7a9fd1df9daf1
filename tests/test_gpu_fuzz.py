"""GPU parity, randomised: every small C-ABI entry point against the CPU oracle on seeded random shapes and values
(ragged sizes, single elements, out-of-range coordinates, rays that miss / graze / start inside the box, zero densities,
sizes that straddle the kernels' 64-lane / 256-thread / 32-pixel tiling).  Complements the golden-vector tests in
test_gpu_parity.py; integer / index work is compared bit-exactly, floating point at the SURVEY 8d bars."""
import numpy as np
import pytest
import torch

from oracle import texpose_oracle as O

pytestmark = pytest.mark.gpu

CASES = list(range(8))


def dev():
    return torch.device("cuda:0")


def cu(t):
    return t.to(dev())


def T(a):
    return torch.from_numpy(np.asarray(a, dtype=np.float32))


def rel_l2(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.fixture(scope="module")
def ops():
    from texpose_amd import ops as _ops
    return _ops


def _scene(rs, B, H, W):
    from texpose_amd import synthetic as S
    sc = S.eval_scene(H, W, B=B, seed=int(rs.randint(1 << 30)))
    K = sc["intr"].clone()
    K[:, 0, 0] = K[:, 1, 1] = float(rs.uniform(0.8, 2.0)) * H
    K[:, 0, 2], K[:, 1, 2] = W / 2 + float(rs.uniform(-2, 2)), H / 2 + float(rs.uniform(-2, 2))
    sc["intr"] = K
    return sc


@pytest.mark.parametrize("case", CASES)
def test_fuzz_raygen_train_and_eval(ops, case):
    rs = np.random.RandomState(100 + case)
    B, H, W = int(rs.randint(1, 4)), int(rs.randint(5, 40)), int(rs.randint(5, 40))
    p = int(rs.randint(1, 9))
    sc = _scene(rs, B, H, W)
    coords = T(rs.uniform(-1.15, 1.15, size=(B, p, p, 2)))            # slightly out of range: zero-padded bounds taps
    zn, zf = T(rs.uniform(5, 7, size=(B, H * W))), T(rs.uniform(8, 10, size=(B, H * W)))
    c, r, n_, f_, _ = ops.raygen(cu(sc["intr"]), cu(sc["pose"]), H=H, W=W, coords=cu(coords), z_near=cu(zn), z_far=cu(zf))
    c_o, r_o = O.rays_train(sc["intr"], coords, sc["pose"], H, W)
    torch.testing.assert_close(c.cpu().view_as(c_o), c_o, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(r.cpu().view_as(r_o), r_o, rtol=1e-5, atol=3e-6)
    n_o = O.bilinear_gather(zn.view(B, 1, H, W), coords)[:, 0]
    torch.testing.assert_close(n_.cpu().view_as(n_o), n_o, rtol=1e-6, atol=1e-6)
    # eval: random subset of pixel indices (with repeats)
    R = int(rs.randint(1, 300))
    idx = torch.from_numpy(rs.randint(0, H * W, size=(B, R)).astype(np.int64))
    c, r, n_, f_, _ = ops.raygen(cu(sc["intr"]), cu(sc["pose"]), H=H, W=W, ray_idx=cu(idx), z_near=cu(zn), z_far=cu(zf))
    c_all, r_all = O.rays_eval(sc["pose"], sc["intr"], H, W)
    g = lambda t: torch.gather(t, 1, idx[..., None].expand(-1, -1, t.shape[-1]))
    torch.testing.assert_close(r.cpu(), g(r_all), rtol=1e-5, atol=3e-6)
    assert torch.equal(n_.cpu(), torch.gather(zn, 1, idx)) and torch.equal(f_.cpu(), torch.gather(zf, 1, idx))


@pytest.mark.parametrize("case", CASES)
def test_fuzz_aabb_and_depths(ops, case):
    rs = np.random.RandomState(200 + case)
    n = int(rs.choice([1, 63, 64, 65, 257, 1000]))
    lo, hi = T(rs.uniform(-1.0, -0.2, size=(1, 1, 3))), T(rs.uniform(0.2, 1.0, size=(1, 1, 3)))
    o = T(rs.uniform(-3, 3, size=(1, n, 3)))
    d = T(rs.normal(size=(1, n, 3)))
    o[0, : n // 4] = T(rs.uniform(-0.1, 0.1, size=(n // 4, 3)))       # origins inside the box
    d[0, n // 2: n // 2 + n // 8, int(rs.randint(3))] = 0.0            # axis-parallel rays (division by zero lanes)
    tn, tf, ok = ops.aabb_intersect(lo, hi, cu(o), cu(d))
    tn_o, tf_o, ok_o = O.aabb_slab(lo, hi, o, d)
    assert torch.equal(ok.cpu().bool(), ok_o.bool())
    m = ok_o.bool()
    torch.testing.assert_close(tn.cpu()[m], tn_o[m], rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(tf.cpu()[m], tf_o[m], rtol=1e-6, atol=1e-6)
    # stratified depths: bit-exact with and without an injected uniform tensor
    N = int(rs.choice([1, 3, 4, 7, 64, 129]))
    near, far = T(rs.uniform(0, 5, size=(2, n))), T(rs.uniform(6, 30, size=(2, n)))
    u = T(rs.uniform(size=(2, n, N, 1)))
    assert torch.equal(ops.sample_depth(cu(near), cu(far), N).cpu()[..., None], O.stratified_depths(near, far, N))
    assert torch.equal(ops.sample_depth(cu(near), cu(far), N, rand=cu(u)).cpu()[..., None], O.stratified_depths(near, far, N, u))


@pytest.mark.parametrize("case", CASES)
def test_fuzz_posenc(ops, case):
    rs = np.random.RandomState(300 + case)
    n, L = int(rs.choice([1, 100, 257])), int(rs.choice([1, 4, 10]))
    x = T(rs.uniform(-9, 9, size=(n, 3)))
    torch.testing.assert_close(ops.posenc(cu(x), L).cpu(), O.posenc(x, L), rtol=1e-5, atol=2e-6)


@pytest.mark.parametrize("case", CASES)
def test_fuzz_composite_fwd_bwd(ops, case):
    rs = np.random.RandomState(400 + case)
    n = int(rs.choice([1, 2, 7, 33]))
    N = int(rs.choice([1, 2, 63, 64, 65, 127, 200, 513]))
    ray = T(rs.normal(size=(1, n, 3)))
    rgb = T(rs.uniform(size=(1, n, N, 3, 2)))
    den = T(rs.gamma(0.5, 0.3, size=(1, n, N, 2)))
    den[0, :, : N // 3] = 0.0                                           # empty space in front
    z = torch.sort(T(rs.uniform(5, 8, size=(1, n, N, 1))), dim=2).values
    unc = T(rs.gamma(1.0, 0.5, size=(1, n, N, 1)))
    leaves = [t.clone().requires_grad_() for t in (rgb, den, unc)]
    ref = O.composite(ray, leaves[0], leaves[1], z, leaves[2], 0.05)
    names = ("rgb", "rgb_static", "rgb_transient", "depth", "opacity", "opacity_static", "opacity_transient",
             "prob", "uncert", "alpha_static", "alpha_transient")
    r = dict(zip(names, ref))
    out, a_s, a_t, prob = ops.composite_fwd(cu(ray), cu(rgb), cu(den), cu(z), cu(unc), 0.05)
    for name, lo, hi in ops.COMPOSITE_RAY_FIELDS:
        torch.testing.assert_close(out.cpu()[..., lo:hi], r[name].detach(), rtol=1e-4, atol=1e-6)
    assert rel_l2(a_t, r["alpha_transient"]) < 1e-4 and rel_l2(prob, r["prob"][..., 0]) < 1e-4
    # backward of a random linear functional of the per-ray outputs (the last interval is 1e10 long: exclude the density
    # gradient of the last sample from the relative norm, it is either exactly 0 or O(1e10))
    cot = {k: T(rs.normal(size=tuple(r[k].shape))) for k in ("rgb", "rgb_static", "rgb_transient", "depth", "uncert")}
    sum((r[k] * cot[k]).sum() for k in cot).backward()
    g_out = torch.zeros(1, n, 14)
    for name, lo, hi in ops.COMPOSITE_RAY_FIELDS:
        if name in cot:
            g_out[..., lo:hi] = cot[name]
    g_rgb, g_den, g_unc = ops.composite_bwd(cu(ray), cu(rgb), cu(den), cu(z), cu(unc), cu(g_out), None, None, None, 0.05)
    assert rel_l2(g_rgb, leaves[0].grad) < 1e-4 and rel_l2(g_unc, leaves[2].grad) < 1e-4
    if N > 1:
        assert rel_l2(g_den[:, :, :-1], leaves[1].grad[:, :, :-1]) < 2e-4


@pytest.mark.parametrize("case", CASES)
def test_fuzz_patch_gather(ops, case):
    rs = np.random.RandomState(500 + case)
    B, H, W, p = int(rs.randint(1, 4)), int(rs.randint(4, 50)), int(rs.randint(4, 50)), int(rs.randint(1, 20))
    imgs = [T(rs.uniform(-1, 1, size=(B, 3, H, W))) for _ in range(4)]
    m1, m2 = T((rs.uniform(size=(B, H, W)) > 0.4).astype(np.float32)), T(rs.uniform(-1, 1, size=(B, H, W)))   # non-binary mask: > 0 rule
    coords = T(rs.uniform(-1.2, 1.2, size=(B, p, p, 2)))
    coords[0, 0, 0] = torch.tensor([1.0, 1.0])
    coords[0, -1, -1] = torch.tensor([-1.0, -1.0])
    out = ops.patch_gather(cu(coords), *[cu(t) for t in imgs], cu(m1), cu(m2)).cpu()
    s = O.patch_gather(coords, *imgs, m1, m2)
    torch.testing.assert_close(out[:, 0:3], s["image"], rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(out[:, 3:6], s["image_syn"], rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(out[:, 6:9], s["nocs_sample"], rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(out[:, 9:12], s["normal_sample"], rtol=1e-6, atol=1e-6)
    assert torch.equal(out[:, 12:13], s["mask"]) and torch.equal(out[:, 13:14], s["mask_syn"])


@pytest.mark.parametrize("case", CASES)
def test_fuzz_eval_metrics(ops, case):
    rs = np.random.RandomState(600 + case)
    B, H, W = int(rs.randint(1, 4)), int(rs.randint(3, 90)), int(rs.randint(3, 90))
    out_hw = None if case % 2 == 0 else (int(rs.randint(H, 3 * H)), int(rs.randint(W, 3 * W)))
    image = T(rs.uniform(size=(B, 3, H, W)))
    rgb = (0.6 * image + 0.4 * T(rs.uniform(size=(B, 3, H, W)))).permute(0, 2, 3, 1).reshape(B, H * W, 3).contiguous()
    mask = T((rs.uniform(size=(B, H, W)) > 0.3).astype(np.float32))
    ref = O.eval_metrics(rgb, image, mask, H, W, out_hw=out_hw)
    psnr, ssim, mse = ops.eval_metrics(cu(rgb), cu(image), cu(mask), H, W, out_hw=out_hw)
    assert abs(float(mse) - float(ref["mse"])) < 2e-5 * float(ref["mse"]) + 1e-9
    assert abs(float(ssim) - float(ref["ssim"])) < 5e-5


@pytest.mark.parametrize("case", CASES[:5])
def test_fuzz_mlp_forward_both_precisions(ops, case):
    rs = np.random.RandomState(700 + case)
    B, R, N = int(rs.randint(1, 4)), int(rs.randint(1, 40)), int(rs.choice([1, 5, 32, 64, 100]))
    params = O.make_params(40 + case)
    pts = T(rs.uniform(-1.5, 1.5, size=(B, R, N, 3)))
    unit = torch.nn.functional.normalize(T(rs.normal(size=(B, R, 1, 3))), dim=-1).expand(B, R, N, 3).contiguous()
    lt, ll = T(rs.normal(size=(B, 16))), T(rs.normal(size=(B, 48)))
    with torch.no_grad():
        ref = O.mlp_forward(params, pts, unit, lt, ll)
    for prec in ("fp32", "f16x3"):
        packed = ops.pack_weights({k: cu(v) for k, v in params.items()}, precision=prec)
        out = ops.mlp_forward(packed, cu(lt), cu(ll), points=cu(pts), ray_unit=cu(unit), precision=prec)
        for a, r in zip(out, ref):
            torch.testing.assert_close(a.cpu(), r, rtol=1e-4, atol=1e-6)
    ops.check_mlp_status(dev())


@pytest.mark.parametrize("case", CASES)
def test_fuzz_nerf_losses_fwd_bwd(ops, case):
    """K8 (render / uncert / trans_reg terms + gradients) against torch autograd through the oracle's nerf_losses."""
    from texpose_amd import autograd_ops
    rs = np.random.RandomState(800 + case)
    B, p, N = int(rs.randint(1, 5)), int(rs.choice([1, 3, 16, 20])), int(rs.choice([1, 8, 64, 100]))
    P = p * p
    gathered = T(rs.uniform(size=(B, 14, p, p)))
    gathered[:, 12] = T((rs.uniform(size=(B, p, p)) > 0.4).astype(np.float32))
    rgb, unc = T(rs.uniform(size=(B, P, 3))), T(rs.uniform(0.05, 1.5, size=(B, P, 1)))
    den = T(rs.gamma(1.0, 1.0, size=(B, P, N, 2)))
    leaves = [t.clone().requires_grad_() for t in (rgb, unc, den)]
    L = O.nerf_losses(leaves[0], leaves[1], leaves[2], dict(image=gathered[:, 0:3], mask=gathered[:, 12:13]))
    w = T(rs.uniform(0.5, 2.0, size=(3,)))
    (w[0] * L["render"] + w[1] * L["uncert"] + w[2] * L["trans_reg"]).backward()
    dl = [cu(t).requires_grad_() for t in (rgb, unc, den)]
    out = autograd_ops.nerf_losses(dl[0], dl[1], dl[2], cu(gathered))
    for a, k in zip(out, ("render", "uncert", "trans_reg")):
        assert abs(float(a) - float(L[k])) <= 2e-5 * abs(float(L[k])) + 1e-6, (k, float(a), float(L[k]))
    wd = cu(w)
    (wd[0] * out[0] + wd[1] * out[1] + wd[2] * out[2]).backward()
    for a, r in zip(dl, leaves):
        assert rel_l2(a.grad, r.grad) < 2e-5


def _split_fp16_vs_fp32_backward(ops, B, R, N, seed, mag):
    from texpose_amd.graph import Graph
    from texpose_amd.options import default_options
    rs = np.random.RandomState(seed)
    params = O.make_params(60 + seed % 37)
    opt = default_options(H=16, W=16, device="cuda:0")
    g = Graph(opt).to(dev())
    g.nerf.load_state_dict({**g.nerf.state_dict(), **{k: cu(v) for k, v in params.items()}})
    pts = cu(T(rs.uniform(-1.5, 1.5, size=(B, R, N, 3))))
    unit = cu(torch.nn.functional.normalize(T(rs.normal(size=(B, R, 1, 3))), dim=-1).expand(B, R, N, 3).contiguous())
    lt, ll = cu(T(rs.normal(size=(B, 16)))), cu(T(rs.normal(size=(B, 48))))
    cots = [cu(T(rs.normal(size=s))) * mag for s in ((B, R, N, 3, 2), (B, R, N, 2), (B, R, N, 1))]
    packed = g.nerf.packed_weights("f16x3")
    rgb, den, unc, saved = ops.mlp_forward(packed, lt, ll, points=pts, ray_unit=unit, save=True, precision="f16x3")
    ops.check_mlp_status(dev())
    res = {p: ops.mlp_backward(g.nerf, lt, ll, saved, rgb, den, unc, *cots, wgrad_precision=p) for p in ("fp32", "f16x3")}
    for a, b in zip(res["f16x3"]["params"], res["fp32"]["params"]):
        assert torch.isfinite(a).all() and rel_l2(a, b) < 2e-5, rel_l2(a, b)
    for k in ("lat_trans", "lat_light"):
        assert rel_l2(res["f16x3"][k], res["fp32"][k]) < 2e-5
    again = ops.mlp_backward(g.nerf, lt, ll, saved, rgb, den, unc, *cots, wgrad_precision="f16x3")
    for a, b in zip(res["f16x3"]["params"], again["params"]):          # fixed-order split-K: bit-identical on a re-run
        assert torch.equal(a, b)


@pytest.mark.parametrize("case", CASES)
def test_fuzz_split_fp16_backward_equals_fp32_mfma_backward(ops, case):
    """The same activation record (written by one f16x3 recording forward) through the fp32-MFMA backward and through
    the split-fp16 backward: identical ReLU gates, so the two differ only by arithmetic -- they must agree to ~1e-5,
    for random shapes (single sample, ragged tiles, up to 32 images) and cotangent magnitudes.  The two weight-gradient
    kernels also differ in everything else (128-row tiles against whole-GEMM workgroups, slice counts, one against two
    finalize passes), so this is also the layout test of the partial sums."""
    rs = np.random.RandomState(900 + case)
    B, R, N = int(rs.choice([1, 2, 5, 32])), int(rs.choice([1, 3, 17, 40])), int(rs.choice([1, 4, 31, 64]))
    _split_fp16_vs_fp32_backward(ops, B, R, N, 900 + case, float(10.0 ** rs.uniform(-8, 4)))


@pytest.mark.parametrize("B,R,N", [(5, 257, 33), (32, 128, 16), (3, 1024, 64)])
def test_split_fp16_backward_long_slices(ops, B, R, N):
    """Sizes at which every workgroup of the pipelined weight gradient walks many groups (42 k - 197 k samples: 9 - 40 groups
    per wide slice, odd and even counts, a ragged last tile, image boundaries inside a group, 32 one-hot columns)."""
    _split_fp16_vs_fp32_backward(ops, B, R, N, 4000 + B, 1.0)


# ------------------------------------------------------------------------------------------ K11 / K12 / K14 / K15 (f1)
@pytest.mark.parametrize("case", CASES)
def test_fuzz_patch_convolutions(ops, case):
    """Random shapes of the PatchGAN / feature-network convolution kernels (csrc/patch_conv.hip) against torch in fp64:
    image counts that do not fill a 32-row tile, channel counts off the 32-column tile and odd (the dgrad lane halves take
    even / odd channels), maps from 8x8 to 64x32, split-K and single-workgroup plans."""
    import torch.nn.functional as F
    rs = np.random.RandomState(300 + case)
    N, C_in, Co = int(rs.randint(1, 9)), int(rs.randint(1, 70)), int(rs.randint(1, 90))
    H, W = int(2 ** rs.randint(3, 7)), int(2 ** rs.randint(3, 6))
    x = cu(T(rs.normal(size=(N, C_in, H, W))))
    w = cu(T(rs.normal(size=(Co, C_in, 4, 4)) / (4 * C_in ** 0.5)))
    gy = cu(T(rs.normal(size=(N, Co, H // 2, W // 2))))
    xd, wd, gd = (t.double().cpu().requires_grad_() for t in (x, w, gy))
    yd = F.conv2d(xd, wd, None, 2, 1)
    gxd, gwd = torch.autograd.grad(yd, (xd, wd), gd)
    assert rel_l2(ops.conv4s2_fwd(x, w), yd) < 2e-6
    assert rel_l2(ops.conv4s2_dgrad(gy, w), gxd) < 2e-6
    assert rel_l2(ops.conv4s2_wgrad(gy, x), gwd) < 2e-6
    # 3x3 + bias + ReLU and its masked data gradient
    H3, W3 = int(2 ** rs.randint(2, 6)), int(2 ** rs.randint(2, 6))
    x3 = cu(T(rs.normal(size=(N, C_in, H3, W3))))
    w3 = cu(T(rs.normal(size=(Co, C_in, 3, 3)) / (3 * C_in ** 0.5)))
    b3 = cu(T(rs.normal(size=(Co,))))
    relu = bool(case & 1)
    x3d = x3.double().cpu().requires_grad_()
    y3d = F.conv2d(x3d, w3.double().cpu(), b3.double().cpu(), 1, 1)
    y3d = torch.relu(y3d) if relu else y3d
    g3 = cu(T(rs.normal(size=tuple(y3d.shape))))
    gx3d, = torch.autograd.grad(y3d, x3d, g3.double().cpu())
    y3 = ops.conv3s1_fwd(x3, w3, b3, relu)
    assert rel_l2(y3, y3d) < 2e-6
    assert rel_l2(ops.conv3s1_dgrad(g3, w3, y3 if relu else None), gx3d) < 2e-6


@pytest.mark.parametrize("case", CASES)
def test_fuzz_skinny_linear_and_head(ops, case):
    """Random sizes of K15 (x W^T for few rows) and K14 (the scale-conditioned head) against torch in fp64."""
    import torch.nn.functional as F
    from texpose_amd import autograd_ops
    rs = np.random.RandomState(400 + case)
    M, K, N = int(rs.randint(1, 40)), int(rs.randint(1, 3000)), int(rs.randint(1, 80))
    x, w, g = cu(T(rs.normal(size=(M, K)))), cu(T(rs.normal(size=(N, K)) / K ** 0.5)), cu(T(rs.normal(size=(M, N))))
    assert rel_l2(ops.skinny_linear_fwd(x, w), x.double().cpu() @ w.double().cpu().t()) < 2e-6
    assert rel_l2(ops.skinny_linear_wgrad(g, x), g.double().cpu().t() @ x.double().cpu()) < 2e-6
    B, C_z, H, L = int(rs.randint(1, 20)), int(rs.randint(1, 70)), int(rs.randint(1, 70)), int(rs.randint(0, 6))
    z0, s0 = cu(T(rs.normal(size=(B, C_z)))), cu(T(rs.uniform(0.25, 1.0, size=(B,))))
    Ws = [cu(T(rs.normal(size=(H, C_z + 2 * L + 1)) / 4)), cu(T(rs.normal(size=(H, H)) / 4)), cu(T(rs.normal(size=(1, H)) / 4))]
    outs = []
    for mine in (False, True):
        cast = (lambda t: t.clone()) if mine else (lambda t: t.double().cpu())
        z = cast(z0).requires_grad_()
        W = [cast(v).requires_grad_() for v in Ws]
        if mine:
            out = autograd_ops.disc_head(z, cast(s0), W[0], W[1], W[2], L, 0.2)
        else:
            s = cast(s0)
            freq = (2 ** torch.arange(L, dtype=torch.float32)).double() * float(torch.tensor(np.pi, dtype=torch.float32))
            spec = s.view(-1, 1) * freq
            a = torch.cat([z, spec.sin(), spec.cos(), s.view(-1, 1)], 1)
            t = F.leaky_relu(F.leaky_relu(F.leaky_relu(a, 0.2) @ W[0].t(), 0.2) @ W[1].t(), 0.2)
            out = (t @ W[2].t()).flatten()
        gz, = torch.autograd.grad(out.sum(), z, create_graph=True)
        grads = torch.autograd.grad(gz.pow(2).sum() + out.sum(), [z] + W)
        outs.append([t.detach().double().cpu() for t in (out, gz) + tuple(grads)])
    for a, b in zip(outs[1], outs[0]):
        assert rel_l2(a, b) < 5e-5
