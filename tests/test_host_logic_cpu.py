"""CPU: host-side logic of the mirror (no kernels): patch sampler vs the reference golden, pose algebra, module
construction / state-dict contract, option container, loud failure without the library."""
import os

import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import texpose_oracle as O
from texpose_amd import _lib
from texpose_amd.geometry import FlexPatchSampler, compose_poses, enlarge_diagonal, invert_pose, make_pose, rotation_distance
from texpose_amd.options import AttrDict, default_options


def test_flex_patch_sampler_matches_reference_g0():
    g = load_golden("g0_patch_sampler")
    ps = FlexPatchSampler(True, scale_anneal=0.0002)
    ps.iterations = g["iterations"]
    u = torch.stack([g["u_scale"], g["u_hoff"], g["u_woff"]]).view(3, 4, 1, 1, 1)
    coords, scales = ps(4, g["patch_size"], device="cpu", u=u)
    assert torch.equal(coords, g["coords"]) and torch.equal(scales, g["scales"])
    assert abs(ps.scales_curr[0] - float(g["scales_curr"][0])) < 1e-12
    # annealing schedule: min(0.8, max(0.25, exp(-it * 2e-4)))
    for it, want in ((0, 0.8), (3000, float(np.exp(-0.6))), (100000, 0.25)):
        ps.iterations = it
        assert abs(ps.scale_range()[0] - want) < 1e-12
    c, s = FlexPatchSampler(True, scale_anneal=0.0002)(3, 16, device="cpu")
    assert c.shape == (3, 16, 16, 2) and s.shape == (3, 1, 1, 1) and float(c.abs().max()) <= 1.0 + 1e-6


def test_pose_algebra():
    rs = np.random.RandomState(0)
    R = torch.from_numpy(np.stack([O.rotation_from_axis_angle(rs.normal(size=3)) for _ in range(3)]).astype(np.float32))
    t = torch.from_numpy(rs.normal(size=(3, 3)).astype(np.float32))
    p = make_pose(R, t)
    assert p.shape == (3, 3, 4)
    torch.testing.assert_close(invert_pose(p), O.pose_inverse(p))
    ident = compose_poses(p, invert_pose(p))
    torch.testing.assert_close(ident[..., :3], torch.eye(3).expand(3, 3, 3), atol=1e-6, rtol=0)
    torch.testing.assert_close(ident[..., 3], torch.zeros(3, 3), atol=1e-6, rtol=0)
    ang = rotation_distance(R, R.roll(1, 0))
    assert ang.shape == (3,) and bool((ang >= 0).all()) and float(rotation_distance(R, R).max()) < 1e-3
    lo, hi = enlarge_diagonal(torch.tensor([[-1.0, -2.0, 0.0]]), torch.tensor([[1.0, 2.0, 4.0]]))
    torch.testing.assert_close(hi - lo, torch.tensor([[2.5, 5.0, 5.0]]))
    with pytest.raises(ValueError):
        make_pose()


def test_nerf_module_contract():
    from texpose_amd.graph import Graph
    opt = default_options(device="cpu")
    g = Graph(opt)
    g.attach_latents(189, opt)
    sd = g.state_dict()
    shapes = {k: tuple(v.shape) for k, v in sd.items()}
    assert sum(p.numel() for p in g.nerf.parameters()) == 914186
    assert shapes["nerf.progress"] == ()
    assert [shapes[f"nerf.mlp_feat.{i}.weight"] for i in range(8)] == \
        [(256, 63), (256, 256), (256, 256), (256, 256), (256, 319), (256, 256), (256, 256), (257, 256)]
    assert [shapes[f"nerf.mlp_rgb.{i}.weight"] for i in range(4)] == [(256, 334), (256, 256), (256, 256), (3, 256)]
    assert [shapes[f"nerf.mlp_trans.{i}.weight"] for i in range(4)] == [(256, 272), (256, 256), (256, 256), (5, 256)]
    assert shapes["latent_vars_trans.weight"] == (189, 16) and shapes["latent_vars_light.weight"] == (189, 48)
    assert all(not p.requires_grad for p in g.nerf.mlp_feat.parameters())
    assert all(p.requires_grad for p in g.nerf.mlp_rgb.parameters())
    assert float(g.nerf.mlp_feat[0].bias.abs().max()) == 0.0                           # tf_init zeroes biases
    bad = default_options(device="cpu")
    bad.arch.layers_rgb = [None, 128, 128, 3]
    with pytest.raises(NotImplementedError):
        Graph(bad)


def test_discriminator_contract():
    from texpose_amd.gan_modules import Discriminator
    opt = default_options(device="cpu")
    d = Discriminator(opt)
    assert sum(p.numel() for p in d.parameters()) == 2667137
    keys = set(d.state_dict())
    for k in ("main.0", "main.3", "main.6", "final.1", "final.3", "final.5"):
        assert {f"{k}.weight_orig", f"{k}.weight_u", f"{k}.weight_v"} <= keys
    assert d.state_dict()["main.0.weight_orig"].shape == (256, 9, 4, 4)
    assert d.state_dict()["final.1.weight_orig"].shape == (64, 73, 1, 1)
    assert d(opt, torch.rand(2, 9, 16, 16), torch.rand(2, 1, 1, 1)).shape == (2,)
    opt.patch_size = 64
    assert sum(p.numel() for p in Discriminator(opt).parameters()) == 3294849


def test_options_container():
    o = AttrDict(a=dict(b=1), c=None)
    assert o.a.b == 1 and o.get("zzz") is None
    o.update(a=dict(b=2, d=[1, 2]))
    assert o.a.d == [1, 2]
    with pytest.raises(AttributeError):
        _ = o.nope
    d = default_options(H=480, W=640)
    assert d.nerf.sample_intvs == 64 and d.nerf.rand_rays == 2048 and d.loss_weight.trans_reg == -2


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", os.path.join(os.path.dirname(_lib.LIB_PATH), "no_such_lib.so"))
    with pytest.raises(_lib.TexposeLibraryError, match="no CPU or eager fallback"):
        _lib.load()


def test_checkpoint_wire_format(tmp_path):
    """Reference-format checkpoint round trip, per-child restore and the mlp_feat-only pre-training restore."""
    from texpose_amd import checkpoint as ck
    from texpose_amd.graph import Graph
    opt = default_options(device="cpu")
    torch.manual_seed(1)
    a = Graph(opt)
    a.attach_latents(7, opt)
    optim = torch.optim.Adam(a.nerf.mlp_rgb.parameters(), lr=1e-3)
    path = str(tmp_path / "model.ckpt")
    ck.save_checkpoint(path, a, epoch=3, it=1234, optim_nerf=optim, not_saved=object())
    blob = torch.load(path, weights_only=False)
    assert set(blob) == {"epoch", "iter", "graph", "optim_nerf"} and blob["iter"] == 1234
    assert "nerf.mlp_feat.7.weight" in blob["graph"] and "latent_vars_light.weight" in blob["graph"]
    torch.manual_seed(2)
    b = Graph(opt)
    b.attach_latents(7, opt)
    assert not torch.equal(a.nerf.mlp_rgb[0].weight, b.nerf.mlp_rgb[0].weight)
    ep, it = ck.restore_checkpoint(b, blob, optim_nerf=torch.optim.Adam(b.nerf.mlp_rgb.parameters(), lr=1e-3))
    assert (ep, it) == (3, 1234)
    for (k, v), (_, w) in zip(a.state_dict().items(), b.state_dict().items()):
        assert torch.equal(v, w), k
    torch.manual_seed(3)
    c = Graph(opt)
    c.attach_latents(7, opt)
    head_before = c.nerf.mlp_rgb[0].weight.clone()
    assert ck.restore_pretrained_trunk(c, blob) == 16                      # 8 weights + 8 biases
    assert torch.equal(c.nerf.mlp_feat[4].weight, a.nerf.mlp_feat[4].weight)
    assert torch.equal(c.nerf.mlp_rgb[0].weight, head_before)              # heads untouched


def test_discriminator_matches_reference_g12():
    """The stock PatchGAN module against a golden captured from the reference's Discriminator:
    forward logits, the R1 squared-gradient penalty (double backward) and the BCE loss."""
    from texpose_amd.gan_modules import Discriminator
    from texpose_amd.graph import Graph
    g = load_golden("g12_discriminator")
    opt = default_options(device="cpu")
    d = Discriminator(opt)
    O.seed_spectral_module(d, g["seed"])
    d.eval()
    x = g["x"].clone().requires_grad_()
    out = d(opt, x, g["scale"])
    torch.testing.assert_close(out.detach(), g["d_out"], rtol=1e-4, atol=1e-6)
    reg = Graph.compute_grad2(opt, out, x)
    torch.testing.assert_close(reg.detach(), g["grad2"], rtol=1e-3, atol=1e-8)
    torch.testing.assert_close(Graph.compute_gan_loss(opt, out, 1).detach(), torch.as_tensor(g["bce_real"]), rtol=1e-5,
                               atol=1e-6)


def test_synthetic_recipes_match_oracle_copies():
    """bench.py builds its scene from texpose_amd.synthetic only (the oracle is reserved for the cpu_baseline leg and
    the tests); the oracle keeps its own copy of the recipes -- both must generate identical data."""
    from oracle import texpose_oracle as O
    from texpose_amd import synthetic as S
    for seed, bs in ((3, 0.05), (0, 0.0)):
        a, b = O.make_params(seed, bias_scale=bs), S.network_weights(seed, bias_scale=bs)
        assert a.keys() == b.keys() and all(torch.equal(a[k], b[k]) for k in a)
    so, sp = O.synthetic_scene(48, 64, B=2, seed=5), S.eval_scene(48, 64, B=2, seed=5)
    for k in ("intr", "pose", "aabb_min", "aabb_max"):
        assert torch.equal(so[k], sp[k]), k
    assert S.LINEMOD_K == O.LINEMOD_K


def test_perceptual_pairs_equal_separate_passes():
    """PerceptualLoss.pairs (one pass over the feature network for both terms of the feature loss) gives the losses and
    gradients of the reference's four separate passes."""
    from texpose_amd.gan_modules import PerceptualLoss
    torch.manual_seed(0)
    P = PerceptualLoss()
    a, c = torch.rand(2, 3, 16, 16, requires_grad=True), torch.rand(2, 3, 16, 16, requires_grad=True)
    b, d = torch.rand(2, 3, 16, 16), torch.rand(2, 3, 16, 16)
    l1, l2 = P(a, b), P(c, d)
    (l1 + 5 * l2).backward()
    g1, g2 = a.grad.clone(), c.grad.clone()
    a.grad = c.grad = None
    m1, m2 = P.pairs((a, b), (c, d))
    (m1 + 5 * m2).backward()
    assert torch.allclose(m1, l1, rtol=1e-6, atol=0) and torch.allclose(m2, l2, rtol=1e-6, atol=0)
    assert torch.allclose(a.grad, g1, rtol=1e-5, atol=1e-9) and torch.allclose(c.grad, g2, rtol=1e-5, atol=1e-9)


def test_checkpoint_wire_format_matches_reference_written_file(tmp_path):
    """G15: texpose_amd writes / resumes / trunk-restores checkpoints exactly as the reference's util.save_checkpoint /
    restore_checkpoint / restore_pretrain_partial_checkpoint do (manifest captured from the reference)."""
    import checkpoint_contract
    checkpoint_contract.run(torch.device("cpu"), tmp_path)


def test_inorm_lrelu_second_order_formulas_fp64():
    """The closed forms csrc/inorm_lrelu.hip implements (header comment) against torch autograd in fp64."""
    torch.manual_seed(0)
    n, eps, slope = 16, 1e-5, 0.2
    x = torch.randn(3, n, dtype=torch.float64, requires_grad=True)
    gy = torch.randn(3, n, dtype=torch.float64, requires_grad=True)
    u = torch.randn(3, n, dtype=torch.float64)
    y = torch.nn.functional.leaky_relu(torch.nn.functional.instance_norm(x.view(1, 3, 4, 4), eps=eps).view(3, n), slope)
    gx, = torch.autograd.grad(y, x, gy, create_graph=True)
    g_x, g_gy = torch.autograd.grad((gx * u).sum(), (x, gy))
    xd = x.detach()
    mu = xd.mean(1, keepdim=True)
    r = 1 / torch.sqrt(((xd - mu) ** 2).mean(1, keepdim=True) + eps)
    xh = (xd - mu) * r
    s = torch.where(xh > 0, torch.ones_like(xh), torch.full_like(xh, slope))
    a = gy.detach() * s
    P = lambda v: v - v.mean(1, keepdim=True) - xh * (v * xh).mean(1, keepdim=True)
    A = (u * a).sum(1, keepdim=True) - n * u.mean(1, keepdim=True) * a.mean(1, keepdim=True)
    C, D = (u * xh).mean(1, keepdim=True), (a * xh).mean(1, keepdim=True)
    g_x_a = -(r * r / n) * xh * (A - n * C * D) - r * r * (D * (u - u.mean(1, keepdim=True)) + C * (a - a.mean(1, keepdim=True)) - 2 * C * D * xh)
    assert float((r * P(a) - gx.detach()).abs().max()) < 1e-12
    assert float((s * r * P(u) - g_gy).abs().max()) < 1e-12 and float((g_x_a - g_x).abs().max()) < 1e-12


def test_bench_spawns_its_own_ranks_for_gpus_n():
    """`python bench.py --gpus 2` with no torchrun environment: the parent starts two fresh rank processes (before any GPU
    call), they meet in a collective (gloo here) and rank 0's line comes through; a failing rank gives a non-zero exit code."""
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--spawn-check"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert lines == [{"spawn_check": 2, "rank_sum": 3.0}]
    # the C5 entry point through the same launcher: objects shard over the two ranks (stub renderer, gloo), rank 0's line relayed
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--config", "c5", "--spawn-check"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and lines[0]["spawn_check"] == 2 and lines[0]["config"] == "c5" and lines[0]["n_gpus"] == 2
    assert lines[0]["parallelism"] == "4 object(s) per GPU on 2 GPU(s)" and len(lines[0]["per_object_ms"]) == 8
    assert all(ms > 0 for ms in lines[0]["per_object_ms"])            # every object rendered by exactly one of the ranks
    # a rank that dies (no GPU here: the real bench raises in every child) must surface as a non-zero exit code of the parent
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", "1", "--no-train",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=300)
    if not torch.cuda.is_available():
        assert r.returncode != 0


def test_bench_legs_fail_loudly_and_traffic_figures_carry_their_source_sha(tmp_path, monkeypatch):
    """bench.run_leg: a leg that raises yields {"error": ...} AND its name in the list the line reports as "legs_failed" (bench.py
    exits non-zero after printing).  bench.traffic_entry: a committed PMC figure is reported stale as soon as one of the kernel
    sources it was measured on changes; every entry of the committed profiles/traffic.json carries a sha."""
    import importlib.util
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(repo, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    failed = []
    assert bench.run_leg(failed, "ok", lambda: {"v": 1}) == {"v": 1} and failed == []
    out = bench.run_leg(failed, "train", lambda: (_ for _ in ()).throw(RuntimeError("rccl said no")))
    assert failed == ["train"] and "rccl said no" in out["error"]
    for entry in bench.TRAFFIC_SOURCES:
        node, stale = bench.traffic_entry(entry)
        assert node is not None and stale is not None, entry           # (stale may be True while a kernel is being worked on)
        assert len(node["sources_sha16"]) == 16
    # a changed source flips the flag
    src = tmp_path / "texpose_amd" / "csrc"
    src.mkdir(parents=True)
    (tmp_path / "profiles").mkdir()
    for name in bench.TRAFFIC_SOURCES["hbm_kernels.raygen"]:
        (src / name).write_text("// v1 " + name)
    monkeypatch.setattr(bench, "REPO", str(tmp_path))
    sha = bench.sources_sha16("hbm_kernels.raygen")
    import json
    (tmp_path / "profiles" / "traffic.json").write_text(json.dumps({"hbm_kernels": {"raygen": {"total": 1.0, "sources_sha16": sha}}}))
    assert bench.traffic_entry("hbm_kernels.raygen")[1] is False
    (src / "raygen.hip").write_text("// v2")
    assert bench.traffic_entry("hbm_kernels.raygen")[1] is True
    assert bench.traffic_entry("f16x3") == (None, None)


def test_bench_byte_model_and_host_topology():
    """SURVEY 8d's per-unit bytes as bench.py prices them, and the /proc/cpuinfo census of the cpu_baseline leg."""
    import importlib.util
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(repo, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    rays = 480 * 640
    assert bench.hbm_bytes("composite_fwd", 1, 128) == 6212                      # 6,212 B per ray at N=128
    assert abs(bench.hbm_bytes("composite_fwd", rays, 128) / 1e9 - 1.908) < 1e-3  # 1.91 GB per 480x640 image
    assert bench.hbm_bytes("raygen", 1, 128) == 544                              # 544 B per ray at N=128
    assert bench.hbm_bytes("composite_bwd", 1, 64) == 64 * 76 + 68
    assert bench.hbm_bytes("patch_gather", 0, pixels=1024) == 1024 * 264
    sockets, phys, avail, model = bench.host_topology()
    assert sockets >= 1 and 1 <= phys <= avail and isinstance(model, str)


def test_disc_step_schedule_structure_and_eligibility():
    """texpose_amd/disc_step.py (K16) covers the [conv4s2, InstanceNorm, LeakyReLU]* + full-map ladders (patch 16 / 32) and says
    why not for the others; it is never eligible for CPU tensors (the caller keeps its autograd form there), and the prefetch
    queue of the discriminator refuses to be left half-used."""
    import pytest
    import torch
    from texpose_amd.disc_step import DiscStepSchedule
    from texpose_amd.gan_modules import Discriminator
    from texpose_amd.options import default_options
    for patch, stages in ((16, 2), (32, 3)):
        opt = default_options(H=128, W=128, device="cpu")
        opt.patch_size = patch
        d = Discriminator(opt).train()
        s = DiscStepSchedule(d)
        assert s.reason is None and len(s.stages) == stages and len(s.convs()) == stages + 4
        assert [c.weight_orig.shape for c in s.convs()] == [c.weight_orig.shape for c in d.sn_convs()]
        x = torch.zeros(4, 9 if opt.gan.geo_conditional else 3, patch, patch)
        assert not s.eligible(opt, x)                                   # CPU tensor
    for patch in (64, 128):                                               # first stage without a norm
        opt = default_options(H=128, W=128, device="cpu")
        opt.patch_size = patch
        s = DiscStepSchedule(Discriminator(opt))
        assert s.reason is not None and "ladder" in s.reason
    opt = default_options(H=128, W=128, device="cpu")
    d = Discriminator(opt)
    d.eval()
    with pytest.raises(RuntimeError):
        d.prefetch_spectral_weights(1)                                    # training mode only
    d.train()
    d._sn_queue.append(("stale",))
    with pytest.raises(RuntimeError):
        d.prefetch_spectral_weights(1)                                    # an unconsumed queue is an error


def test_eval_light_latent_pick_g18():
    """Graph.eval_light_index (reference model/nerf_adapt_st_gan.py:487-494, camera.py:345-350) against golden G18 -- the REFERENCE's
    rotation distances (bit for bit), its top-k candidates and the row its seeded `torch.randperm` draw handed to the renderer, for
    N_candidate 1 / 2 / 3 and several seeds (two anchors are nearly tied at 0.034 / 0.039 rad)."""
    from conftest import load_golden
    from texpose_amd.graph import Graph
    from texpose_amd.options import AttrDict
    g = load_golden("g18_eval_latent")
    opt = default_options(device="cpu")
    var = AttrDict(pose=g["pose"], pose_anchor=g["pose_anchor"])
    assert torch.equal(rotation_distance(var.pose[..., :3, :3], var.pose_anchor[..., :3, :3]), g["R_dist"])
    picks = set()
    for k, seed, picked in g["cases"].tolist():
        opt.render.N_candidate = k
        cand = torch.topk(g["R_dist"][:, None], k=k, dim=0, largest=False, sorted=True)[1][:, 0]
        assert torch.equal(cand, g["k%d_s%d_cand" % (k, seed)])
        torch.manual_seed(seed)
        idx = Graph.eval_light_index(opt, var)
        assert idx.dim() == 0 and int(idx) == picked == int(g["k%d_s%d_picked" % (k, seed)])
        picks.add(picked)
    assert len(picks) >= 3


def test_discriminator_geometry_encodings_cpu_mirror_g19c():
    """The mirror's geometry encodings (gan.L_nocs / L_normal / geo_c2f) on CPU tensors against the reference's own outputs (golden
    G19c): same check as the GPU test, stock torch ops for the ladder."""
    import g19_checks as T
    G = load_golden("g19_options")
    opt, disc = T.g19c_disc(G, torch.device("cpu"))
    T.g19c_check(G, opt, disc, torch.device("cpu"), wtol=1e-4)
    from texpose_amd.gan_modules import Discriminator
    from texpose_amd.options import default_options
    bad = default_options(H=32, W=32, device="cpu")
    bad.gan.L_nocs, bad.gan.L_normal = 2, 3
    with pytest.raises(ValueError):
        Discriminator(bad)


def test_knobs_are_parsed_once_warn_on_unknown_names_and_are_all_documented(monkeypatch):
    """texpose_amd.knobs: one frozen object; reload() follows the environment; a TP_* name nobody knows warns; INTEGRATION.md lists
    every switch; no other module of the package reads os.environ for a TP_* switch."""
    import glob
    import re
    import warnings
    from texpose_amd import knobs
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    doc = open(os.path.join(repo, "INTEGRATION.md")).read()
    for _field, env, _default, _doc in knobs._SPEC:
        assert "`%s`" % env in doc, env
    for env in knobs.LIBRARY_SWITCHES:
        assert "`%s`" % env in doc, env
    with pytest.raises(Exception):
        knobs.K.no_disc_pairs = True                              # frozen
    monkeypatch.setenv("TP_NO_DISC_PAIRS", "1")
    assert not knobs.K.no_disc_pairs                              # parsed once ...
    assert knobs.reload().no_disc_pairs and knobs.K.no_disc_pairs  # ... until asked again
    monkeypatch.setenv("TP_NO_DISK_PAIRS", "1")                   # a typo
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        knobs.reload()
    assert any("TP_NO_DISK_PAIRS" in str(x.message) for x in w)
    with knobs.override(linear_graphs=False):
        assert not knobs.K.linear_graphs
    assert knobs.K.linear_graphs
    for path in glob.glob(os.path.join(repo, "texpose_amd", "*.py")):
        if path.endswith(("knobs.py", "_lib.py", "dist.py")):      # (_lib: TEXPOSE_AMD_LIB; dist: the launcher's RANK / WORLD_SIZE)
            continue
        assert not re.search(r"os\.environ|getenv", open(path).read()), path
    # every switch the library itself reads is a known name
    for path in glob.glob(os.path.join(repo, "texpose_amd", "csrc", "*.h*")):
        for name in re.findall(r'getenv\("(TP_[A-Z0-9_]+)"\)', open(path).read()):
            assert name in knobs.LIBRARY_SWITCHES, (path, name)


def test_bench_stdout_carries_only_the_result_line():
    """bench.py's contract is ONE JSON line on stdout; libraries write to file descriptor 1 behind Python's back (RCCL prints a version
    block from C stdio when a communicator is created): `protect_stdout` sends everything but `emit` to stderr."""
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import bench, os, ctypes; bench.protect_stdout(); os.write(1, b'library banner\\n'); "
            "ctypes.CDLL(None).puts(b'C stdio banner, flushed at exit'); print('python chatter'); bench.emit({'ok': 1})")
    r = subprocess.run([sys.executable, "-c", code], cwd=repo, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.strip().splitlines() == ['{"ok": 1}'], r.stdout
    assert "library banner" in r.stderr and "C stdio banner" in r.stderr and "python chatter" in r.stderr
