"""Shared body of the checkpoint wire-format tests: texpose_amd against the manifest G15 of a checkpoint
the REFERENCE wrote (tests/golden/make_golden_g15_checkpoint.py).  Device-agnostic: the CPU suite runs it on "cpu" (host
logic), the GPU suite on "cuda:0"."""
import json
import os

import torch

from conftest import GOLDEN
from oracle import texpose_oracle as O


def _close(a, b, rtol=2e-5):
    return abs(a - b) <= rtol * max(abs(a), abs(b)) + 1e-9


def assert_state_matches(summary_ref, summary_ours, what):
    assert sorted(summary_ref) == sorted(summary_ours), (what, set(summary_ref) ^ set(summary_ours))
    for k, r in summary_ref.items():
        o = summary_ours[k]
        assert r["shape"] == o["shape"] and r["dtype"] == o["dtype"], (what, k, r, o)
        assert _close(r["sum"], o["sum"]) and _close(r["abssum"], o["abssum"]), (what, k, r, o)


def assert_optim_matches(ref, ours, what):
    assert len(ref["param_groups"]) == len(ours["param_groups"]), what
    for gr, go in zip(ref["param_groups"], ours["param_groups"]):
        assert gr["params"] == go["params"], (what, gr["params"], go["params"])         # group sizes and numbering
        for k in ("lr", "betas", "eps", "weight_decay", "alpha", "momentum", "amsgrad", "centered"):
            if k in gr:
                assert k in go and (gr[k] == go[k] or _close(float(gr[k]), float(go[k]))), (what, k, gr[k], go.get(k))
    assert sorted(ref["state"]) == sorted(ours["state"]), (what, sorted(ref["state"]), sorted(ours["state"]))
    for i, st in ref["state"].items():
        assert sorted(st) == sorted(ours["state"][i]), (what, i)
        for n, r in st.items():
            o = ours["state"][i][n]
            assert r["shape"] == o["shape"] and r["dtype"] == o["dtype"], (what, i, n, r, o)
            assert _close(r["sum"], o["sum"], 1e-4) and _close(r["abssum"], o["abssum"], 1e-4), (what, i, n, r, o)


def build(device, salt, n_train=6):
    from texpose_amd.gan_modules import Discriminator, PerceptualLoss
    from texpose_amd.graph import Graph
    from texpose_amd.options import default_options
    from texpose_amd.trainer import GanTrainer
    opt = default_options(device=str(device))
    opt.patch_size = 16
    graph = Graph(opt, discriminator=Discriminator(opt), perceptual_loss=PerceptualLoss()).to(device)
    graph.attach_latents(n_train, opt)
    graph.load_state_dict({k: v.to(device) for k, v in O.seeded_state(graph.state_dict(), salt=salt).items()})
    tr = GanTrainer(opt, graph, n_train=n_train)
    sched = torch.optim.lr_scheduler.ExponentialLR(tr.optim_nerf, gamma=float(opt.optim.sched.gamma))
    return opt, graph, tr, sched


def run(device, tmp_path):
    from texpose_amd import checkpoint as ck
    man = json.load(open(os.path.join(GOLDEN, "g15_checkpoint_manifest.json")))
    # ---- write: same contents, same optimiser steps -> the file texpose_amd writes has the reference file's manifest
    opt, graph, tr, sched = build(device, salt=11)
    assert abs(float(opt.optim.sched.gamma) - man["gamma"]) < 1e-15
    for optim in (tr.optim_nerf, tr.optim_disc):
        O.seeded_grads(optim, salt=5)
        optim.step()
    sched.step()
    path = str(tmp_path / "model.ckpt")
    ck.save_checkpoint(path, graph, epoch=3, it=1234, optim_nerf=tr.optim_nerf, optim_disc=tr.optim_disc, sched_nerf=sched)
    blob = torch.load(path, map_location="cpu", weights_only=False)
    ours = O.checkpoint_manifest(blob)
    assert ours["top_level"] == man["top_level"] and (ours["epoch"], ours["iter"]) == (man["epoch"], man["iter"])
    assert_state_matches(man["graph"], ours["graph"], "graph")
    assert_optim_matches(man["optim_nerf"], ours["optim_nerf"], "optim_nerf")
    assert_optim_matches(man["optim_disc"], ours["optim_disc"], "optim_disc")
    for k in ("gamma", "last_epoch", "_step_count", "base_lrs", "_last_lr"):
        a, b = man["sched_nerf"][k], ours["sched_nerf"][k]
        assert a == b or all(_close(x, y) for x, y in zip(a, b)), (k, a, b)
    # ---- read (resume): a graph with other contents restored from that blob == what the reference's restore left
    opt2, graph2, tr2, sched2 = build(device, salt=22)
    ep, it = ck.restore_checkpoint(graph2, torch.load(path, map_location=device, weights_only=False), resume=True,
                                   optim_nerf=tr2.optim_nerf, optim_disc=tr2.optim_disc, sched_nerf=sched2)
    rr = man["restored_resume"]
    assert (ep, it) == (rr["epoch"], rr["iter"])
    assert_state_matches(rr["graph"], O.state_summary(graph2.state_dict()), "resume graph")
    assert_optim_matches(rr["optim_nerf"], O.optim_summary(tr2.optim_nerf.state_dict()), "resume optim_nerf")
    assert_optim_matches(rr["optim_disc"], O.optim_summary(tr2.optim_disc.state_dict()), "resume optim_disc")
    # the resumed optimiser keeps working (state tensors on the right device, group layout 33 / 1 / 1)
    O.seeded_grads(tr2.optim_nerf, salt=6)
    tr2.optim_nerf.step()
    assert [len(g["params"]) for g in tr2.optim_nerf.param_groups] == [33, 1, 1]
    # ---- read (pre-training policy): only the frozen trunk is taken
    opt3, graph3, tr3, _ = build(device, salt=33)
    n = ck.restore_pretrained_trunk(graph3, torch.load(path, map_location=device, weights_only=False))
    assert n == 16
    assert_state_matches(man["restored_trunk_only"]["graph"], O.state_summary(graph3.state_dict()), "trunk-only graph")
    # ---- G15b: save_checkpoint(children=..., latest=False) and restore_pretrain_nerf (reference util.py:225-263)
    var = json.load(open(os.path.join(GOLDEN, "g15b_checkpoint_variants.json")))
    out = tmp_path / "run"
    main = ck.save_checkpoint_dir(str(out), graph, epoch=None, it=77, latest=False, children=tuple(var["children"]),
                                  optim_nerf=tr.optim_nerf, optim_disc=tr.optim_disc, sched_nerf=sched)
    copy = out / var["copy_relpath"]
    assert main == str(out / "model.ckpt") and copy.exists() and open(main, "rb").read() == open(copy, "rb").read()
    ours = O.checkpoint_manifest(torch.load(main, map_location="cpu", weights_only=False))
    ref = var["saved_children"]
    assert ours["top_level"] == ref["top_level"] and (ours["epoch"], ours["iter"]) == (ref["epoch"], ref["iter"]) == (None, 77)
    assert_state_matches(ref["graph"], ours["graph"], "children graph")
    assert all(k.startswith(tuple(var["children"])) for k in ours["graph"]) and len(ours["graph"]) == 34
    assert_optim_matches(ref["optim_nerf"], ours["optim_nerf"], "children optim_nerf")      # (optimisers are saved whole)
    ck.save_checkpoint_dir(str(out), graph, epoch=3, it=1234, latest=True, optim_nerf=tr.optim_nerf, optim_disc=tr.optim_disc,
                           sched_nerf=sched)
    assert not (out / "model" / "1234.ckpt").exists()                                        # latest=True: no copy
    opt4, graph4, tr4, _ = build(device, salt=44)
    ep, it = ck.restore_pretrain_nerf(graph4, torch.load(str(out / "model.ckpt"), map_location=device, weights_only=False))
    assert (ep, it) == (var["restored_nerf_only"]["epoch"], var["restored_nerf_only"]["iter"]) == (None, None)
    assert_state_matches(var["restored_nerf_only"]["graph"], O.state_summary(graph4.state_dict()), "nerf-only graph")
