"""CPU, world_size 2, gloo: the data-parallel layer (texpose_amd/dist.py).  Each rank renders its shard of the
image batch with the CPU oracle (test infrastructure; the HIP kernels need a GPU), back-propagates a per-image
loss, and the single flat all-reduce must reproduce the gradients of the full batch computed in one process."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from oracle import texpose_oracle as O          # noqa: E402
from texpose_amd import dist as tdist           # noqa: E402
from texpose_amd import knobs                  # noqa: E402

B, P, N, H, W, N_TRAIN = 4, 3, 4, 12, 12, 6


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _setup(width=32):
    params = O.make_params(5, width=width)
    rs = np.random.RandomState(2)
    emb_t = torch.from_numpy(rs.normal(size=(N_TRAIN, 16)).astype(np.float32))
    emb_l = torch.from_numpy(rs.normal(size=(N_TRAIN, 48)).astype(np.float32))
    sc = O.synthetic_scene(H, W, B=B, seed=4)
    K = sc["intr"].clone()
    K[:, 0, 0] = K[:, 1, 1] = 700.0 * H / 128.0
    K[:, 0, 2], K[:, 1, 2] = W / 2.0, H / 2.0
    coords = torch.from_numpy(rs.uniform(-0.8, 0.8, size=(B, P, P, 2)).astype(np.float32))
    rand = torch.from_numpy(rs.uniform(size=(B, P * P, N, 1)).astype(np.float32))
    target = torch.from_numpy(rs.uniform(size=(B, P * P, 3)).astype(np.float32))
    idx = torch.tensor([1, 4, 0, 3])
    return params, emb_t, emb_l, sc, K, coords, rand, target, idx


def _loss_and_grads(images, params, emb_t, emb_l, sc, K, coords, rand, target, idx, denom):
    p = {k: v.clone().requires_grad_(not k.startswith("mlp_feat")) for k, v in params.items()}
    et, el = emb_t.clone().requires_grad_(), emb_l.clone().requires_grad_()
    sel = torch.tensor(list(images))
    ret = O.render(p, et, el, sc["pose"][sel], K[sel], coords[sel],
                   (sc["z_near"][sel][:, :, None], sc["z_far"][sel][:, :, None]), idx[sel], "train", H, W, N,
                   rand=rand[sel])
    loss = ((ret["rgb"] - target[sel]) ** 2).sum() / denom + ret["density"][..., 1].sum() / denom
    loss.backward()
    named = [(k, v) for k, v in p.items() if v.requires_grad] + [("emb_t", et), ("emb_l", el)]
    return named


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    r, w, _ = tdist.init_distributed("gloo")
    assert (r, w) == (rank, world)
    setup = _setup()
    shard = tdist.shard_batch(B, rank, world)
    # per-rank loss is normalised by the LOCAL element count; averaging the ranks' gradients then equals the
    # full-batch mean because the shards are equal-sized
    named = _loss_and_grads(shard, *setup, denom=float(len(shard) * P * P))
    params = [torch.nn.Parameter(v.detach().clone()) for _, v in named]
    for q, (_, v) in zip(params, named):
        q.grad = None if v.grad is None else v.grad.clone()
    if rank == 1:
        params[0].grad = None                       # a rank without a gradient contributes zeros
    red = tdist.FlatGradAllReducer(params)
    red.reduce()
    # step-gate words ride in the tail of the same buffer: set on every rank if set on any rank; gradients unaffected even
    # when another element of the buffer is non-finite
    keep = [q.grad.clone() for q in params]
    flags = torch.tensor([1, 0, 0] if rank == 0 else [0, 7, 0], dtype=torch.int32)
    if rank == 0:
        params[1].grad.view(-1)[0] = float("nan")
    red.reduce(flags=flags)
    assert flags.tolist() == [1, 1, 0], flags
    assert all(torch.equal(a.grad, b) for i, (a, b) in enumerate(zip(params, keep)) if i != 1)
    for q, k in zip(params, keep):
        q.grad = k
    word = tdist.all_reduce_flags(torch.tensor([0, rank], dtype=torch.int32))
    assert word.tolist() == [0, 1]
    bufs = torch.nn.Linear(2, 2)
    with torch.no_grad():
        bufs.weight.fill_(float(rank))
    tdist.broadcast_module_state(bufs, src=0)
    s = tdist.all_reduce_scalars(torch.tensor(float(rank + 1)), torch.tensor(2.0))
    torch.save(dict(grads=[q.grad for q in params], nbytes=red.nbytes, w=bufs.weight.clone(), s=s,
                    first_local=None if rank == 1 else named[0][1].grad.clone()),
               os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_flat_allreduce_matches_full_batch(tmp_path):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(tmp_path / "rank0.pt")
    r1 = torch.load(tmp_path / "rank1.pt")
    full = _loss_and_grads(range(B), *_setup(), denom=float(B * P * P))
    assert r0["nbytes"] == (sum(v.numel() for _, v in full) + tdist.FlatGradAllReducer.FLAG_WORDS) * 4
    for i, ((name, v), g0, g1) in enumerate(zip(full, r0["grads"], r1["grads"])):
        assert torch.equal(g0, g1), name                      # identical on every rank after the collective
        if i == 0:
            torch.testing.assert_close(g0, r0["first_local"] / 2, rtol=1e-6, atol=1e-8)   # rank 1 had None -> zeros
            continue
        torch.testing.assert_close(g0, v.grad, rtol=2e-4, atol=1e-6, msg=name)
    # embedding rows: only the rows of the images some rank rendered are non-zero
    emb_l_grad = r0["grads"][-1]
    assert set(torch.nonzero(emb_l_grad.abs().sum(1)).flatten().tolist()) == {0, 1, 3, 4}
    assert torch.all(r1["w"] == 0) and torch.all(r0["w"] == 0)
    assert [float(x) for x in r0["s"]] == [3.0, 4.0]


def test_shard_batch_partitions():
    for n in (1, 4, 7, 32, 307200):
        for world in (1, 2, 3, 8):
            parts = [tdist.shard_batch(n, r, world) for r in range(world)]
            flat = [i for p in parts for i in p]
            assert flat == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


# ---------------------------------------------------------------------------------------------------------------------
# trainer-level glue (texpose_amd.trainer.GanTrainer + dist.setup_data_parallel / shard_training_batch), world size 2.
# The HIP render and the HIP patch gather are replaced by the CPU oracle IN THIS TEST ONLY (the product has no CPU path).
# ---------------------------------------------------------------------------------------------------------------------
TB, TH, TN = 4, 32, 4          # global batch, crop size, samples per ray


def _oracle_backed_graph(opt):
    from texpose_amd.gan_modules import Discriminator
    from texpose_amd.graph import Graph
    from texpose_amd.options import AttrDict

    class CpuGraph(Graph):
        def render(self, opt, pose, intr=None, ray_idx=None, depth_range=None, sample_idx=None, mode=None, rand=None):
            p = {k: v for k, v in self.nerf.named_parameters() if k.startswith("mlp_")}
            if rand is None:
                rand = torch.rand(pose.shape[0], ray_idx.shape[1] * ray_idx.shape[2], opt.nerf.sample_intvs, 1)
            return AttrDict(O.render(p, self.latent_vars_trans.weight, self.latent_vars_light.weight, pose, intr, ray_idx,
                                     depth_range, sample_idx, mode, opt.H, opt.W, opt.nerf.sample_intvs, rand=rand))

        def gather_patches(self, opt, var):
            B = len(var.idx)
            g = O.patch_gather(var.ray_idx, var.image, var.image_syn, var.nocs_pred, var.normal_pred,
                               var.obj_mask.view(B, opt.H, opt.W), var.mask_syn.view(B, opt.H, opt.W))
            var.image_sample, var.image_syn_sample = g["image"], g["image_syn"]
            var.nocs_sample, var.normal_sample = g["nocs_sample"], g["normal_sample"]
            var.mask_sample, var.mask_syn_sample = g["mask"], g["mask_syn"]
            return var

    return CpuGraph(opt, discriminator=Discriminator(opt))


def _trainer_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from texpose_amd.options import AttrDict, default_options
    from texpose_amd.synthetic import training_batch
    from texpose_amd.trainer import GanTrainer
    tdist.init_distributed("gloo")
    opt = default_options(H=TH, W=TH, device="cpu")
    opt.batch_size, opt.patch_size, opt.nerf.sample_intvs = TB // world, 16, TN
    opt.loss_weight.feat = None
    torch.manual_seed(100 + rank)                    # ranks start from DIFFERENT weights: the broadcast must fix that
    graph = _oracle_backed_graph(opt)
    graph.attach_latents(6, opt)
    graph.train()
    before = graph.nerf.mlp_rgb[0].weight.detach().clone()
    r, w = tdist.setup_data_parallel(graph, seed=7)
    assert (r, w) == (rank, world)
    tr = GanTrainer(opt, graph, n_train=6, max_iter=10)
    calls = []
    for name, red in (("nerf", tr.red_nerf), ("disc", tr.red_disc)):
        def wrapped(orig=red.all_reduce, name=name, red=red):
            calls.append((name, [p.grad is not None for p in red.params].count(True)))
            return orig()
        red.all_reduce = wrapped
    full = training_batch(TB, TH, TH, n_train=6, seed=3, device="cpu")
    mine = tdist.shard_training_batch(full, rank, world)
    start = {k: v.detach().clone() for k, v in graph.state_dict().items()}
    coords = []
    for _ in range(2):
        var, loss = tr.train_iteration(AttrDict(dict(mine)))
        coords.append(var.ray_idx.detach().clone())
        assert all(bool(torch.isfinite(v)) for v in loss.values() if torch.is_tensor(v))
    torch.save(dict(start=start, end={k: v.detach().clone() for k, v in graph.state_dict().items()}, calls=calls,
                    coords=coords, idx=mine.idx.clone(), changed_by_broadcast=not torch.equal(before, start["nerf.mlp_rgb.0.weight"]),
                    frame=mine.frame_index.clone()), os.path.join(out_dir, f"trainer_rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_data_parallel_trainer_glue(tmp_path):
    """Two ranks, two full GAN iterations each on its half of a global batch of 4: identical start (broadcast), different
    patch draws per rank, disjoint images, one gradient all-reduce per optimiser step issued after ALL backward passes of
    the step (nerf: 1; discriminator: real + R1 + fake), identical parameters and buffers on both ranks afterwards."""
    world = 2
    mp.spawn(_trainer_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(tmp_path / "trainer_rank0.pt")
    r1 = torch.load(tmp_path / "trainer_rank1.pt")
    assert not r0["changed_by_broadcast"] and r1["changed_by_broadcast"]
    for k in r0["start"]:
        assert torch.equal(r0["start"][k], r1["start"][k]), k
    assert set(r0["frame"].tolist()).isdisjoint(r1["frame"].tolist()) and len(r0["idx"]) == len(r1["idx"]) == 2
    assert not torch.equal(r0["coords"][0], r1["coords"][0])               # per-rank random streams
    # per iteration: one reduce for the nerf step, then one for the discriminator step, every trainable parameter
    # holding a gradient at that point (discriminator: the real, R1 and fake backward have all run)
    assert [c[0] for c in r0["calls"]] == ["nerf", "disc", "nerf", "disc"] == [c[0] for c in r1["calls"]]
    n_heads = 16 + 2
    assert all(c[1] == n_heads for c in r0["calls"] if c[0] == "nerf")
    assert all(c[1] == 6 for c in r0["calls"] if c[0] == "disc")
    moved = 0
    for k in r0["end"]:
        assert torch.equal(r0["end"][k], r1["end"][k]), k                  # same averaged gradients -> same updates
        moved += int(not torch.equal(r0["end"][k], r0["start"][k]))
    assert moved >= n_heads + 6
    assert all(torch.equal(r0["end"][k], r0["start"][k]) for k in r0["end"] if k.startswith("nerf.mlp_feat"))


# ---------------------------------------------------------------------------------------------------------------------
# DP(2 x B/2) == single(B): the several-rank step's data path -- gradients scaled by 1 / world into the flat buffer (pack), ONE SUM
# all-reduce, the optimiser reading the buffer's views (adopt) -- against ONE process on the whole batch with the same patch
# coordinates and stratified draws.  The hipGraph segments of trainer.GraphedGanTrainer need the GPU; what they capture around
# the collective are these same three FlatGradAllReducer calls (tests/test_gpu_parity.py::test_linear_form_with_all_reduces_between_
# graphs_is_bit_identical runs them captured, in a 1-rank RCCL group).
# ---------------------------------------------------------------------------------------------------------------------
def _dp_equiv_setup(world, rank):
    from texpose_amd.options import AttrDict, default_options
    from texpose_amd.synthetic import training_batch
    from texpose_amd.trainer import GanTrainer
    opt = default_options(H=TH, W=TH, device="cpu")
    opt.batch_size, opt.patch_size, opt.nerf.sample_intvs = TB // world, 16, TN
    opt.loss_weight.feat = None
    torch.manual_seed(11)
    graph = _oracle_backed_graph(opt)
    graph.attach_latents(6, opt)
    graph.train()
    tr = GanTrainer(opt, graph, n_train=6, max_iter=10)
    full = training_batch(TB, TH, TH, n_train=6, seed=3, device="cpu")
    full.obj_mask = torch.ones_like(full.obj_mask)           # equal photometric normalisers on every shard (dist.setup_data_parallel)
    gen = torch.Generator().manual_seed(5)
    patch_u = torch.rand(3, TB, 1, 1, 1, generator=gen)
    jitter = torch.rand(TB, 256, TN, 1, generator=gen)
    sl = tdist.shard_batch(TB, rank, world)
    mine = tdist.shard_training_batch(full, rank, world)
    mine.patch_u, mine.jitter_rand = patch_u[:, sl.start:sl.stop].contiguous(), jitter[sl.start:sl.stop].contiguous()
    grads = {}
    for name, optim, params in (("nerf", tr.optim_nerf, tr.nerf_group), ("disc", tr.optim_disc, tr.disc_group)):
        def step(orig=optim.step, name=name, params=params):
            grads.setdefault(name, [None if p.grad is None else p.grad.detach().clone() for p in params])     # (first iteration's)
            return orig()
        optim.step = step
    return tr, graph, AttrDict(dict(mine)), grads


def _dp_equiv_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    tdist.init_distributed("gloo")
    tr, graph, mine, grads = _dp_equiv_setup(world, rank)
    start = {k: v.detach().clone() for k, v in graph.state_dict().items()}
    for _ in range(2):
        tr.train_iteration(type(mine)(dict(mine)))
    flat_backed = all(p.grad is None or p.grad.data_ptr() == v.data_ptr() for red in (tr.red_nerf, tr.red_disc)
                      for p, v in zip(red.params, red.views))
    torch.save(dict(start=start, end={k: v.detach().clone() for k, v in graph.state_dict().items()}, grads=grads,
                    flat_backed=flat_backed), os.path.join(out_dir, f"dp_equiv_rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_two_ranks_on_half_batches_equal_one_process_on_the_whole_batch(tmp_path):
    world = 2
    mp.spawn(_dp_equiv_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0, r1 = (torch.load(tmp_path / f"dp_equiv_rank{r}.pt") for r in range(world))
    torch.set_num_threads(4)
    tr, graph, whole, grads = _dp_equiv_setup(1, 0)
    start = {k: v.detach().clone() for k, v in graph.state_dict().items()}
    for _ in range(2):
        tr.train_iteration(type(whole)(dict(whole)))
    end = {k: v.detach().clone() for k, v in graph.state_dict().items()}
    assert r0["flat_backed"] and r1["flat_backed"]                       # the optimisers read the averaged gradients in the flat buffers
    for k in start:
        assert torch.equal(start[k], r0["start"][k]) and torch.equal(r0["end"][k], r1["end"][k]), k
    # first-iteration gradients: average over the ranks of the half-batch gradients == the whole-batch gradient
    n_checked = 0
    for name in ("nerf", "disc"):
        for g_dp, g_dp1, g_one in zip(r0["grads"][name], r1["grads"][name], grads[name]):
            assert (g_dp is None) == (g_one is None)
            if g_one is None:
                continue
            assert torch.equal(g_dp, g_dp1)
            scale = float(g_one.abs().max())
            if name == "nerf" and g_one.dim() == 2 and g_one.shape[0] == 6:
                # (latent tables: each rank touches its own images' rows; the other rows are zero on it)
                assert float((g_dp != 0).any(dim=1).sum()) == float((g_one != 0).any(dim=1).sum())
            torch.testing.assert_close(g_dp, g_one, rtol=2e-4, atol=2e-6 * max(scale, 1e-30))
            n_checked += 1
    assert n_checked == 18 + 6
    # ... and two optimiser steps later the parameters have moved alike (Adam / RMSprop normalise every entry: compare the bulk)
    moved = 0
    for k in end:
        if not end[k].dtype.is_floating_point or torch.equal(end[k], start[k]) or k.endswith(("weight_u", "weight_v")):
            continue
        da, db = (end[k] - start[k]).double().flatten(), (r0["end"][k] - start[k]).double().flatten()
        assert float((da - db).norm() / da.norm()) < 0.05, (k, float((da - db).norm() / da.norm()))
        moved += 1
    assert moved >= 18 + 6


def _tail_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    tdist.init_distributed("gloo")
    ps = [torch.nn.Parameter(torch.zeros(5, 3)), torch.nn.Parameter(torch.zeros(4))]
    red = tdist.FlatGradAllReducer(ps)
    seen = []
    for it in range(4):
        ps[0].grad = torch.full((5, 3), float(rank + 1 + it))
        ps[1].grad = None
        words = torch.zeros(3, dtype=torch.int32)
        if it == 1 and rank == 1:
            words[1] = 1                              # ONE rank raises a gate word in ONE iteration
        red.pack(flags=words)
        red.all_reduce()
        red.adopt()
        seen.append((red.gate_words.ne(0).tolist(), float(ps[0].grad[0, 0]), ps[1].grad is None, float(red.views[1].abs().sum())))
        if it == 2:
            red.clear_gate()                          # (what the trainer does once the host has acted on the words)
    torch.save(seen, os.path.join(out_dir, f"tail_rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_gate_words_in_the_tail_are_job_wide_and_sticky(tmp_path):
    """The tail of the flat buffer: a word raised by one rank in one iteration is non-zero on EVERY rank from that all-reduce on (the
    optimiser launches of both ranks are withheld together, and stay withheld) until it is cleared; the gradients are the average."""
    world = 2
    mp.spawn(_tail_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0, r1 = (torch.load(tmp_path / f"tail_rank{r}.pt") for r in range(world))
    assert r0 == r1
    assert [s[0] for s in r0] == [[False] * 4, [False, True, False, False], [False, True, False, False], [False] * 4]
    assert [s[1] for s in r0] == [1.5, 2.5, 3.5, 4.5] and all(s[2] and s[3] == 0.0 for s in r0)


def _guard_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from texpose_amd.options import AttrDict, default_options
    from texpose_amd.synthetic import training_batch
    from texpose_amd.trainer import GanTrainer
    tdist.init_distributed("gloo")
    opt = default_options(H=TH, W=TH, device="cpu")
    opt.batch_size, opt.patch_size, opt.nerf.sample_intvs = TB // world, 16, TN
    opt.loss_weight.feat = None
    torch.manual_seed(5)
    graph = _oracle_backed_graph(opt)
    graph.attach_latents(6, opt)
    graph.train()
    tdist.setup_data_parallel(graph, seed=7)
    tr = GanTrainer(opt, graph, n_train=6, max_iter=10)
    full = training_batch(TB, TH, TH, n_train=6, seed=3, device="cpu")
    mine = tdist.shard_training_batch(full, rank, world)
    tr.train_iteration(AttrDict(dict(mine)))                    # a clean iteration first
    start = {k: v.detach().clone() for k, v in graph.state_dict().items()}
    if rank == 1:
        mine.image = mine.image.clone()
        mine.image[0, 0, 0, 0] = float("nan")                   # only THIS rank's photometric loss is non-finite
        mine.image[:] = float("nan")
    raised = False
    try:
        tr.train_iteration(AttrDict(dict(mine)))
    except FloatingPointError:
        raised = True
    # (the spectral-norm vectors move in every discriminator FORWARD, as in the reference: not an update)
    same = all(torch.equal(v, start[k]) for k, v in graph.state_dict().items() if not k.endswith(("weight_u", "weight_v")))
    torch.save(dict(raised=raised, untouched=same), os.path.join(out_dir, f"guard_rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_step_gate_decision_is_global(tmp_path):
    """A non-finite loss on ONE rank: every rank raises FloatingPointError in the same iteration, before any gradient
    all-reduce or optimiser step -- nobody is left waiting in a collective, no rank applies gradients another one dropped."""
    world = 2
    mp.spawn(_guard_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        res = torch.load(tmp_path / f"guard_rank{r}.pt")
        assert res["raised"] and res["untouched"], (r, res)


def test_shard_training_batch_slices_only_per_sample_entries():
    from texpose_amd.options import AttrDict
    B = 3
    var = AttrDict(idx=torch.arange(B), image=torch.zeros(B, 3, 4, 4), intr=torch.zeros(B, 3, 3), pose=torch.zeros(B, 3, 4),
                   pose_anchor=torch.ones(B, 3, 4), K_shared=torch.eye(3), note="x")
    out = tdist.shard_training_batch(var, 1, 3)
    assert out.idx.tolist() == [1] and out.image.shape[0] == 1 and out.intr.shape == (1, 3, 3)
    assert out.pose_anchor.shape == (B, 3, 4) and out.K_shared.shape == (3, 3) and out.note == "x"    # passed through whole
    with pytest.raises(ValueError):
        tdist.shard_training_batch(AttrDict(idx=torch.arange(B), image=torch.zeros(B + 1, 3, 4, 4)), 0, 3)


def test_graphed_trainer_form_selection_for_several_ranks(monkeypatch):
    """What `world > 1` selects in GraphedGanTrainer (host logic, no GPU).  A configuration the linear graphs cover gets them WITH the
    collectives (`_dp`: gradients + pack | all-reduce | optimiser graph, gates = the tails of the flat buffers); anything else gets the
    generic two-graph form with the eager all-reduces between the replays; a single rank keeps the one-rank forms and the `_bad`
    snapshots as gates.  TP_NO_LINEAR_DP opts out of the linear graphs for several ranks."""
    from texpose_amd.options import default_options, AttrDict
    from texpose_amd.gan_modules import Discriminator
    from texpose_amd.graph import Graph
    from texpose_amd.trainer import FusedAdam, FusedRMSprop, GraphedGanTrainer
    for var in ("TP_SPLIT_GRAPH", "TP_NO_BRANCH_OVERLAP", "TP_NO_LINEAR_DP"):
        monkeypatch.delenv(var, raising=False)
    # (a) no discriminator: never linear
    opt = default_options(H=32, W=32, device="cpu")
    opt.loss_weight.feat = opt.loss_weight.gan_nerf = None
    opt.gan = None
    g = Graph(opt)
    g.attach_latents(4, opt)
    tr = GraphedGanTrainer(opt, g, n_train=4)
    batch = AttrDict(idx=torch.arange(2))
    assert not tr._split_around_collectives() and not tr._has_collective()          # one rank: one graph, no collective
    tr._select_form(batch)
    assert not (tr._linear or tr._dp)
    monkeypatch.setattr(tdist.FlatGradAllReducer, "world_size", property(lambda self: 2))
    assert tr._has_collective() and tr._split_around_collectives()                   # several ranks: A | reduce | B
    tr._select_form(batch)
    assert not (tr._linear or tr._dp)
    # (b) the full GAN iteration, as the GPU sees it (the predicates that need the HIP kernels are answered "yes" here)
    opt = default_options(H=32, W=32, device="cpu")
    g = Graph(opt, discriminator=Discriminator(opt))
    g.attach_latents(4, opt)
    tr = GraphedGanTrainer(opt, g, n_train=4)
    monkeypatch.setattr(GraphedGanTrainer, "_use_linear_graphs", lambda self, var: True)
    tr.optim_nerf = FusedAdam([dict(params=tr.nerf_group, lr=tr.lr_nerf_used)], capturable=True)
    tr.optim_disc = FusedRMSprop([dict(params=tr.disc_group, lr=tr.lr_disc_used)], capturable=True)
    tr._select_form(batch)
    assert tr._linear and tr._dp
    assert tr.optim_nerf.gate.data_ptr() == tr.red_nerf.gate_words.data_ptr() and tr.optim_nerf.gate.dtype == torch.int32
    assert tr.optim_disc.gate.data_ptr() == tr.red_disc.gate_words.data_ptr() and len(tr.optim_disc.gate) == 3
    assert tr._poll_words().data_ptr() == tr.red_nerf.gate_words.data_ptr()
    monkeypatch.setenv("TP_NO_LINEAR_DP", "1")
    knobs.reload()
    tr._select_form(batch)
    assert not (tr._linear or tr._dp) and tr.optim_nerf.gate is tr._gate_nerf      # the generic two-graph form
    monkeypatch.delenv("TP_NO_LINEAR_DP")
    knobs.reload()
    monkeypatch.setattr(tdist.FlatGradAllReducer, "world_size", property(lambda self: 1))
    tr._select_form(batch)
    assert tr._linear and not tr._dp and tr.optim_nerf.gate is tr._gate_nerf and tr.optim_disc.gate is tr._gate_disc
    assert tr._poll_words() is tr._bad
    # the job-wide gate state of the `_dp` form is read from the two tails
    tr._dp = True
    tr.red_disc.flag_tail[2] = 2.0                   # (two ranks raised word 2)
    assert tr._read_bad(blocking=True) == [0, 0, 1]


# ------------------------------------------------------------------------------------------ round 4: multi-GPU readiness without hardware
def _load_by_path(name, rel):
    import importlib.util
    spec = importlib.util.spec_from_file_location(name, os.path.join(REPO, rel))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _c5_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    emo = _load_by_path("eval_multi_object", "tools/eval_multi_object.py")
    tdist.init_distributed("gloo")
    be = emo.StubBackend(n_samples=4)
    line = emo.measure(torch.device("cpu"), rank, world, n_objects=5, images_per_object=3, n_samples=4, precision="f16x3",
                       warm=1, steps=2, backend=be)
    torch.save(dict(line=line, calls=be.calls, mine=emo.objects_of_rank(5, rank, world)), os.path.join(out_dir, f"c5_rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_c5_objects_shard_over_ranks_and_aggregate(tmp_path):
    """tools/eval_multi_object.measure (BASELINE C5) under gloo, world size 2, with the stub renderer: objects shard
    contiguously and disjointly (5 objects -> 3 + 2), every rank renders ONLY its objects' images, the per-object times
    land in their own slots (SUM over disjoint entries), kernel totals add up over the ranks, the wall time is the MAX over the
    ranks, and both ranks hold the identical line."""
    world = 2
    mp.spawn(_c5_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(tmp_path / f"c5_rank{k}.pt") for k in range(world)]
    assert r[0]["mine"] == [0, 1, 2] and r[1]["mine"] == [3, 4]
    for k in range(world):                                     # timed region: steps x (every image of every own object), in order
        assert r[k]["calls"] == [(o, i) for _ in range(2) for o in r[k]["mine"] for i in range(3)]
    l0, l1 = r[0]["line"], r[1]["line"]
    assert l0 == l1
    ms = [o["ms"] for o in l0["per_object"]]
    for o, v in enumerate(ms):                                 # stub: (o + 1) ms per render, 3 images per step
        assert 3.0 * (o + 1) <= v < 3.0 * (o + 1) + 6.0, (o, v)
    rays_obj = 240 * 320 * 2 + 480 * 640
    assert l0["config"]["rays_per_object"] == rays_obj and l0["n_gpus"] == 2
    assert l0["roofline"]["samples_all_ranks"] == 5 * rays_obj * 4              # all ranks' launches, per step
    assert l0["roofline"]["kernel_ms_total_all_ranks"] == 5 * 3 * 0.5
    # wall time per step = the slower rank's (rank 0: 3 x (1 + 2 + 3) ms, rank 1: 3 x (4 + 5) ms)
    assert 27.0 <= l0["ms_per_step"] < 45.0
    assert abs(l0["value"] - 5 * rays_obj / (l0["ms_per_step"] * 1e-3)) < 1e-6 * l0["value"]


def _fallback_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    bench = _load_by_path("bench_mod", "bench.py")
    tdist.init_distributed("gloo")
    calls = []

    def measure(graphed):
        calls.append(graphed)
        if graphed:
            if rank == 1:
                raise RuntimeError("capture failed on this rank only")
            return dict(value=1.0, launch="graph")             # (rank 0's captured loop "worked": no collective left pending)
        t = torch.tensor([float(rank + 1)])
        dist.all_reduce(t)                                     # the eager loop's collectives: BOTH ranks must be in here
        return dict(value=float(t), launch="eager")

    res, note = bench.captured_or_eager(measure, torch.device("cpu"), world, graphed=True)
    torch.save(dict(calls=calls, res=res, note=note), os.path.join(out_dir, f"fb_rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_failed_capture_on_one_rank_sends_every_rank_to_the_eager_loop(tmp_path):
    """bench.train_leg's fallback (bench.captured_or_eager): the capture fails on rank 1 only; the "somebody failed" all-reduce
    makes BOTH ranks drop the captured result and run the eager loop, whose collective they then meet in."""
    world = 2
    mp.spawn(_fallback_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(tmp_path / f"fb_rank{k}.pt") for k in range(world)]
    for k in range(world):
        assert r[k]["calls"] == [True, False] and r[k]["res"] == dict(value=3.0, launch="eager")
    assert "capture failed" in r[1]["note"] and "another rank" in r[0]["note"]


def _probe_order_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from texpose_amd import trainer as ttrainer
    tdist.init_distributed("gloo")
    before = tdist.warm_collective_done()
    seen_at_probe = []

    def fake_probe(dev, n):
        # what the real probe (distinct_queue_streams) would find: the job's first collective has already run
        seen_at_probe.append((tdist.warm_collective_done(), tdist.ranks_seen()))
        return ["stream%d" % i for i in range(n)]

    streams = ttrainer.streams_after_collectives(torch.device("cpu"), 3, None, probe=fake_probe)
    torch.save(dict(before=before, seen_at_probe=seen_at_probe, streams=streams, ranks_seen=tdist.ranks_seen()),
               os.path.join(out_dir, f"po_rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_queue_probe_runs_behind_the_first_collective_and_ranks_seen_counts_the_job(tmp_path):
    """Several ranks: the stream -> hardware-queue probe of the captured step (trainer.streams_after_collectives) runs only AFTER the
    communicator's first collective (RCCL makes its own streams there), and that collective -- dist.ranks_seen, a SUM of ones --
    reports the number of ranks the communicator joined (what every multi-GPU line of bench.py / tools carries)."""
    world = 2
    mp.spawn(_probe_order_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for k in range(world):
        r = torch.load(tmp_path / f"po_rank{k}.pt")
        assert r["before"] is False                                # init_process_group alone is not the warm collective
        assert r["seen_at_probe"] == [(True, world)] and r["ranks_seen"] == world
        assert r["streams"] == ["stream0", "stream1", "stream2"]


def test_ranks_seen_is_one_without_a_process_group():
    assert not dist.is_initialized()
    assert tdist.ranks_seen() == 1 and tdist.warm_collective_done()
