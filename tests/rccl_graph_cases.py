"""Training-iteration cases with RCCL collectives (1-rank communicator, collectives forced), shared by two GPU tests.  The collectives
are stream-ordered calls BETWEEN graph replays in every case: capturing them into a hipGraph (an opt-in up to round 6) was removed --
ProcessGroupNCCL's watchdog thread polls the work event the captured call recorded and aborts the process ("operation not permitted
on an event last recorded in a capturing stream"), about one run in five on this stack."""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def ensure_group():
    """A 1-rank NCCL (= RCCL) process group on cuda:0; True if this call made it."""
    import socket
    import torch.distributed as dist
    if dist.is_initialized():
        return False
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=torch.device("cuda", 0))
    return True


def run_linear(mode):
    """Eight iterations of the full GAN step (B=4, C3 size) in one of: "one_rank" (no collective), "between" (the several-rank form:
    all-reduces as stream-ordered calls between the replays), "between_pipelined" (+ pipeline_disc_tail / defer_results).
    -> (dict(state, loss, optim, launches), objects to keep alive)."""
    from texpose_amd import knobs
    from texpose_amd.gan_modules import Discriminator, PerceptualLoss
    from texpose_amd.graph import Graph
    from texpose_amd.options import default_options, AttrDict
    from texpose_amd.synthetic import training_batch
    from texpose_amd.trainer import GraphedGanTrainer
    knobs.reload()
    try:
        torch.manual_seed(0)
        opt = default_options(H=128, W=128, device="cuda:0")
        opt.batch_size, opt.patch_size, opt.nerf.sample_intvs = 4, 16, 64
        graph = Graph(opt, discriminator=Discriminator(opt), perceptual_loss=PerceptualLoss()).to("cuda:0")
        tr = GraphedGanTrainer(opt, graph, n_train=189)
        forced = mode != "one_rank"
        tr.red_nerf.single_rank_collective = tr.red_disc.single_rank_collective = forced
        tr.pipeline_disc_tail = tr.defer_results = mode == "between_pipelined"
        batches = [training_batch(4, 128, 128, seed=s_, device="cuda:0") for s_ in range(2)]
        for it in range(8):
            _, loss = tr.train_iteration(AttrDict(dict(batches[it % 2])))
        assert tr._linear and tr._dp == forced and "D2a" in tr._graphs
        assert ("G2c" in tr._graphs and "D2c" in tr._graphs) == forced
        assert tr.finish() == [0, 0, 0]
        torch.cuda.synchronize()
        if forced:
            assert tr.optim_nerf.gate.data_ptr() == tr.red_nerf.gate_words.data_ptr()
            assert int(tr.red_nerf.gate_words.abs().sum()) == 0 and int(tr.red_disc.gate_words.abs().sum()) == 0
            # the averaged gradients live in the flat buffers: the optimisers read them there
            assert all(p.grad is None or p.grad.data_ptr() == v.data_ptr() for p, v in zip(tr.red_disc.params, tr.red_disc.views))
            assert sum(p.grad is not None for p in tr.red_disc.params) == 6 and graph.discriminator.progress.grad is None
        counts = dict(tr.launch_counts)
        assert all(v is not None and v > 0 for v in counts.values()), counts
        res = dict(state={k: v.detach().cpu().clone() for k, v in graph.state_dict().items()},
                   loss={k: v.detach().cpu().clone() for k, v in loss.items() if torch.is_tensor(v)},
                   optim={"%d.%d.%s" % (oi, pi, name): t.detach().cpu().clone() for oi, o in enumerate((tr.optim_nerf, tr.optim_disc))
                          for pi, p in enumerate(q for gr in o.param_groups for q in gr["params"]) if p in o.state
                          for name, t in o.state[p].items() if torch.is_tensor(t)},
                   launches=sum(counts.values()))
        # the host side of the gate in this form: a word raised on the device (here: by hand) reaches the host through the copy the
        # form queues -- one rank: the step-input launch; several ranks: the nerf buffer's tail one pack later, copied out on the
        # discriminator stream when the results are deferred -- and the call that sees it raises
        tr._bad[1] = 1
        raised = None
        for extra in range(5):
            try:
                if extra < 4:
                    tr.train_iteration(AttrDict(dict(batches[extra % 2])))
                else:
                    tr.finish()             # (the host may be iterations ahead of the copies it polls: the blocking read at the end)
            except FloatingPointError:
                raised = extra
                break
        assert raised is not None and (raised >= 1 or not forced), raised           # (several ranks: the word travels through a pack first)
        assert "_poll_on_side" not in tr.__dict__
        res["raised_after"] = raised
        return res, (tr, graph)
    finally:
        knobs.reload()


def run_generic(forced, split):
    """Two iterations of the GENERIC captured form (TP_NO_LINEAR_DP=1, no feature loss, 32x32 crops): no collective (one graph; with
    ``split`` two), or the all-reduces forced -- then always eagerly between two replays.  -> (dict(state, snap), objects)."""
    from oracle import texpose_oracle as O
    from texpose_amd import knobs
    from texpose_amd.gan_modules import Discriminator
    from texpose_amd.graph import Graph
    from texpose_amd.options import default_options, AttrDict
    from texpose_amd.synthetic import training_batch
    from texpose_amd.trainer import GraphedGanTrainer
    os.environ["TP_NO_LINEAR_DP"] = "1"
    os.environ.pop("TP_SPLIT_GRAPH", None)
    if split:
        os.environ["TP_SPLIT_GRAPH"] = "1"
    knobs.reload()
    try:
        dev = torch.device("cuda", 0)
        B, H, W, N = 2, 32, 32, 8
        batch = training_batch(B, H, W, n_train=5, seed=2, device="cuda:0")
        gen = torch.Generator().manual_seed(12)
        rnd = (torch.rand(3, B, 1, 1, 1, generator=gen).to(dev), torch.rand(B, 256, N, 1, generator=gen).to(dev))
        opt = default_options(H=H, W=W, device="cuda:0")
        opt.batch_size, opt.patch_size, opt.nerf.sample_intvs = B, 16, N
        opt.loss_weight.feat = None
        graph = Graph(opt, discriminator=Discriminator(opt)).to(dev)
        graph.nerf.load_state_dict({**graph.nerf.state_dict(), **{k: v.to(dev) for k, v in O.make_params(6).items()}})
        dcpu = Discriminator(opt)
        O.seed_spectral_module(dcpu, 10)
        graph.discriminator.load_state_dict(dcpu.state_dict())
        graph.train()
        graph.nerf.precision = "fp32"
        tr = GraphedGanTrainer(opt, graph, n_train=5)
        with torch.no_grad():
            graph.latent_vars_trans.weight.fill_(0.1)
            graph.latent_vars_light.weight.fill_(-0.2)
        tr.red_nerf.single_rank_collective = tr.red_disc.single_rank_collective = forced
        snap = {k: v.detach().clone() for k, v in graph.state_dict().items()}
        ex = AttrDict(dict(batch))
        ex.patch_u, ex.jitter_rand = rnd
        tr.capture(ex, warmup=2)
        graph.load_state_dict(snap)
        for o in (tr.optim_nerf, tr.optim_disc):
            for st in o.state.values():
                for v in st.values():
                    if torch.is_tensor(v):
                        v.zero_()
        graph.nerf.mark_heads_dirty()
        for _ in range(2):
            v = AttrDict(dict(batch))
            v.patch_u, v.jitter_rand = rnd
            _, loss = tr.train_iteration(v)
        assert all(np.isfinite(float(x)) for x in loss.values())
        assert (tr._graph_b is not None) == (split or forced) and tr._linear == (not (forced or split)) and not tr._dp
        torch.cuda.synchronize()
        return dict(state={k: v.detach().cpu().clone() for k, v in graph.state_dict().items()},
                    snap={k: v.cpu() for k, v in snap.items()}), (tr, graph)
    finally:
        os.environ.pop("TP_SPLIT_GRAPH", None)
        os.environ.pop("TP_NO_LINEAR_DP", None)
        knobs.reload()
