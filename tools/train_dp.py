#!/usr/bin/env python3
"""Data-parallel training entry point (BASELINE config C4: Duck train, global batch 32 over 8 GPUs; any N works):

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        tools/train_dp.py [--global-batch 32] [--iters 50] [--eager]

One process per GPU (the launcher starts them; nothing re-executes after the GPU is initialised).  Each rank:
init_distributed -> Graph + PatchGAN + feature network -> setup_data_parallel (broadcast rank 0's parameters and
buffers, per-rank random streams) -> its shard of every global batch -> GraphedGanTrainer, whose captured step ends
each optimiser step's backward passes with ONE flat RCCL all-reduce (1.7 MB nerf step, 10.7 MB discriminator step).
Rank 0 prints one JSON line: training iterations / s of the whole job (max over ranks of the timed loop).
The reference has no counterpart (single GPU by assertion, options.py:112); the iteration itself is
model/nerf_adapt_st_gan.py:108-202.  Synthetic data (no dataset offline), random-init VGG19[:15] feature network."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                                                    # noqa: E402


def build(device, per_rank_batch, full=True, train_precision="f16x3", graphed=True, group=None):
    from texpose_amd.gan_modules import Discriminator, PerceptualLoss
    from texpose_amd.graph import Graph
    from texpose_amd.options import default_options
    from texpose_amd.trainer import GanTrainer, GraphedGanTrainer
    opt = default_options(H=128, W=128, device=str(device))
    opt.batch_size, opt.patch_size, opt.nerf.sample_intvs = per_rank_batch, 16, 64
    if not full:
        opt.loss_weight.feat = None
        opt.loss_weight.gan_nerf = None
        opt.gan = None
    torch.manual_seed(0)                                   # same construction-time draws everywhere (then broadcast)
    graph = Graph(opt, discriminator=Discriminator(opt) if full else None,
                  perceptual_loss=PerceptualLoss() if full else None).to(device)
    graph.nerf.train_precision = train_precision
    graph.train()
    if full:
        graph.perceptual_loss.model.eval()
    graph.attach_latents(189, opt)
    return opt, graph, (GraphedGanTrainer if graphed else GanTrainer)


def measure(device, rank, world, global_batch=32, iters=50, warm=5, graphed=True, full=True, train_precision="f16x3",
            seed=0, hook=None, force_collectives=False):
    """Timed training loop of an already initialised job (world == 1: no process group needed).  ``hook(trainer)`` may
    instrument the trainer before the warm-up (bench.py times the training kernels with HIP events that way)."""
    from texpose_amd import dist as tdist
    from texpose_amd.options import AttrDict
    from texpose_amd.synthetic import training_batch
    torch.backends.cudnn.benchmark = os.environ.get("TP_MIOPEN_FIND", "1") == "1"
    if global_batch % world:
        raise SystemExit("global batch %d is not divisible by %d ranks" % (global_batch, world))
    opt, graph, cls = build(device, global_batch // world, full, train_precision, graphed)
    tdist.setup_data_parallel(graph, seed=seed)
    trainer = cls(opt, graph, n_train=189)
    if force_collectives:
        # ONE rank with the several-rank step: real RCCL calls in a 1-rank communicator (bench.py `train.c4_form`: what the form costs
        # before a byte crosses xGMI)
        trainer.red_nerf.single_rank_collective = True
        if trainer.red_disc is not None:
            trainer.red_disc.single_rank_collective = True
    if hasattr(trainer, "pipeline_disc_tail") and os.environ.get("TP_NO_PIPELINE_DISC") != "1":
        # the discriminator step's second half may run beside the next render (one rank, six-graph form); this loop reads losses and
        # state only behind flush_flags() / a device synchronise
        trainer.pipeline_disc_tail = True
        trainer.defer_results = os.environ.get("TP_NO_DEFER") != "1"
    if hook is not None:
        hook(trainer)
    # a global batch, identical on all ranks (stands in for the sampler of a distributed data loader); each rank keeps its shard
    batches = [tdist.shard_training_batch(training_batch(global_batch, 128, 128, seed=s, device=device), rank, world)
               for s in range(2)]

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for i in range(warm):
        trainer.train_iteration(AttrDict(dict(batches[i % 2])))
    barrier()
    t0 = time.perf_counter()
    for i in range(iters):
        _, loss = trainer.train_iteration(AttrDict(dict(batches[i % 2])))
    barrier()
    dt = torch.tensor([time.perf_counter() - t0], device=device, dtype=torch.float64)
    if world > 1:
        torch.distributed.all_reduce(dt, op=torch.distributed.ReduceOp.MAX)
    dt = float(dt) / iters
    # HIP events around the step's gradient all-reduces (the xGMI figure), in a short loop of its own BEHIND the timed one: the event
    # records sit between graph replays and would cost the timed loop a few microseconds each
    coll = coll_by = None
    if graphed and (getattr(trainer, "_graph_b", None) is not None or getattr(trainer, "_dp", False)):
        trainer.collective_events = []
        for i in range(min(iters, 20)):
            trainer.train_iteration(AttrDict(dict(batches[i % 2])))
        barrier()
        by = {}
        for name, a, b in trainer.collective_events:
            by.setdefault(name, []).append(a.elapsed_time(b))
        trainer.collective_events = None
        coll_by = {k: sum(v) / len(v) for k, v in by.items()}
        coll = sum(coll_by.values())                       # (every name occurs once per iteration)
    if hasattr(trainer, "flush_flags"):
        trainer.flush_flags()                              # (the last replay's gate words: a withheld final step raises here)
    counts = getattr(trainer, "launch_counts", None)
    return dict(collective_ms=coll, collective_ms_by_step=coll_by, ranks_seen=tdist.ranks_seen(), form=_form(trainer, graphed),
                # nodes of the captured graphs of ONE iteration + the tp_step_inputs launch in front of them (None: eager loop)
                launches=(sum(v for v in counts.values() if v) + 1) if counts and all(v is not None for v in counts.values()) else None,
                launch_counts=counts, queues=getattr(trainer, "queue_probe", None),
                metric="train iters/sec", value=1.0 / dt, ms_per_iter=dt * 1e3, n_gpus=world, global_batch=global_batch,
                per_gpu_batch=global_batch // world, rays_per_iter=global_batch * 256, samples_per_iter=global_batch * 256 * 64,
                launch=("linear hipGraph replays on three streams with one flat all-reduce between each gradient graph and its optimiser graph (render | D(fake) + its backward | generator backward, pack | all-reduce | Adam || feature chain || spectral norm | discriminator step, pack | all-reduce | RMSprop)" if getattr(trainer, "_dp", False)
                        else "six linear hipGraph replays on three streams (render | D(fake) + its backward | generator backward + Adam || feature chain || spectral norm | discriminator step)" if getattr(trainer, "_linear", False)
                        else "hipGraph replay" if getattr(trainer, "_graph_b", None) is None
                        else "two hipGraph replays with the gradient all-reduces between them") if graphed else "eager", recording_forward=graph.nerf.train_precision,
                collective="one flat all-reduce per optimiser step (%.1f MB nerf, %.1f MB discriminator)"
                           % (trainer.red_nerf.nbytes / 1e6, (trainer.red_disc.nbytes if trainer.red_disc else 0) / 1e6),
                loop="full GAN (render fwd+bwd, gathers, random-init VGG19[:15] feature loss, PatchGAN + R1, Adam + RMSprop)"
                if full else "nerf step only (render fwd+bwd, photometric/uncert/trans_reg losses, Adam)",
                finite=all(bool(torch.isfinite(v)) for v in loss.values() if torch.is_tensor(v)),
                skipped_steps=trainer.skipped_steps)


def _form(trainer, graphed):
    if not graphed:
        return "eager"
    if getattr(trainer, "_dp", False):
        return "linear_dp"
    if getattr(trainer, "_linear", False):
        return "linear"
    return "one_graph" if getattr(trainer, "_graph_b", None) is None else "two_graphs_eager_collectives"


def run(global_batch=32, iters=50, warm=5, graphed=True, full=True, train_precision="f16x3", seed=0):
    from texpose_amd import dist as tdist
    rank, world, local = tdist.init_distributed()
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    out = measure(device, rank, world, global_batch, iters, warm, graphed, full, train_precision, seed)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--global-batch", type=int, default=32)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--eager", action="store_true")
    ap.add_argument("--nerf-only", action="store_true")
    ap.add_argument("--train-precision", default="f16x3")
    a = ap.parse_args()
    run(a.global_batch, a.iters, a.warmup, graphed=not a.eager, full=not a.nerf_only, train_precision=a.train_precision)
