"""The GAN training iteration captured into hipGraphs and replayed (reference ``Model.train_iteration``,
model/nerf_adapt_st_gan.py:108-202): `GraphedGanTrainer`, the captured twin of `texpose_amd.trainer.GanTrainer`.  The step forms, what
bounds the iteration and the several-rank form are described in DESIGN.md sections 5 and 6."""
from __future__ import annotations

import copy
import warnings

import torch

from . import knobs, ops
from .graph import Graph
from .options import AttrDict
from .queue_probe import LAST_QUEUE_PROBE, streams_after_collectives
from .trainer import FusedAdam, FusedRMSprop, GanTrainer


class _FormUnavailable(Exception):
    """Raised inside a warm-up iteration when the selected multi-graph form does not cover this configuration after all."""


class GraphedGanTrainer(GanTrainer):
    """The same iteration captured ONCE into a hipGraph and replayed.

    A training iteration at the reference's batch sizes is launch-bound: ~1,350 kernel launches (autograd through the
    PatchGAN incl. the R1 double backward, spectral-norm power iterations, VGG, optimisers) for ~8 ms of GPU work
    (profiles/r1).  Everything in it is static in shape, so the whole step -- patch coordinates, stratified jitter,
    render forward / backward (the ctypes launches enqueue on the capturing stream), gathers, both optimiser steps,
    the data-parallel all-reduces -- is recorded once and replayed with one launch.

    Per-iteration host state goes through device memory: the batch is copied into static input tensors, the annealed
    patch-scale bound is a 0-dim device tensor, the jitter comes from torch's graph-safe Philox stream, the optimisers
    are ``capturable``.  Losses come back as static tensors (read them only when logging: that is the one sync).

    Two captured forms (`_select_form`): the LINEAR graphs on three streams (the full GAN iteration; with several ranks each
    optimiser launch is a graph of its own behind ONE flat all-reduce), and a generic form for every other configuration (one graph
    with its branches forked inside; with several ranks gradients | eager all-reduces | optimisers).

    The step gate across the two optimisers is best-effort in time: a discriminator step's non-finite flag withholds the NEXT nerf step
    at the latest (with `pipeline_disc_tail` the next render may already be running when the flag is written; in strict mode the same
    step's Adam launch may have snapshotted the words before the discriminator branch wrote its own) -- each optimiser's own flags
    always gate its own step, and once a word is set every later step of both is withheld until the host has acted.

    By default `train_iteration` returns with the calling stream ordered behind everything the iteration enqueued.  Two opt-in
    attributes relax that for throughput (the linear graphs; bit-identical results either way): ``defer_results`` (the
    calling stream is ordered behind the consumption of the iteration's INPUTS only) and ``pipeline_disc_tail`` (the discriminator
    step's second half may still run beside the next render).  With either set, order the calling stream behind the results with
    `wait_all()` -- or `finish()` / `flush_flags()`, which also read the gate words -- BEFORE reading losses, parameters or optimiser
    state, saving a checkpoint or loading one.
    """
    capturable = True

    def __init__(self, opt, graph: Graph, n_train: int, max_iter: int = 6000 * 189 // 8, group=None):
        super().__init__(opt, graph, n_train, max_iter, group=group)
        self._graph = None
        self._graph_b = None
        self._deferred = False
        self._static_in = None
        self._static_loss = None
        dev = self.lr_nerf.device
        # sticky device words of the step gate: [range flag, non-finite nerf loss, non-finite discriminator loss] seen; the
        # gates are 1 only while all are 0 (separate words: the two branches of the captured step never write the same one)
        self._bad = torch.zeros(3, dtype=torch.int32, device=dev)
        self._side = None                        # second stream of the captured step (discriminator branch)
        # Opt-in: the discriminator step's second half (R1 passes, backward, RMSprop) of iteration i may run beside the render of
        # iteration i + 1 -- every dependency is an event (the next render waits for the first half, the next spectral norm for the
        # second), so the arithmetic is unchanged; but the CALLING stream no longer waits for it at the end of `train_iteration`:
        # discriminator losses / weights / optimiser state may be read only behind `wait_all()` or `finish()` (tools/train_dp.py,
        # bench.py and tools/train_bench.py set it; default off = everything ordered on the calling stream).
        self.pipeline_disc_tail = knobs.K.pipeline_disc
        # opt-in: `train_iteration` returns with the calling stream ordered behind the consumption of its inputs, not behind its results
        # (read losses / parameters behind `wait_all()`); the linear graphs only (`_replay_linear`)
        self.defer_results = knobs.K.defer_results
        self._d2_pending = False
        self._graphs, self._events = None, None  # the captured linear graphs by name, the events that order them across streams
        self._linear = False                     # the step as LINEAR graphs on three streams (`_use_linear_graphs`)
        self._dp = False                         # ... with a gradient all-reduce between each gradient graph and its optimiser graph
        # what the optimiser launches read: the words as they stood when the step's own flags had been folded in
        self._gate_nerf, self._gate_disc = torch.zeros_like(self._bad), torch.zeros_like(self._bad)
        if isinstance(self.optim_nerf, FusedAdam):
            self.optim_nerf.gate = self._gate_nerf
        if self.has_disc and isinstance(self.optim_disc, FusedRMSprop):
            self.optim_disc.gate = self._gate_disc
        self._bad_poll = None

    # the guards run INSIDE the captured step: no host read, the update is multiplied by the 0/1 gate
    def _gate(self, ok, grads, lr, lr_used):
        """grads <- ok ? grads : 0 (non-finite entries included) and lr_used <- ok ? lr : 0, in four launches whatever the
        number of parameters: one flat copy, one select, one scatter back, one scalar product."""
        grads = [g for g in grads if g is not None]
        flat = torch.cat([g.reshape(-1) for g in grads])
        flat = torch.where(ok, flat, torch.zeros((), device=flat.device))
        torch._foreach_copy_(grads, [v.view_as(g) for v, g in zip(flat.split([g.numel() for g in grads]), grads)])
        lr_used.copy_(lr * ok.to(torch.float32))

    def _flag_nerf(self, loss):
        """Before the backward (and before the branches of the step fork): fold this forward's range flag and the finiteness
        of its loss into the sticky words and snapshot them as the nerf gate (which so also sees a discriminator flag of
        EARLIER steps, never a concurrent write of this one).  One launch (K13 tp_step_flags)."""
        status = ops.mlp_status(loss.all.device) if self._uses_f16x3() else None
        ops.step_flags(loss.all, self._bad, 1, self._gate_nerf, status=status, word_status=0)

    def _guard_nerf(self, var, loss):
        if getattr(self.optim_nerf, "gate", None) is None:       # a stock optimiser: zero the gradients and the rate instead
            self._gate(self._gate_nerf.sum() == 0, [p.grad for p in self.nerf_group], self.lr_nerf, self.lr_nerf_used)
        return True

    def _flag_disc(self, total):
        """Fold the finiteness of the discriminator loss into the sticky words and snapshot them as the discriminator gate."""
        ops.step_flags(total, self._bad, 2, self._gate_disc)

    def _disc_gate_flags(self):
        return dict(bad=self._bad, word_finite=2, snapshot=self._gate_disc)

    def _guard_disc(self, total=None):
        if total is not None:
            self._flag_disc(total)
        if getattr(self.optim_disc, "gate", None) is None:
            self._gate(self._gate_disc.sum() == 0, [p.grad for p in self.disc_group], self.lr_disc, self.lr_disc_used)
        return True

    def _has_collective(self):
        """A gradient all-reduce is part of the step (several ranks, or forced in a 1-rank group by the tests)."""
        reds = [r for r in (self.red_nerf, self.red_disc) if r is not None]
        return any(r.world_size > 1 or r.single_rank_collective for r in reds)

    def _split_around_collectives(self):
        """Generic form: the gradient all-reduces stay OUTSIDE the captured graphs (replay A = everything up to the gradients, two
        eager collectives, replay B = the optimiser steps) -- whenever a collective is part of the step.  Between two replays they
        are ordinary stream-ordered calls.  (Capturing the RCCL calls INTO a graph was an opt-in up to round 6 and is gone:
        ProcessGroupNCCL's watchdog thread polls the work event that the captured call recorded and aborts the process with "operation
        not permitted on an event last recorded in a capturing stream" -- about one run in five on this stack.)"""
        return self._has_collective() or knobs.K.split_graph

    def _reduce_all(self):
        """The step's collectives, in ONE fixed order on every rank, after both branches have joined.  The sticky gate words
        ride in the tail of the first buffer: afterwards a word is set on every rank if any rank set it, so all ranks
        withhold (and later re-capture or raise) together."""
        self.red_nerf.reduce(flags=self._bad)
        if self.red_disc is not None:
            self.red_disc.reduce()

    def _body(self, var):
        """One iteration (what is captured as ONE graph; `_body_a` / `_body_b` when the collectives stay outside; `_linear_eager` is
        the same iteration with the segments and dependencies of the linear graphs)."""
        if self._linear:
            return self._linear_eager(var)
        out = self._body_a(var)
        if self._deferred:
            self._reduce_all()
            self._body_b()
        return out

    def _body_a(self, var):
        """After the render and its losses the step forks (reference order kept where it matters: the nerf
        step's discriminator forward -- its power iteration -- comes first): the generator branch (feature-network and
        discriminator backward, composite / MLP backward, Adam) stays on the capturing stream, the discriminator branch
        (real / fake forward, R1 double backward, RMSprop) runs on a second stream.  They share no written state -- nerf
        vs discriminator parameters, gradients, optimiser moments, separate gate words -- and in the replayed graph the
        launch gaps of one chain of small dependent kernels are filled by the other: 2.50 -> 2.30 ms at B=4.
        With a gradient all-reduce in the step (`_deferred`) the branches only produce gradients and flags; reductions and
        both optimiser steps follow after the join (`_reduce_all`, `_body_b`)."""
        opt = self.opt
        dev = var.idx.device
        overlap = self.has_disc and not knobs.K.no_branch_overlap
        if overlap and self._side is None:
            self._side = torch.cuda.Stream(device=dev)
        if overlap and getattr(self.graph, "feat_stream", None) is None and not knobs.K.no_feat_branch:
            # third chain of the replayed step: the feature network's forward and backward (graph.Graph._feature_loss_early)
            self.graph.feat_stream = torch.cuda.Stream(device=dev)
        self._prefetch_spectral_weights(var)                         # (first: it runs beside everything up to the render)
        B, R = opt.batch_size, opt.patch_size ** 2
        # the step counter of the in-kernel random draws belongs to THIS step function (its loss-total launch advances it): the
        # graph object and its patch sampler see it only while the step is being issued, so that an eager trainer driving the same
        # Graph afterwards draws from torch's generator as before instead of repeating one jitter pattern
        counter = getattr(self, "_rng_counter", None)
        self.graph.step_counter = self.graph.patch_sampler.device_counter = counter
        try:
            var = self.graph.get_ray_idx(opt, var)
            if opt.nerf.sample_stratified and "jitter_rand" not in var and counter is None:
                var.jitter_rand = torch.rand(B, R, opt.nerf.sample_intvs, 1, device=var.ray_idx.device)   # (TP_TORCH_RNG=1)
            var, loss = self.nerf_forward_loss(var)
        finally:
            self.graph.step_counter = self.graph.patch_sampler.device_counter = None
        # loss total + step gate in one launch: this forward's range flag and the finiteness of its loss go into the sticky words,
        # which are snapshot as the nerf gate BEFORE the backward and before the branches fork (so the gate also sees a
        # discriminator flag of earlier steps, never a concurrent write of this one)
        status = ops.mlp_status(dev) if self._uses_f16x3() else None
        terms, ws = self._weighted_total(loss, flags=dict(bad=self._bad, word_finite=1, snapshot=self._gate_nerf, status=status,
                                                          word_status=0, step_counter=getattr(self, "_rng_counter", None)))
        dloss = None
        # optimiser steps (and reductions) after this function: whenever a collective is part of the step, or on request
        self._deferred = self._has_collective() or self._split_around_collectives()

        def generator_backward():
            torch.autograd.backward(terms, ws)
            if not self._deferred:
                self._guard_nerf(var, loss)
                self.nerf_apply()

        if overlap:
            main = torch.cuda.current_stream(var.rgb.device)
            self._side.wait_stream(main)                          # fork
            with torch.cuda.stream(self._side):
                var, dloss = self.disc_step(var, apply=not self._deferred)
        generator_backward()
        if overlap:
            main.wait_stream(self._side)                          # join
        elif self.has_disc:
            var, dloss = self.disc_step(var, apply=not self._deferred)
        if dloss is not None:
            loss.update({k: v for k, v in dloss.items() if k != "all"})
        return {k: v.detach() for k, v in loss.items() if torch.is_tensor(v)}

    # ------------------------------------------------------------------ which form
    def _use_linear_graphs(self, var):
        """The full GAN iteration of the reference's configuration: PatchGAN with the explicit schedule (K16) and prefetched spectral
        norms, the generator's GAN term, patch rays.  The feature chain is optional (its graph F is left out without it)."""
        opt, lw = self.opt, self.opt.loss_weight
        if not (knobs.K.linear_graphs and self.has_disc and opt.gan is not None and lw.gan_nerf is not None and bool(opt.nerf.rand_rays)):
            return False
        if knobs.K.no_sn_prefetch or knobs.K.no_branch_overlap or knobs.K.no_feat_branch:
            return False
        if lw.feat is not None:
            pl = getattr(self.graph, "perceptual_loss", None)
            if pl is None or not (hasattr(pl, "loss_from_patches") and hasattr(pl, "pairs_from_patches")):
                return False
        p, B = int(opt.patch_size), len(var.idx)
        probe = torch.empty(0, device=var.idx.device).new_empty((B, 0, p, p))
        disc = self.graph.discriminator
        return self._disc_schedule(probe) is not None and hasattr(disc, "prefetch_spectral_weights") and disc.training

    def _set_wgrad_share(self):
        nerf, dev = self.graph.nerf, self._bad.device
        cus = torch.cuda.get_device_properties(dev).multi_processor_count if dev.type == "cuda" else 0
        nerf.wgrad_cus = int(cus * self.WGRAD_CU_SHARE) if (self._linear and cus and not knobs.K.wgrad_all_cus) else 0

    def _select_form(self, var):
        """Which captured form this configuration gets: the linear graphs (several ranks: the same with every optimiser launch as a
        graph of its own behind its all-reduce, `_dp`), or ONE graph (`_body`; with several ranks: gradients | eager all-reduces |
        optimiser steps).  Also points the fused optimisers at the gate words of that form."""
        collective = self._has_collective() or self._split_around_collectives()
        self._linear = self._use_linear_graphs(var)
        fused = isinstance(self.optim_nerf, FusedAdam) and self.has_disc and isinstance(self.optim_disc, FusedRMSprop)
        self._dp = bool(collective and self._linear and fused and not knobs.K.no_linear_dp)
        if collective and not self._dp:
            self._linear = False
        self._point_gates()
        self._set_wgrad_share()

    def _point_gates(self):
        """`_dp`: the optimiser launches read the (job-wide, sticky) tail of their all-reduce buffer; else the snapshots of `_bad`."""
        if isinstance(self.optim_nerf, FusedAdam):
            self.optim_nerf.gate = self.red_nerf.gate_words[:len(self._bad)] if self._dp else self._gate_nerf
        if self.has_disc and isinstance(self.optim_disc, FusedRMSprop):
            self.optim_disc.gate = self.red_disc.gate_words[:len(self._bad)] if self._dp else self._gate_disc

    def _seg_sn(self, part=None):
        """The three spectral normalisations of an iteration; `part` 0 / 1: the first one alone (the generator's pass through the frozen
        discriminator waits for nothing else) / the other two."""
        disc = self.graph.discriminator
        if part is None:
            disc.prefetch_spectral_weights(3)
        else:
            disc.prefetch_spectral_weights(1 if part == 0 else 2, append=part == 1)
        # (the consumers run in OTHER graphs / on the other stream: their ordering behind this segment is the ev_sn event of the
        # replay loop, not a wait recorded while one of them is being captured)
        disc._sn_queue = [(o, sg, u, v, None) for o, sg, u, v, _ in disc._sn_queue]

    def _seg_render(self, var):
        opt = self.opt
        counter = getattr(self, "_rng_counter", None)
        self.graph.step_counter = self.graph.patch_sampler.device_counter = counter
        # the patch coordinates and the latent rows ride in the ray-generation launch (Graph.render, tp_raygen_train)
        self.graph.fuse_prologue = not knobs.K.no_fused_prologue
        try:
            var = self.graph.get_ray_idx(opt, var)
            if opt.nerf.sample_stratified and "jitter_rand" not in var and counter is None:
                B, R = opt.batch_size, opt.patch_size ** 2
                var.jitter_rand = torch.rand(B, R, opt.nerf.sample_intvs, 1, device=var.ray_idx.device)
            var, _ = self.nerf_forward_loss(var, stage="render")
        finally:
            self.graph.step_counter = self.graph.patch_sampler.device_counter = None
            self.graph.fuse_prologue = False
        return var

    def _seg_disc(self, var):
        out = self.disc_step(var, apply=not self._dp)
        if self._dp:
            self._seg_disc_pack()
        return out

    def disc_step_zero_grads(self):
        self._toggle(self.graph.discriminator, True)
        self.optim_disc.zero_grad(set_to_none=True)

    def _seg_disc_a(self, var, run=False):
        """First half of the discriminator step as a segment of its own (see `_capture_linear`).  ``run=False``: only whether the split
        applies (the paired schedule covers this step) -- True / None."""
        opt, g, lw = self.opt, self.graph, self.opt.loss_weight
        if not (var.rgb.is_cuda and lw.gan_reg_real is not None
                and (var.get("disc_patches_for") is var.ray_idx or ("gathered" in var and var.get("gathered_for") is var.ray_idx))):
            return None
        real, fake, stack = g.disc_patch_stacks(opt, var)
        sched = self._disc_schedule(real)
        if sched is None or not sched.pairs_eligible(real):
            return None
        if not run:
            return True
        self._disc_flagged = False
        self.disc_step_zero_grads()
        w = lambda k: 10 ** float(lw[k])
        with torch.no_grad():
            ctx = sched.run_paired_a(stack, fake, var.ray_scales, w("gan_disc_real"), w("gan_disc_fake"), own_inputs=True)
        ctx.sched = sched
        return ctx

    def _seg_disc_b(self, var, ctx):
        lw = self.opt.loss_weight
        with torch.no_grad():
            res = ctx.sched.run_paired_b(ctx, 10 ** float(lw.gan_reg_real), step=self._disc_step_tail(ctx.sched))
        out = self._disc_step_scheduled_post(var, res, ctx.real, ctx.fake, not self._dp)
        if self._dp:
            self._seg_disc_pack()
        return out

    def _disc_step_tail(self, sched):
        """The `step=` of disc_step.run_paired_b when the discriminator step may end inside the spectral-norm backward's two launches
        (loss total + gate, RMSprop): one rank (no gradient reduction between gradient and step), the fused RMSprop with one plain group
        over exactly the schedule's weights, the gate words on the device.  Else None: total, gate and step as launches of their own."""
        flags, optim = self._disc_gate_flags(), self.optim_disc
        if (flags is None or knobs.K.no_disc_step_tail or not isinstance(optim, FusedRMSprop) or optim.gate is None
                or self._has_collective() or (self.red_disc is not None and self.red_disc.world_size > 1) or len(optim.param_groups) != 1):
            return None
        group = optim.param_groups[0]
        params = [c.weight_orig for c in sched.convs()]
        others = [p for p in group["params"] if not any(p is q for q in params)]          # (`progress`: never a gradient)
        if (group["momentum"] != 0 or group["centered"] or group["weight_decay"] != 0 or group["maximize"] or not group["capturable"]
                or any(p.grad is not None for p in others) or not all(any(p is q for q in group["params"]) for p in params)
                or optim.gate is not flags["snapshot"] or not all(p.is_contiguous() and p.dtype == torch.float32 for p in params)):
            return None
        for p in params:
            st = optim.state[p]
            if len(st) == 0:
                st["step"] = torch.zeros((), dtype=torch.float32, device=p.device)
                st["square_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            if not st["step"].is_cuda:
                return None
        lw = self.opt.loss_weight
        return dict(w_real=10 ** float(lw.gan_disc_real), w_fake=10 ** float(lw.gan_disc_fake), flags=flags,
                    square_avgs=[optim.state[p]["square_avg"] for p in params], steps=[optim.state[p]["step"] for p in params],
                    lr=group["lr"], alpha=group["alpha"], eps=group["eps"])

    # ------------------------------------------------------------------ the step as LINEAR graphs on three streams
    # A replayed multi-branch hipGraph resolves a dependency between two of its hardware queues late (profiles/r4: the generator's
    # backward sat behind ~60 % of a discriminator chain it does not depend on), picks the hardware queue of a forked branch by the
    # runtime's round-robin over ALL streams the process ever made (870 it/s in a fresh process, 660 behind other work), and costs the
    # host 110 us to launch against 6-14 us for a linear graph.  So the iteration is cut into chains, every one a graph without a fork,
    # on three streams of the trainer's own; the only dependencies between streams are events:
    #     main:  G1 = patch coordinates .. render .. gathers, patch stacks -> ev patches;  F = feature inputs, network, pair loss and its
    #            backward down to the render's rgb (K18);  [wait ev g2a] G2b = loss total + gate, composite / MLP backward from (terms,
    #            d gan / d rgb, d feat / d rgb), Adam  -> ev g2
    #     third: [wait ev patches, ev sn] G2a = D(fake) for the generator, its loss term and gradient wrt the rendered colours -> ev g2a
    #     disc:  D1 = the first spectral normalisation (after its own RMSprop step, stream order) -> ev sn;  D1b = the other two;
    #            [wait ev patches] D2a = forward pairs, BCE -> ev d2a;  D2b = R1, backward pairs, spectral-norm backward + RMSprop -> ev d2
    # Several ranks (`_dp`): G2b and D2b end with the gradients packed into their flat buffer, the all-reduce is a stream-ordered call
    # behind the replay, and the optimiser launch is a graph of its own (G2c, D2c) behind it.
    # Same kernels and the same cotangent values as the one-graph form: the composite's backward sums the cotangents of its rgb
    # aliases either way (autograd_ops._Composite, fan_out).
    def _seg_feat(self, var):
        """The feature loss of the nerf step and its cotangent 10^w d feat / d rgb on the rgb alias the composite made for it."""
        _, h, w, _ = var.ray_idx.shape
        pl = self.graph.perceptual_loss
        fused = pl.loss_and_grad_from_patches(var.rgb_feat, var.gathered, (h, w), 5.0, 10 ** float(self.opt.loss_weight.feat)) \
            if hasattr(pl, "loss_and_grad_from_patches") else None
        if fused is not None:                      # K18: value and weighted cotangent from one call, no autograd graph
            feat, var.g_rgb_feat = fused
        else:
            feat = pl.loss_from_patches(var.rgb_feat, var.gathered, (h, w), 5.0)
            (var.g_rgb_feat,) = torch.autograd.grad(feat, var.rgb_feat, grad_outputs=self._weight(self.opt.loss_weight.feat, feat.device))
        # (compute_loss takes the value from here; `joined`: the replay loop orders the streams, no wait is recorded in a capture)
        var.feat_early, var.feat_early_for, var.feat_early_joined = feat.detach(), var.ray_idx, True

    def _seg_gen_a(self, var):
        lw = self.opt.loss_weight
        sched = None
        if (self.has_disc and self.opt.gan is not None and lw.gan_nerf is not None and var.get("disc_patches_for") is var.ray_idx
                and "rgb_disc" in var):
            sched = self._disc_schedule(var.patch_fake_nerf)
            if sched is not None and not sched.generator_pass_eligible(self.opt, var.patch_fake_nerf):
                sched = None
        if sched is not None:
            # the pass through the frozen discriminator as an explicit schedule: 9 launches instead of autograd's 11, no graph
            with torch.no_grad():
                val, g_disc, d_out = sched.generator_pass(var.patch_fake_nerf.detach(), var.ray_scales, 10 ** float(lw.gan_nerf))
            var.gan_nerf_precomputed, var.d_fake_nerf = val, d_out
            var, loss = self.nerf_forward_loss(var, stage="consume")
            return var, loss, g_disc
        var, loss = self.nerf_forward_loss(var, stage="consume")
        (g_disc,) = torch.autograd.grad(loss.gan_nerf, var.rgb_disc, grad_outputs=self._weight(self.opt.loss_weight.gan_nerf, var.rgb.device))
        return var, loss, g_disc

    def _seg_gen_b(self, var, loss, g_disc):
        dev, lw = var.idx.device, self.opt.loss_weight
        status = ops.mlp_status(dev) if self._uses_f16x3() else None
        keys = [k for k in loss if k != "all" and lw[k] is not None]
        # (the total + gate ride in the first launch of the backward pass, K8's: nothing differentiates through them)
        terms, ws = self._weighted_total(loss, flags=dict(bad=self._bad, word_finite=1, snapshot=self._gate_nerf, status=status,
                                                          word_status=0, step_counter=getattr(self, "_rng_counter", None)),
                                         defer=not knobs.K.no_total_in_bwd)
        roots = {"gan_nerf": (var.rgb_disc, g_disc)}
        if "g_rgb_feat" in var:                        # (the feature chain ran as a graph of its own, `_seg_feat`)
            roots["feat"] = (var.rgb_feat, var.g_rgb_feat)
        pairs = [roots.get(k, (t, w)) for k, t, w in zip(keys, terms, ws)]
        torch.autograd.backward([r for r, _ in pairs], [c for _, c in pairs])
        ops.flush_pending_total()
        if self._dp:
            # several ranks: this graph ENDS with the gradients (scaled by 1 / world) and the gate words in the flat buffer; the
            # all-reduce is a stream-ordered call behind the replay, the Adam launch a graph of its own (`_seg_gen_c`)
            self.red_nerf.pack(flags=self._bad)
            return var, loss
        self._guard_nerf(var, loss)
        self.nerf_apply()
        return var, loss

    def _seg_gen_c(self):
        """Behind the nerf step's all-reduce: Adam on the averaged gradients where they lie in the flat buffer, gated by its tail."""
        self.red_nerf.adopt()
        self.optim_nerf.step()
        self.graph.nerf.mark_heads_dirty()

    def _seg_disc_pack(self):
        """End of the discriminator step's gradient graph with several ranks (see `_seg_gen_b`)."""
        self.red_disc.pack(flags=self._bad)

    def _seg_disc_c(self):
        self.red_disc.adopt()
        self.optim_disc.step()

    def _collective(self, name, red):
        """One flat all-reduce on the current stream, between two graph replays; HIP events around it when `collective_events` is a
        list (bench.py / tools/train_dp.py: the xGMI figure)."""
        ev = getattr(self, "collective_events", None)
        if ev is None:
            red.all_reduce()
            return
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        red.all_reduce()
        e1.record()
        ev.append((name, e0, e1))

    def _linear_eager(self, var):
        """The segments on their streams, eagerly (warm-up), with the same dependencies as the replays."""
        main = torch.cuda.current_stream(var.idx.device)
        side = self._side
        side.wait_stream(main)
        with torch.cuda.stream(side):
            self._seg_sn()
        var = self._seg_render(var)
        side.wait_stream(main)
        if not ("rgb_disc" in var and var.get("gathered_for") is var.ray_idx and (self.opt.loss_weight.feat is None or "rgb_feat" in var)):
            raise _FormUnavailable("no fan-out aliases / fused gathers in this configuration")
        if self.opt.loss_weight.feat is not None:
            third = self._third
            third.wait_stream(main)
            with torch.cuda.stream(third):            # (the stream F is CAPTURED on, see `_capture_linear`)
                self._seg_feat(var)
        main.wait_stream(side)
        var, loss, g_disc = self._seg_gen_a(var)           # (on `main`, the stream G2a is CAPTURED on: per-stream kernel state is keyed by it)
        if self.opt.loss_weight.feat is not None:
            main.wait_stream(self._third)
        var, loss = self._seg_gen_b(var, loss, g_disc)
        if self._dp:
            self._collective("nerf", self.red_nerf)                # (same order on every rank: nerf, then discriminator)
            self._seg_gen_c()
        with torch.cuda.stream(side):
            var, dloss = self._seg_disc(var)
            if self._dp:
                self._collective("disc", self.red_disc)
                self._seg_disc_c()
        main.wait_stream(side)
        loss.update({k: v for k, v in dloss.items() if k != "all"})
        return {k: v.detach() for k, v in loss.items() if torch.is_tensor(v)}

    def _extra(self, name):
        """TP_EXTRA_LAUNCHES="G2a=10,F=10" (experiment): that many one-thread launches at the end of the named graph -- the slope of the
        iteration time over the count says whether that graph's end is on the critical path (~3 us per launch) or not (0)."""
        for item in knobs.K.extra_launches.split(","):
            if item.strip() and item.split("=")[0].strip() == name:
                for _ in range(int(item.split("=")[1])):
                    ops.stamp(self._extra_slots, 0)

    def _stamp(self, name):
        """TP_STAMPS=1 (tools/linear_timeline.py): a one-thread launch writing the device clock, captured at the segment boundaries."""
        if not knobs.K.stamps:
            return
        if getattr(self, "_stamps", None) is None:
            self._stamps, self._stamp_names = torch.zeros(32, dtype=torch.int64, device=self._bad.device), []
        if name not in self._stamp_names:
            self._stamp_names.append(name)
        ops.stamp(self._stamps, self._stamp_names.index(name))

    # share of the device the weight gradient of the render's backward fills in the linear graphs: the discriminator step's second half
    # runs beside it, and with every CU taken (one 158-KB workgroup each) its launches sat out the whole kernel -- 7/8 measured best
    # (256 / 248 / 240 / 232 / 224 / 208 CUs: 956 / 958 / 950 / 956 / 977 / 969 it/s on one box, profiles/r6)
    WGRAD_CU_SHARE = 7 / 8

    def _capture_linear(self, cap):
        side = self._side
        has_feat = self.opt.loss_weight.feat is not None
        if knobs.K.stamps and getattr(self, "_stamps", None) is None:
            # (made OUTSIDE the captures: a zero-fill captured into the first graph would wipe the other streams' stamps on every replay)
            self._stamps, self._stamp_names = torch.zeros(32, dtype=torch.int64, device=self._bad.device), []
            torch.cuda.synchronize(self._bad.device)
        self._graphs = g = {k: torch.cuda.CUDAGraph() for k in ["D1", "G1"] + (["F"] if has_feat else []) + ["G2a", "G2b"]}
        self._events = {k: torch.cuda.Event() for k in ("sn", "patches", "g2a", "g2", "d2")}
        self._extra_slots = torch.zeros(1, dtype=torch.int64, device=self._bad.device)
        # Memory pools: graphs that share a pool must replay in capture order and never concurrently (a block freed while one is
        # captured is handed to the next), which holds for D1 -> D2a -> D2b and for G1 -> G2a -> G2b but not across the streams (F: its own).
        # Tensors that cross the streams (the spectral-norm sets, the render's patch stacks / scales) are kept referenced for the life of
        # the graphs, so that no capture re-uses their memory.
        counts = self.launch_counts = {}                 # nodes per captured graph (kernel launches; TP_STAMPS adds two to each)
        # The generator's pass through the frozen discriminator (G2a) needs the FIRST normalised weight set only, and D1 does not start
        # before the render's MLP kernel ends (it sits that kernel out: every CU is taken), so G2a is what the render's backward waits
        # for: that set as a graph of its own (3 launches), the other two behind it (D1b: 2 x 2 + 1), 8 launches instead of 7.
        # Measured +0.8 % (939-940 -> 945-948 it/s on one box, profiles/r6) now that the iteration boundary is short; in round 5, with
        # a 70-us boundary in which D1 ran unhindered, the extra launch cost more than the split gave (914-920 against 925-926).
        if not knobs.K.no_sn_split:
            g["D1b"] = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g["D1"], stream=side):
                self._stamp("D1.0"); self._seg_sn(0); self._stamp("D1.1"); self._extra("D1"); counts["D1"] = ops.capture_node_count()
            with torch.cuda.graph(g["D1b"], stream=side, pool=g["D1"].pool()):
                self._seg_sn(1); self._extra("D1b"); counts["D1b"] = ops.capture_node_count()
        else:
            with torch.cuda.graph(g["D1"], stream=side):
                self._stamp("D1.0"); self._seg_sn(); self._stamp("D1.1"); self._extra("D1"); counts["D1"] = ops.capture_node_count()
        keep = [list(self.graph.discriminator._sn_queue)]
        with torch.cuda.graph(g["G1"], stream=cap):
            self._stamp("G1.0"); var = self._seg_render(AttrDict(dict(self._static_in))); self._stamp("G1.1"); self._extra("G1"); counts["G1"] = ops.capture_node_count()
        keep.append(dict(var))
        if has_feat:
            # (captured on the THIRD stream although it replays on `main`: per-stream kernel state -- tile counters, workspaces -- is keyed
            # by the capturing stream, and F runs beside G2a, which is captured on `cap`; a pool of its own for the same reason)
            with torch.cuda.graph(g["F"], stream=self._third):
                self._stamp("F.0"); self._seg_feat(var); self._stamp("F.1"); self._extra("F"); counts["F"] = ops.capture_node_count()
            keep.append(dict(var))
        with torch.cuda.graph(g["G2a"], stream=cap, pool=g["G1"].pool()):
            self._stamp("G2a.0"); var, loss, g_disc = self._seg_gen_a(var); self._stamp("G2a.1"); self._extra("G2a"); counts["G2a"] = ops.capture_node_count()
        keep.append((dict(var), dict(loss), g_disc))
        with torch.cuda.graph(g["G2b"], stream=cap, pool=g["G1"].pool()):
            self._stamp("G2b.0"); var, loss = self._seg_gen_b(var, loss, g_disc); self._stamp("G2b.1"); self._extra("G2b"); counts["G2b"] = ops.capture_node_count()
        keep.append(dict(var))
        split_dp = self._dp
        if split_dp:
            # several ranks: [G2b: ... gradients, pack] | flat all-reduce (stream-ordered call) | [G2c: Adam from the flat buffer]
            g["G2c"] = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g["G2c"], stream=cap, pool=g["G1"].pool()):
                self._seg_gen_c(); self._extra("G2c"); counts["G2c"] = ops.capture_node_count()
        # The discriminator step in TWO graphs when its paired schedule applies: D2a = private copies of the patch stacks, forward pairs,
        # BCE terms -- the last reads of anything the render wrote -- and D2b = the R1 passes, backward pairs, spectral-norm backward,
        # RMSprop.  With `pipeline_disc_tail` the next iteration's render starts behind D2a instead of behind D2b (`_replay_linear`).
        ctx = self._seg_disc_a(var) if not knobs.K.no_disc_split else None
        if ctx is not None:
            g["D2a"], g["D2b"] = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            self._events["d2a"] = torch.cuda.Event()
            self.disc_step_zero_grads()
            with torch.cuda.graph(g["D2a"], stream=side, pool=g["D1"].pool()):
                self._stamp("D2.0"); ctx = self._seg_disc_a(var, run=True); self._stamp("D2a.1"); self._extra("D2a"); counts["D2a"] = ops.capture_node_count()
            keep.append(ctx)
            with torch.cuda.graph(g["D2b"], stream=side, pool=g["D1"].pool()):
                var, dloss = self._seg_disc_b(var, ctx); self._stamp("D2.1"); counts["D2b"] = ops.capture_node_count()
                self._extra("D2b")
        else:
            g["D2"] = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g["D2"], stream=side, pool=g["D1"].pool()):
                self._stamp("D2.0"); var, dloss = self._seg_disc(var); self._stamp("D2.1"); counts["D2"] = ops.capture_node_count()
        keep.append(dict(var))
        if split_dp:
            g["D2c"] = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g["D2c"], stream=side, pool=g["D1"].pool()):
                self._seg_disc_c(); self._extra("D2c"); counts["D2c"] = ops.capture_node_count()
        self._graphs_keep = keep
        loss.update({k: v for k, v in dloss.items() if k != "all"})
        self._static_loss = {k: v.detach() for k, v in loss.items() if torch.is_tensor(v)}
        self._first_replay = True

    def _replay_linear(self):
        g, ev = self._graphs, self._events
        main, side, third = self._capture_stream, self._side, self._third
        cur = torch.cuda.current_stream(self._bad.device)
        # the caller's stream hands over the step's inputs (and, the first time, parameters restored / loaded since the capture); it
        # waited for the previous iteration's last launches below, so does everything here
        inputs = self.__dict__.pop("_inputs_event", None)
        if inputs is not None:
            main.wait_event(inputs)                   # (`defer_results`: the inputs went in on the third stream, behind the caller's mark)
        else:
            main.wait_stream(cur)
        if getattr(self, "_g2_pending", False) and not self._pipelined():
            main.wait_event(ev["d2"])                 # (nothing else orders the render behind the previous discriminator step)
        if getattr(self, "_d2_pending", False) and self._pipelined():
            # pipelined: this render may start while the previous discriminator step's second half still runs; it overwrites the patch
            # stacks / scales, whose last readers are in that step's FIRST half
            main.wait_event(ev["d2a"])
        with torch.cuda.stream(side):
            # ... and so does the discriminator stream, on EVERY replay (D1 reads `weight_orig` and rewrites `weight_u` / `weight_v` in
            # place: a load_state_dict, parameter broadcast or checkpoint copy the caller enqueued between two iterations comes first).
            # `train_iteration` marks the caller's stream BEFORE its own tp_step_inputs launch (which writes nothing D1 reads): D1 then
            # starts behind the caller's work without waiting for that launch; a bare `replay()` waits for the whole stream.
            mark = self.__dict__.pop("_caller_mark", None)
            if mark is not None:
                side.wait_event(mark)
            else:
                side.wait_stream(cur)
            if not self._first_replay:
                side.wait_event(ev["g2"])            # set 1 is read by the generator's passes through the frozen discriminator
            poll = self.__dict__.pop("_poll_on_side", None)
            if poll is not None:
                # (several ranks + `defer_results`: the gate words out to the host HERE -- behind both all-reduces of the previous
                # iteration, in front of this iteration's packs, on every rank alike; D1 sits the render's forward out anyway)
                ops.step_inputs([], [], words=self._poll_words(), words_host=poll)
                polled = torch.cuda.Event()
                polled.record(side)
                self._bad_poll = (poll, polled)
            g["D1"].replay()
            ev["sn"].record(side)
            if "D1b" in g:
                g["D1b"].replay()
        self._first_replay = False
        # (The order in which the host submits F / G2a + G2b / D2 makes no difference -- six orders measured within 0.5 % on one box -- and
        # the host is 4x ahead of the device: 260 us of launches per 1.17 ms iteration.)
        # Which of the two chains between the render and its backward shares the render's stream: each stream hop (event -> first launch
        # of a graph on another hardware queue) costs ~20 us, and the feature chain (137 us alone) is the longer of the two, the
        # generator's pass through the discriminator (101 us) also waits for D1 -- so the feature chain stays on `main` and that pass
        # takes the third stream (the other layout: profiles/r5/21).
        with torch.cuda.stream(main):
            g["G1"].replay()
            ev["patches"].record(main)
            if "F" in g:
                g["F"].replay()
        with torch.cuda.stream(third):
            third.wait_event(ev["patches"])           # (recorded on `main` behind its wait for the caller's stream)
            third.wait_event(ev["sn"])
            g["G2a"].replay()
            ev["g2a"].record(third)
        with torch.cuda.stream(main):
            main.wait_event(ev["g2a"])
            g["G2b"].replay()
            if "G2c" in g:                           # several ranks: gradients | all-reduce | Adam.  The host issues the nerf step's
                self._collective("nerf", self.red_nerf)          # collective before the discriminator step's on EVERY rank.
                g["G2c"].replay()
            ev["g2"].record(main)
        with torch.cuda.stream(side):
            side.wait_event(ev["patches"])
            if "D2a" in g:
                g["D2a"].replay()
                ev["d2a"].record(side)
                g["D2b"].replay()
            else:
                g["D2"].replay()
            if "D2c" in g:                           # several ranks: gradients | all-reduce | RMSprop
                self._collective("disc", self.red_disc)
                g["D2c"].replay()
            ev["d2"].record(side)
        # what the caller enqueues next -- reads of the losses, the next iteration's inputs -- comes after both optimiser steps ...
        if self._defers_results():
            # `defer_results`: the calling stream is ordered behind the consumption of this iteration's INPUTS only (train_iteration);
            # `wait_all()` / `finish()` / `flush_flags()` order it behind the results.  The next iteration then starts on `main`
            # straight behind this one's Adam launch, without the two stream hops main -> caller -> main.
            self._g2_pending = True
            self._d2_pending = True
            return
        cur.wait_event(ev["g2"])
        if self._pipelined():
            # ... unless the caller asked for the discriminator step's second half to run beside the next iteration's render
            # (`pipeline_disc_tail`): the calling stream then waits for the generator's step only; `finish()` waits for the rest.
            self._d2_pending = True
        else:
            cur.wait_event(ev["d2"])

    @property
    def _third(self):
        """Third stream of the linear graphs (the generator's pass through the frozen discriminator); the generic form forks the feature
        chain onto it (graph.Graph._feature_loss_early)."""
        return self.graph.feat_stream

    def _pipelined(self):
        return bool(self.pipeline_disc_tail) and self._linear and self._graphs is not None and "D2a" in self._graphs

    def _defers_results(self):
        return bool(self.defer_results) and self._linear and self._graphs is not None

    def wait_all(self):
        """Make the calling stream wait for everything the last `train_iteration` enqueued (with `pipeline_disc_tail`: the
        discriminator step's second half and its RMSprop step)."""
        cur = torch.cuda.current_stream(self._bad.device)
        if getattr(self, "_g2_pending", False):
            cur.wait_event(self._events["g2"])
            self._g2_pending = False
        if getattr(self, "_d2_pending", False):
            cur.wait_event(self._events["d2"])
            self._d2_pending = False

    def _prefetch_spectral_weights(self, var):
        """The spectral normalisations of this iteration's three discriminator passes (nerf step's D(fake), D(real), D(fake):
        the reference's order of power iterations) depend on the weights only: issue them NOW on the discriminator branch's
        stream, next to the render's MLP kernel, instead of 5 launches in front of each pass (K7; gan_modules.Discriminator.
        prefetch_spectral_weights).  Only when every consumer of the iteration takes prefetched weights: the frozen
        discriminator of the nerf step always does, the discriminator step when it runs as the explicit schedule (K16)."""
        if not self.has_disc or self._side is None or knobs.K.no_sn_prefetch:
            return
        opt, disc = self.opt, self.graph.discriminator
        if not (hasattr(disc, "prefetch_spectral_weights") and disc.training and opt.gan is not None):
            return
        p, B = int(opt.patch_size), len(var.idx)
        probe = torch.empty(0, device=var.idx.device).new_empty((B, 0, p, p))          # (shape / device carrier: no data)
        n = 1 + (2 if self._disc_schedule(probe) is not None else 0)
        main = torch.cuda.current_stream(probe.device)
        self._side.wait_stream(main)                 # (after the previous iteration's RMSprop step, whichever stream ran it)
        with torch.cuda.stream(self._side):
            disc.prefetch_spectral_weights(n)

    def disc_step(self, var, apply=True):
        self._disc_flagged = False
        var, loss = super().disc_step(var, apply=apply)
        if not apply and not self._disc_flagged:
            self._flag_disc(self._disc_total)                     # (before the join: the reductions carry the word)
        return var, loss

    def _body_b(self):
        """The optimiser steps of a step whose reductions ran after the join (gradients are averaged, the gate words global:
        snapshot them again for the two optimiser launches)."""
        self._gate_nerf.copy_(self._bad)
        self._guard_nerf(None, None)
        self.optim_nerf.step()
        self.graph.nerf.mark_heads_dirty()
        if self.has_disc:
            self._gate_disc.copy_(self._bad)
            self._guard_disc()
            self.optim_disc.step()

    # ------------------------------------------------------------------ capture
    def _snapshot(self):
        optims = [self.optim_nerf] + ([self.optim_disc] if self.has_disc else [])
        return dict(graph={k: v.detach().clone() for k, v in self.graph.state_dict().items()},
                    optim=[copy.deepcopy(o.state_dict()["state"]) for o in optims], it=self.it,
                    sampler_it=self.graph.patch_sampler.iterations)

    def _restore(self, snap):
        """Undo the warm-up iterations IN PLACE (the captured graph keeps the addresses): parameters, buffers (spectral
        norm u / v, progress), optimiser moments and step counters, iteration counters."""
        with torch.no_grad():
            for k, v in self.graph.state_dict().items():
                v.copy_(snap["graph"][k])
        optims = [self.optim_nerf] + ([self.optim_disc] if self.has_disc else [])
        for o, saved in zip(optims, snap["optim"]):
            index = {id(p): i for i, p in enumerate(p for g in o.param_groups for p in g["params"])}
            for p, st in o.state.items():
                old = saved.get(index[id(p)])
                for name, t in st.items():
                    if torch.is_tensor(t):
                        if old is None:
                            t.zero_()
                        else:
                            t.copy_(old[name])
        self.it = snap["it"]
        self.graph.patch_sampler.iterations = snap["sampler_it"]
        self.graph.nerf.mark_heads_dirty()

    def load_optim_state(self, optim_nerf=None, optim_disc=None):
        """As GanTrainer.load_optim_state; ``Optimizer.load_state_dict`` REPLACES the moment / step tensors, whose addresses a
        captured step holds: the next train_iteration captures again (from the loaded state)."""
        super().load_optim_state(optim_nerf=optim_nerf, optim_disc=optim_disc)
        self._graph = self._graph_b = None

    def capture(self, var: AttrDict, warmup: int = 3):
        """Warm up eagerly on a side stream (lazy inits: MIOpen solver search, weight packing, optimiser state), then
        record the step.  ``var`` fixes the shapes.  The warm-up iterations are real optimiser steps on ``var``; they
        are rolled back afterwards, so that a captured run starts from the same state as an eager one."""
        dev = var.image.device
        for name, optim in (("lr_nerf", self.optim_nerf), ("lr_disc", getattr(self, "optim_disc", None))):
            if optim is not None:
                self._adopt_group_lr(name, optim)             # (a load_state_dict before the capture replaced the tensors)
        self.graph.patch_sampler.device_lo = torch.zeros((), device=dev)
        if not knobs.K.torch_rng:
            # the step's random draws (patch scale / shifts, stratified jitter) come from Philox streams keyed by the seed and a
            # step counter on the device, read inside tp_patch_coords / tp_raygen and advanced by the loss-total launch: no
            # torch.rand launches in the step and no generator-state fills before every replay
            if getattr(self, "_rng_counter", None) is None:
                self._rng_counter = torch.zeros(1, dtype=torch.int64, device=dev)       # (attached to the graph only inside `_body_a`)
        self._static_in = AttrDict({k: v.clone() for k, v in var.items() if torch.is_tensor(v)})
        self._select_form(var)
        # the step's streams are made here, one after the other (main / capture, discriminator, feature chain): consecutive hardware queues
        if self._linear and (getattr(self, "_capture_stream", None) is None or self._side is None
                             or getattr(self.graph, "feat_stream", None) is None) and not knobs.K.no_queue_probe:
            self._capture_stream, self._side, self.graph.feat_stream = streams_after_collectives(dev, 3, self.red_nerf.group)
            self.queue_probe = dict(LAST_QUEUE_PROBE)                    # (bench lines report it: `train.queues`)
        if getattr(self, "_capture_stream", None) is None:
            self._capture_stream = torch.cuda.Stream(device=dev)
        if self._linear and self._side is None:
            self._side = torch.cuda.Stream(device=dev)
        if self._linear and getattr(self.graph, "feat_stream", None) is None:
            self.graph.feat_stream = torch.cuda.Stream(device=dev)
        snap = self._snapshot()
        # warm up on the stream the capture will use: per-stream state (the tile counters of the convolution kernels,
        # ops._conv_scratch) must exist before the capture and is keyed by the stream
        if getattr(self, "_capture_stream", None) is None:
            self._capture_stream = torch.cuda.Stream(device=dev)
        side = self._capture_stream
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            done = 0
            while done < warmup:
                self.graph.patch_sampler.update_device_bound()
                try:
                    self._body(AttrDict(dict(self._static_in)))
                except _FormUnavailable:
                    # (every rank runs the same configuration and takes this branch in the same iteration, before any collective of it)
                    self._restore(snap)
                    self._linear = self._dp = False
                    self._point_gates()
                    self._set_wgrad_share()
                    done = 0
                    continue
                self._after_step()
                done += 1
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self._d2_pending = self._g2_pending = False
        self._graph = torch.cuda.CUDAGraph()
        self.optim_nerf.zero_grad(set_to_none=True)
        if self.has_disc:
            self.optim_disc.zero_grad(set_to_none=True)
        self.graph.patch_sampler.update_device_bound()          # outside the capture
        self._graph_b = None
        if self._linear:
            self._capture_linear(side)
        elif self._split_around_collectives():
            self.launch_counts = {}
            with torch.cuda.graph(self._graph, stream=side):
                self._static_loss = self._body_a(AttrDict(dict(self._static_in)))
                self.launch_counts["A"] = ops.capture_node_count()
            self._graph_b = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._graph_b, stream=side, pool=self._graph.pool()):
                self._body_b()
                self.launch_counts["B"] = ops.capture_node_count()
        else:
            self.launch_counts = {}
            with torch.cuda.graph(self._graph, stream=side):
                self._static_loss = self._body(AttrDict(dict(self._static_in)))
                self.launch_counts["step"] = ops.capture_node_count()
        self._restore(snap)
        flagged = self._read_bad(blocking=True)
        if flagged[0] and self._uses_f16x3():
            return self._fall_back_to_fp32(var, warmup)
        if flagged[1] or flagged[2]:
            raise FloatingPointError("non-finite loss during the warm-up iterations of the captured step")
        return self

    def _after_step(self, fill_progress=True):
        """``fill_progress=False``: the replay loop writes the value with the NEXT iteration's tp_step_inputs launch (nothing reads
        `progress` in between: the reference's discriminator never uses it in its forward)."""
        if self.has_disc and fill_progress:
            self.graph.discriminator.progress.data.fill_(self.it / self.max_iter)
        self.it += 1
        self.graph.patch_sampler.iterations = self.it
        self.graph.nerf.mark_heads_dirty()

    # ------------------------------------------------------------------ host side of the step gate
    def _read_bad(self, blocking=False):
        """[range flag, non-finite] seen by the gate; blocking, or from the asynchronous copy queued by the previous call.
        With several ranks the words are global (`_reduce_all`) and every rank must ACT on them at the same iteration (a
        re-capture issues collectives): the copy queued by the previous call is then waited for instead of polled -- by then
        it is one whole iteration old, so the wait only bounds how far the host runs ahead of the GPU."""
        if blocking:
            self._bad_poll = None
            if self._dp:
                # the job-wide words: both buffers' tails (a word a rank raises is packed into a tail in the same iteration)
                n = len(self._bad)
                return ((self.red_nerf.gate_words[:n] != 0) | (self.red_disc.gate_words[:n] != 0)).to(torch.int32).tolist()
            return self._bad.tolist()
        seen = [0, 0, 0]
        prev = self._bad_poll
        if prev is not None and self.red_nerf.world_size > 1:
            prev[1].synchronize()
        if prev is not None and prev[1].query():
            seen = [int(w != 0) for w in prev[0].tolist()[:3]]
            self._bad_poll = None
        return seen

    def _poll_words(self):
        """The device words `train_iteration` copies out to the host: the sticky words of this rank, or (`_dp`) the job-wide tail of
        the nerf step's buffer, which carries them one pack later."""
        return self.red_nerf.gate_words if self._dp else self._bad

    def flush_flags(self):
        """Blocking read of the gate words as the LAST replay left them; acts on them like `train_iteration` does.  The words of a
        replay are copied out by the NEXT iteration's tp_step_inputs launch, so a caller that stops issuing iterations -- end of training,
        before a checkpoint -- calls this to learn about a withheld final step (re-capture with fp32, or FloatingPointError)."""
        if self._graph is None:
            return [0, 0, 0]
        self.wait_all()
        flagged = self._read_bad(blocking=True)
        if flagged[0] and self._uses_f16x3():
            self._fall_back_to_fp32(AttrDict(dict(self._static_in)))
        elif flagged[1] or flagged[2]:
            raise FloatingPointError("non-finite loss in a captured training step (the update was withheld on the device)")
        return flagged

    finish = flush_flags

    def _bad_poll_slot(self):
        """Pinned host words for this iteration's copy of the gate words (written by the tp_step_inputs launch in front of the
        replay, i.e. the words as the PREVIOUS replay left them), or None while an earlier copy has not been read yet."""
        if self._bad_poll is not None:
            return None
        n = self._poll_words().numel()
        if getattr(self, "_bad_host", None) is None or self._bad_host.numel() != n:
            self._bad_host = torch.zeros(n, dtype=torch.int32).pin_memory()
        return self._bad_host

    def _fall_back_to_fp32(self, var, warmup=3):
        warnings.warn("texpose_amd: an activation left the fp16 range of the f16x3 recording forward; the flagged steps "
                      "were withheld on the device and the step is re-captured with arch.mlp_train_precision='fp32'")
        torch.cuda.synchronize()
        self.skipped_steps += 1
        self.graph.nerf.train_precision = "fp32"
        self._bad.zero_()
        self.red_nerf.clear_gate()
        if self.red_disc is not None:
            self.red_disc.clear_gate()
        ops.mlp_status(self._bad.device).zero_()
        self._bad_poll = None
        return self.capture(var, warmup=warmup)

    def replay(self):
        """Issue the captured step once: the linear graphs on their three streams (`_replay_linear`), or the generic form on the current
        stream -- one graph, or (several ranks) graph A, the gradient all-reduces, graph B."""
        if self._linear:
            return self._replay_linear()
        self._graph.replay()
        if self._graph_b is not None:                            # collectives between the two replays, stream-ordered
            ev = getattr(self, "collective_events", None)        # (a list: HIP-event pairs around the step's reductions, bench.py)
            if ev is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            self._reduce_all()
            if ev is not None:
                e1.record()
                ev.append(("both", e0, e1))
            self._graph_b.replay()

    def train_iteration(self, var: AttrDict):
        if self._graph is None:
            self.capture(var)
        # the batch into the static input tensors, the annealed patch-scale bound, the discriminator's progress value (of the
        # iteration before, as in the reference :182) and the gate words out to pinned memory: ONE launch (K13 tp_step_inputs;
        # torch: a multi-tensor copy per dtype, two fills, a copyBuffer)
        pairs = [(dst, var[k]) for k, dst in self._static_in.items() if var[k] is not dst]
        fused = [(d, s) for d, s in pairs if torch.is_tensor(s) and s.device == d.device and s.dtype == d.dtype and s.shape == d.shape
                 and s.is_contiguous() and d.is_contiguous() and d.data_ptr() % 16 == 0 and s.data_ptr() % 16 == 0]
        if self._defers_results() and len(fused) != len(pairs):
            self.wait_all()                                   # (a batch member the one-launch copy does not take: strict order this once)
        for d, s in pairs:
            if not any(d is d2 for d2, _ in fused):
                d.copy_(s, non_blocking=True)                 # (on the CALLER's stream, in front of the mark below)
        if self._linear:
            # everything the caller enqueued on its stream up to here -- the copies above included: with `defer_results` the render
            # waits for this mark and nothing else of the caller's -- is ordered in front of the step's first graphs.  (It sits in front of
            # the tp_step_inputs launch, which writes nothing the discriminator stream's first graph reads.)
            self._caller_mark = torch.cuda.Event()
            self._caller_mark.record()
        for name, optim in (("lr_nerf", self.optim_nerf), ("lr_disc", getattr(self, "optim_disc", None))):
            if optim is not None and any(g["lr"] is not getattr(self, name + "_used") for g in optim.param_groups):
                self._adopt_group_lr(name, optim)             # an Optimizer.load_state_dict since the last replay
        sampler = self.graph.patch_sampler
        scalars = [(sampler.device_lo, sampler._host_range()[0])]
        if self.has_disc:
            # reference :182 leaves it / max_iter in `progress` after iteration `it`; nothing reads it during the iteration
            scalars.append((self.graph.discriminator.progress.data, self.it / self.max_iter))
        poll = self._bad_poll_slot()
        if self._defers_results():
            # `defer_results`: the step's inputs go in on the THIRD stream, behind the caller's mark (its batch is ready) and behind the
            # previous iteration's render -- the last reader of the static inputs (the latent rows' backward reads a private copy of
            # `idx`, autograd_ops._LatentRows) -- i.e. while that iteration's backward still runs.  The generator stream then goes from
            # its Adam graph straight into the next render graph: an eager launch between two graphs of one stream cost 39 + 13 us of
            # idle stream there (graph end -> kernel -> graph start; profiles/r6), an event wait costs nothing.
            # Several ranks (`_dp`): the gate words the host polls must be read at the SAME point of the step on every rank -- behind
            # the previous iteration's all-reduces, in front of this one's pack -- or two ranks could act on a word one iteration apart
            # and meet in different collectives.  This launch has no such place (it floats against the previous backward), so the words
            # are copied out by a launch of their own on the discriminator stream (`_replay_linear`: behind ev g2 and D2c, in front of D1).
            words_here = poll is not None and not self._dp
            third = self._third
            third.wait_event(self._caller_mark)
            if not self._first_replay:
                third.wait_event(self._events["patches"])
            with torch.cuda.stream(third):
                ops.step_inputs(fused, scalars, words=self._poll_words() if words_here else None, words_host=poll if words_here else None)
                ev = torch.cuda.Event()
                ev.record()
            torch.cuda.current_stream(self._bad.device).wait_event(ev)      # (the caller may overwrite its batch tensors behind this)
            self._inputs_event = ev
            if poll is not None and not words_here:
                self._poll_on_side, poll = poll, None
        else:
            ops.step_inputs(fused, scalars, words=self._poll_words() if poll is not None else None, words_host=poll)
            if poll is not None:
                ev = torch.cuda.Event()
                ev.record()
        if poll is not None:
            self._bad_poll = (poll, ev)
        self.replay()
        self._after_step(fill_progress=False)
        flagged = self._read_bad()                              # outside the graph: event query of the pinned copy
        if flagged[0] and self._uses_f16x3():
            self._fall_back_to_fp32(AttrDict(dict(self._static_in)))
        elif flagged[1] or flagged[2]:
            raise FloatingPointError("non-finite loss in a captured training step (the update was withheld on the device)")
        return self._static_in, AttrDict(self._static_loss)
