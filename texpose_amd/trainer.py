"""One GAN training iteration of the adapt_st_gan stage on the HIP path, following the reference's
``Model.train_iteration`` (model/nerf_adapt_st_gan.py:108-202): patch coordinates -> nerf step (render,
gathers, discriminator on the fake patch, photometric / uncertainty / transient / feature / GAN losses,
Adam) -> discriminator step (real + R1 penalty + fake, RMSprop).  Used by bench.py and the tests; the
reference's own engine runs unchanged against texpose_amd.graph.Graph.

Data parallel: pass ``world_size > 1`` after texpose_amd.dist.init_distributed(); gradients of each
optimiser are averaged with ONE flat all-reduce after all backward calls of that step.
"""
from __future__ import annotations

import torch

from . import dist as tdist
from . import ops
from .graph import Graph, summarize_loss
from .options import AttrDict


class GanTrainer:
    capturable = False                       # optimiser state on the device (required inside a hipGraph)

    def __init__(self, opt, graph: Graph, n_train: int, max_iter: int = 6000 * 189 // 8):
        self.opt, self.graph, self.max_iter, self.it = opt, graph, max_iter, 0
        if not hasattr(graph, "latent_vars_trans"):
            graph.attach_latents(n_train, opt)
        nerf_params = [p for p in graph.nerf.parameters() if p.requires_grad]
        self.nerf_group = nerf_params + list(graph.latent_vars_light.parameters()) + \
            list(graph.latent_vars_trans.parameters())
        # a captured step reads its learning rates from device memory, so that a scheduler (the reference decays the
        # nerf rate per epoch, ExponentialLR, model/nerf_adapt_st_gan.py:73-84) can change them between replays
        dev = nerf_params[0].device
        self.lr_nerf = torch.tensor(float(opt.optim.lr), device=dev) if self.capturable else float(opt.optim.lr)
        self.optim_nerf = torch.optim.Adam([dict(params=nerf_params, lr=self.lr_nerf),
                                            dict(params=graph.latent_vars_light.parameters(), lr=self.lr_nerf),
                                            dict(params=graph.latent_vars_trans.parameters(), lr=self.lr_nerf)],
                                           capturable=self.capturable)
        self.has_disc = hasattr(graph, "discriminator") and opt.gan is not None
        if self.has_disc:
            self.disc_group = [p for p in graph.discriminator.parameters()]
            self.lr_disc = torch.tensor(float(opt.optim_disc.lr), device=dev) if self.capturable else float(opt.optim_disc.lr)
            self.optim_disc = torch.optim.RMSprop([dict(params=self.disc_group, lr=self.lr_disc)],
                                                  capturable=self.capturable)
        self.red_nerf = tdist.FlatGradAllReducer(self.nerf_group)
        self.red_disc = tdist.FlatGradAllReducer(self.disc_group) if self.has_disc else None

    @staticmethod
    def _toggle(module, flag):
        for p in module.parameters():
            p.requires_grad_(flag)

    def nerf_step(self, var):
        opt, g = self.opt, self.graph
        if self.has_disc:
            self._toggle(g.discriminator, False)
        self.optim_nerf.zero_grad(set_to_none=True)
        var = g.nerf_forward(opt, var, mode="train")
        loss = summarize_loss(opt, g.compute_loss(opt, var, mode="train", train_step="nerf"))
        loss.all.backward()
        self.red_nerf.reduce()
        self.optim_nerf.step()
        return var, loss

    def disc_step(self, var):
        opt, g = self.opt, self.graph
        self._toggle(g.discriminator, True)
        self.optim_disc.zero_grad(set_to_none=True)
        var = g.disc_forward(opt, var, mode="train")
        loss = g.compute_loss(opt, var, mode="train", train_step="disc")
        w = lambda k: 10 ** float(opt.loss_weight[k])
        (w("gan_disc_real") * loss.gan_disc_real).backward(retain_graph=True)
        if opt.loss_weight.gan_reg_real is not None:          # R1: double backward through the discriminator
            reg = g.compute_grad2(opt, var.d_real_disc, var.patch_real).mean()
            (w("gan_reg_real") * reg).backward()
            # the reference logs the WEIGHTED penalty: it scales the tensor it has just stored, in place (:151-153)
            loss.gan_reg_real = (w("gan_reg_real") * reg).detach()
        (w("gan_disc_fake") * loss.gan_disc_fake).backward()
        self.red_disc.reduce()
        self.optim_disc.step()
        return var, loss

    def set_lr(self, nerf: float = None, disc: float = None):
        """Learning rates for the following iterations (eager: param groups; captured: the device scalars the graph reads)."""
        for value, name, optim in ((nerf, "lr_nerf", self.optim_nerf), (disc, "lr_disc", getattr(self, "optim_disc", None))):
            if value is None or optim is None:
                continue
            if self.capturable:
                getattr(self, name).fill_(float(value))
            else:
                setattr(self, name, float(value))
                for g in optim.param_groups:
                    g["lr"] = float(value)

    def _poll_range(self, device):
        """f16x3 recording forward: surface a raised range flag (an activation beyond 6e4) without a host sync; it
        shows up one or two iterations late at most.  Remedy: arch.mlp_train_precision = 'fp32'."""
        if self.graph.nerf.train_precision == "f16x3":
            ops.poll_mlp_status(device)

    def train_iteration(self, var: AttrDict):
        var = self.graph.get_ray_idx(self.opt, var)
        var, loss = self.nerf_step(var)
        if self.has_disc:
            var, dloss = self.disc_step(var)
            loss.update({k: v for k, v in dloss.items() if k != "all"})
            self.graph.discriminator.progress.data.fill_(self.it / self.max_iter)
        self.it += 1
        self.graph.patch_sampler.iterations = self.it
        self._poll_range(var.image.device)
        return var, loss


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
    """
    capturable = True

    def __init__(self, opt, graph: Graph, n_train: int, max_iter: int = 6000 * 189 // 8):
        super().__init__(opt, graph, n_train, max_iter)
        self._graph = None
        self._static_in = None
        self._static_loss = None

    def _body(self, var):
        opt = self.opt
        B, R = opt.batch_size, opt.patch_size ** 2
        var = self.graph.get_ray_idx(opt, var)
        if opt.nerf.sample_stratified and "jitter_rand" not in var:   # (a caller-supplied static tensor wins: tests)
            var.jitter_rand = torch.rand(B, R, opt.nerf.sample_intvs, 1, device=var.ray_idx.device)
        var, loss = self.nerf_step(var)
        if self.has_disc:
            var, dloss = self.disc_step(var)
            loss.update({k: v for k, v in dloss.items() if k != "all"})
        return {k: v.detach() for k, v in loss.items() if torch.is_tensor(v)}

    def capture(self, var: AttrDict, warmup: int = 3):
        """Warm up eagerly on a side stream (lazy inits: MIOpen solver search, weight packing, optimiser state), then
        record the step.  ``var`` fixes the shapes; its values are used for the warm-up iterations."""
        dev = var.image.device
        self.graph.patch_sampler.device_lo = torch.zeros((), device=dev)
        self._static_in = AttrDict({k: v.clone() for k, v in var.items() if torch.is_tensor(v)})
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self.graph.patch_sampler.update_device_bound()
                self._body(AttrDict(dict(self._static_in)))
                self._after_step()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self._graph = torch.cuda.CUDAGraph()
        self.optim_nerf.zero_grad(set_to_none=True)
        if self.has_disc:
            self.optim_disc.zero_grad(set_to_none=True)
        self.graph.patch_sampler.update_device_bound()          # outside the capture
        with torch.cuda.graph(self._graph):
            self._static_loss = self._body(AttrDict(dict(self._static_in)))
        return self

    def _after_step(self):
        if self.has_disc:
            self.graph.discriminator.progress.data.fill_(self.it / self.max_iter)
        self.it += 1
        self.graph.patch_sampler.iterations = self.it
        self.graph.nerf.mark_heads_dirty()

    def train_iteration(self, var: AttrDict):
        if self._graph is None:
            self.capture(var)
        for k, dst in self._static_in.items():
            dst.copy_(var[k], non_blocking=True)
        self.graph.patch_sampler.update_device_bound()          # one fill_ of the annealed bound
        self._graph.replay()
        self._after_step()
        self._poll_range(var.image.device)                      # outside the graph: event query + pinned copy
        return self._static_in, AttrDict(self._static_loss)
