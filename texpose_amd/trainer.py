"""One GAN training iteration of the adapt_st_gan stage on the HIP path, following the reference's
``Model.train_iteration`` (model/nerf_adapt_st_gan.py:108-202): patch coordinates -> nerf step (render,
gathers, discriminator on the fake patch, photometric / uncertainty / transient / feature / GAN losses,
Adam) -> discriminator step (real + R1 penalty + fake, RMSprop).  Used by bench.py and the tests; the
reference's own engine runs unchanged against texpose_amd.graph.Graph.

No optimiser step is ever applied from a bad forward.  The reference asserts every loss term finite before
``backward`` (model/base.py:153-154); here the check sits before ``optimizer.step()`` and also covers the range
flag of the f16x3 recording forward (an activation beyond the fp16 range):
  * eager trainer: one blocking read per optimiser step of [range flag, isfinite(loss)].  A raised range flag
    drops the gradients, switches the recording forward to the exact-fp32 kernel for good and repeats the step;
    a non-finite loss raises FloatingPointError with the parameters untouched;
  * hipGraph trainer: the same predicate is evaluated on the device inside the captured step and gates the
    update (gradients and learning rate multiplied by 0/1: Adam and RMSprop then leave every parameter bit-for-bit
    unchanged); the host learns about it from an asynchronous copy one or two replays later, and either
    re-captures with the fp32 recording forward (range flag) or raises (non-finite loss).

Data parallel: pass a process group after texpose_amd.dist.init_distributed(); gradients of each optimiser
are averaged with ONE flat all-reduce after all backward calls of that step (tools/train_dp.py is the entry point).
"""
from __future__ import annotations

import warnings

import torch

from . import dist as tdist
from . import knobs, ops
from .graph import Graph, summarize_loss  # noqa: F401  (summarize_loss: re-exported, the reference keeps it on the model)
from .options import AttrDict
from .options import AttrDict as edict


def _finite(t):
    """isfinite of a tensor in two launches instead of torch's four: x - x is 0 exactly for finite x, NaN otherwise."""
    return (t - t) == 0


class FusedRMSprop(torch.optim.RMSprop):
    """torch.optim.RMSprop with the reference's settings (no momentum, not centred, no weight decay), its state-dict
    layout (``square_avg``, ``step``) and its arithmetic, but the whole step as ONE launch (K10, csrc/rmsprop.hip) that
    reads a tensor learning rate on the device.  CUDA float32 parameters only; anything else uses the stock step."""
    gate = None                              # int32 device words: a non-zero word withholds the update

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None:
            raise NotImplementedError("closure")
        todo = []
        for group in self.param_groups:                    # decide for ALL groups before anything is launched
            ps = [p for p in group["params"] if p.grad is not None]
            plain = (group["momentum"] == 0 and not group["centered"] and group["weight_decay"] == 0 and not group["maximize"]
                     and all(p.is_cuda and p.dtype == torch.float32 and not p.grad.is_sparse for p in ps))
            if not plain:
                return super().step()
            todo.append((group, ps))
        for group, ps in todo:
            for p in ps:
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.zeros((), dtype=torch.float32, device=p.device) if group["capturable"] else torch.tensor(0.0)
                    st["square_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            if ps:
                steps = [self.state[p]["step"] for p in ps]
                on_device = all(st.is_cuda for st in steps)        # (capturable: the launch advances them; else a host-side add)
                ops.rmsprop_step([p.data for p in ps], [p.grad.contiguous() for p in ps], [self.state[p]["square_avg"] for p in ps],
                                 group["lr"], group["alpha"], group["eps"], gate=self.gate, steps=steps if on_device else None)
                if not on_device:
                    torch._foreach_add_(steps, 1)
        return None


class FusedAdam(torch.optim.Adam):
    """torch.optim.Adam (no weight decay, no amsgrad) with its state-dict layout (``step``, ``exp_avg``, ``exp_avg_sq``),
    but every tensor of the step in ONE launch (K13 tp_adam_step) that reads a tensor learning rate and the step gate on
    the device.  Capturable CUDA float32 groups only; anything else uses the stock step."""
    gate = None                              # int32 device words: a non-zero word withholds the whole step

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None:
            raise NotImplementedError("closure")
        todo = []
        for group in self.param_groups:
            ps = [p for p in group["params"] if p.grad is not None]
            plain = (group["capturable"] and not group["amsgrad"] and group["weight_decay"] == 0 and not group["maximize"]
                     and all(p.is_cuda and p.dtype == torch.float32 and not p.grad.is_sparse for p in ps))
            if not plain:
                return super().step()
            for p in ps:
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.zeros((), dtype=torch.float32, device=p.device)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            todo.append((group, ps))
        # groups with the same hyper-parameters (the reference's three groups share them) go into one launch
        merged = {}
        for group, ps in todo:
            lr = group["lr"]
            key = (lr.data_ptr() if torch.is_tensor(lr) else float(lr), tuple(group["betas"]), group["eps"])
            merged.setdefault(key, (group, []))[1].extend(ps)
        for group, ps in merged.values():
            if ps:
                ops.adam_step([p.data for p in ps], [p.grad.contiguous() for p in ps], [self.state[p]["exp_avg"] for p in ps],
                              [self.state[p]["exp_avg_sq"] for p in ps], [self.state[p]["step"] for p in ps], group["lr"],
                              group["betas"][0], group["betas"][1], group["eps"], gate=self.gate)
        return None


class GanTrainer:
    capturable = False                       # optimiser state on the device (required inside a hipGraph)

    def __init__(self, opt, graph: Graph, n_train: int, max_iter: int = 6000 * 189 // 8, group=None):
        self.opt, self.graph, self.max_iter, self.it = opt, graph, max_iter, 0
        if not hasattr(graph, "latent_vars_trans"):
            graph.attach_latents(n_train, opt)
        # group 0 holds EVERY nerf parameter like the reference's optimiser (model/nerf_adapt_st_gan.py:62-68: the frozen
        # trunk and `progress` never get a gradient and are skipped by Adam), so that a saved `optim_nerf` state has
        # the reference's group sizes (33 / 1 / 1) and loads on resume in either direction
        nerf_params = list(graph.nerf.parameters())
        self.nerf_group = [p for p in nerf_params if p.requires_grad] + list(graph.latent_vars_light.parameters()) + \
            list(graph.latent_vars_trans.parameters())
        # a captured step reads its learning rates from device memory, so that a scheduler (the reference decays the
        # nerf rate per epoch, ExponentialLR, model/nerf_adapt_st_gan.py:73-84) can change them between replays
        dev = nerf_params[0].device
        mk = (lambda v: torch.tensor(float(v), device=dev)) if self.capturable else float
        self.lr_nerf = mk(opt.optim.lr)
        self.lr_nerf_used = mk(opt.optim.lr)               # what the optimiser reads (captured: lr x the step gate)
        # captured trainer: every tensor of the Adam step in one launch (torch's foreach implementation with a tensor learning
        # rate is ~40 multi-tensor + 2 per-parameter launches, its fused one 4 launches / 60 us for these 25 small tensors)
        fused_own = self.capturable and dev.type == "cuda" and not knobs.K.no_fused_adam
        adam = FusedAdam if fused_own else torch.optim.Adam
        self.optim_nerf = adam([dict(params=nerf_params, lr=self.lr_nerf_used),
                                dict(params=graph.latent_vars_light.parameters(), lr=self.lr_nerf_used),
                                dict(params=graph.latent_vars_trans.parameters(), lr=self.lr_nerf_used)],
                               capturable=self.capturable)
        self.has_disc = hasattr(graph, "discriminator") and opt.gan is not None
        if self.has_disc:
            self.disc_group = [p for p in graph.discriminator.parameters()]
            self.lr_disc = mk(opt.optim_disc.lr)
            self.lr_disc_used = mk(opt.optim_disc.lr)
            rms = FusedRMSprop if self.capturable and dev.type == "cuda" and not knobs.K.no_fused_rmsprop else torch.optim.RMSprop
            self.optim_disc = rms([dict(params=self.disc_group, lr=self.lr_disc_used)], capturable=self.capturable)
        self.red_nerf = tdist.FlatGradAllReducer(self.nerf_group, group=group)
        self.red_disc = tdist.FlatGradAllReducer(self.disc_group, group=group) if self.has_disc else None
        self.skipped_steps = 0                   # optimiser steps withheld because the forward was flagged

    @staticmethod
    def _toggle(module, flag):
        for p in module.parameters():
            p.requires_grad_(flag)

    # ------------------------------------------------------------------ guards (see the module docstring)
    def _uses_f16x3(self):
        return self.graph.nerf.train_precision == "f16x3"

    def _guard_nerf(self, var, loss):
        """Eager: True = apply the step.  Blocking read of [range flag, isfinite(total)] (the reference syncs once per
        loss term at this point)."""
        dev = loss.all.device
        bad = (~torch.isfinite(loss.all.detach())).to(torch.int32).reshape(1)
        word = torch.cat([ops.mlp_status(dev) & 1, bad]) if self._uses_f16x3() else torch.cat([torch.zeros_like(bad), bad])
        # data parallel: the decision is GLOBAL -- every rank withholds, repeats or raises together (a rank acting alone
        # would leave the others in the gradient all-reduce, or apply gradients its peers dropped)
        word = tdist.all_reduce_flags(word, group=self.red_nerf.group)
        flag, not_finite = word.tolist()
        finite = not not_finite
        if flag & 1:
            ops.mlp_status(dev).zero_()
            return False
        if not finite:
            raise FloatingPointError("non-finite nerf loss (no update applied): " +
                                     ", ".join("%s=%g" % (k, float(v.detach())) for k, v in loss.items()))
        return True

    def _guard_disc(self, total):
        bad = (~torch.isfinite(total.detach())).to(torch.int32).reshape(1)
        if int(tdist.all_reduce_flags(bad, group=self.red_disc.group if self.red_disc else None)):
            raise FloatingPointError("non-finite discriminator loss on some rank (no update applied); here: %g" % float(total))
        return True

    def _weight(self, exponent, dev):
        """10^exponent as a 0-dim tensor on ``dev``, made once (a captured step may not create tensors from host values)."""
        cache = self.__dict__.setdefault("_weights", {})
        key = (float(exponent), str(dev))
        if key not in cache:
            cache[key] = torch.tensor(10 ** float(exponent), device=dev)
        return cache[key]

    def _weighted_total(self, loss, flags=None, defer=False):
        """``loss.all`` = detached sum 10^w_k loss_k (logging, the finite check); returns the terms and their weights.
        ``flags``: the step-gate update for this total (ops.step_flags arguments), folded into the same launch where possible."""
        opt = self.opt
        keys = [k for k in loss if k != "all" and opt.loss_weight[k] is not None]
        for k in loss:
            assert k == "all" or (k in opt.loss_weight and loss[k].shape == ()), k
        dev = loss[keys[0]].device
        ws = [self._weight(opt.loss_weight[k], dev) for k in keys]
        cache = self.__dict__.setdefault("_weight_vectors", {})
        vec_key = (tuple(keys), str(dev))
        if vec_key not in cache:
            cache[vec_key] = torch.stack(ws)
        with torch.no_grad():
            if dev.type == "cuda" and len(keys) <= 16 and all(loss[k].dtype == torch.float32 for k in keys):
                loss.all = ops.weighted_sum([loss[k] for k in keys], [10 ** float(opt.loss_weight[k]) for k in keys], flags=flags,
                                            defer=defer and flags is not None)   # K13: one launch (or a side job of the losses' backward)
            else:
                loss.all = torch.dot(torch.stack([loss[k].detach() for k in keys]), cache[vec_key])
                if flags is not None:
                    ops.step_flags(loss.all, flags["bad"], flags["word_finite"], flags["snapshot"], status=flags.get("status"),
                                   word_status=flags.get("word_status", 0))
        return [loss[k] for k in keys], ws

    def _backward_weighted(self, loss):
        """Back-propagate sum 10^w_k loss_k (reference model/base.py:145-157 + ``loss.all.backward()``) without building the
        scalar graph: the terms are the roots and the weights their cotangents (no mul / add / MulBackward launches per
        term)."""
        terms, ws = self._weighted_total(loss)
        torch.autograd.backward(terms, ws)
        return loss

    def nerf_forward_loss(self, var, stage=None):
        """First half of the nerf step: render, losses, weighted total -- everything the discriminator step needs.
        ``stage``: "render" (up to the discriminator's patch stacks; returns (var, None)) / "consume" (the rest), see
        Graph.nerf_forward."""
        opt, g = self.opt, self.graph
        if stage != "consume":
            if self.has_disc:
                self._toggle(g.discriminator, False)
            self.optim_nerf.zero_grad(set_to_none=True)
        v = g.nerf_forward(opt, var, mode="train", stage=stage)
        if stage == "render":
            return v, None
        loss = g.compute_loss(opt, v, mode="train", train_step="nerf")
        return v, loss

    def nerf_apply(self):
        self.red_nerf.average()                  # (several ranks: pack | ONE flat all-reduce | .grad = views of the flat buffer)
        self.optim_nerf.step()
        # the packed weight image is re-built when a head parameter's version changes; fused optimiser kernels (and a
        # replayed hipGraph) update parameters WITHOUT bumping tensor versions, so say it explicitly
        self.graph.nerf.mark_heads_dirty()

    def nerf_step(self, var):
        g = self.graph
        for attempt in range(2):
            v, loss = self.nerf_forward_loss(var)
            self._backward_weighted(loss)
            if self._guard_nerf(v, loss):
                break
            # range flag of the f16x3 recording forward: nothing of this forward / backward is used
            self.skipped_steps += 1
            if attempt == 1:
                raise ops._lib.TexposeLibraryError("the f16x3 range flag is raised although the fp32 recording forward is selected")
            warnings.warn("texpose_amd: an activation left the fp16 range of the f16x3 recording forward; the step was "
                          "repeated and training continues with arch.mlp_train_precision='fp32'")
            g.nerf.train_precision = "fp32"
        self.nerf_apply()
        return v, loss

    def _disc_schedule(self, x):
        """The explicit launch schedule of the discriminator step (K16, texpose_amd/disc_step.py) when it covers this
        discriminator and these tensors, else None (CPU tensors, other ladders / GAN losses: the autograd form below)."""
        if knobs.K.disc_autograd:
            return None
        sched = self.__dict__.get("_disc_sched")
        if sched is None or sched.disc is not self.graph.discriminator:
            from .disc_step import DiscStepSchedule
            sched = self._disc_sched = DiscStepSchedule(self.graph.discriminator)
        return sched if sched.eligible(self.opt, x) else None

    def disc_step(self, var, apply=True):
        opt, g = self.opt, self.graph
        self._toggle(g.discriminator, True)
        self.optim_disc.zero_grad(set_to_none=True)
        if var.rgb.is_cuda and (var.get("disc_patches_for") is var.ray_idx or ("gathered" in var and var.get("gathered_for") is var.ray_idx)):
            real, fake, stack = g.disc_patch_stacks(opt, var)
            sched = self._disc_schedule(real)
            if sched is not None:
                return self._disc_step_scheduled(sched, var, real, fake, stack, apply)
        var = g.disc_forward(opt, var, mode="train")
        loss = g.compute_loss(opt, var, mode="train", train_step="disc")
        # The reference back-propagates the three terms one after the other (real, R1 penalty, fake: :139-160).  The sum of
        # the three gradients is the gradient of the weighted sum: ONE traversal with three roots visits the real-patch
        # graph once instead of twice and accumulates into every parameter once instead of three times.
        terms = edict(gan_disc_real=loss.gan_disc_real)
        if opt.loss_weight.gan_reg_real is not None:          # R1: double backward through the discriminator
            terms.gan_reg_real = g.compute_grad2_mean(opt, var.d_real_disc, var.patch_real)
        terms.gan_disc_fake = loss.gan_disc_fake
        total = self._backward_weighted(terms).all
        if "gan_reg_real" in terms:
            # the reference logs the WEIGHTED penalty: it scales the tensor it has just stored, in place (:151-153)
            loss.gan_reg_real = self._weight(opt.loss_weight.gan_reg_real, total.device) * terms.gan_reg_real.detach()
        if apply:
            self.disc_apply(total)
        else:
            self._disc_total = total
        return var, loss

    def _disc_step_scheduled(self, sched, var, real, fake, stack, apply):
        """`disc_step` through the explicit schedule: same losses, same gradients (sums of two terms in another order), about a
        third fewer launches -- and no autograd graph."""
        lw = self.opt.loss_weight
        w = lambda k: 10 ** float(lw[k])
        with torch.no_grad():
            res = sched.run(real, fake, var.ray_scales, w("gan_disc_real"), w("gan_disc_fake"),
                            None if lw.gan_reg_real is None else w("gan_reg_real"), real_stack=stack)
        return self._disc_step_scheduled_post(var, res, real, fake, apply)

    def _disc_step_scheduled_post(self, var, res, real, fake, apply):
        """What follows the schedule's launches: the step's outputs, the loss total (+ gate), the optimiser step."""
        lw = self.opt.loss_weight
        w = lambda k: 10 ** float(lw[k])
        with torch.no_grad():
            var.patch_real, var.patch_fake, var.d_real_disc, var.d_fake_disc = real, fake, res.d_real, res.d_fake
            loss = edict(gan_disc_real=res.gan_disc_real)
            if res.gan_reg_real is not None:
                loss.gan_reg_real = res.gan_reg_real
            loss.gan_disc_fake = res.gan_disc_fake
            keys = list(loss.keys())                               # (real, R1, fake: the order of the autograd form's total)
            stepped = res.get("total") is not None                 # (total, gate and RMSprop rode in the schedule's last two launches)
            flags = self._disc_gate_flags()                        # (captured step: the gate update rides in the same launch)
            total = res.total if stepped else ops.weighted_sum([loss[k] for k in keys], [w(k) for k in keys], flags=flags)
            if res.gan_reg_real is not None:                       # logged WEIGHTED, as the reference does (:151-153)
                loss.gan_reg_real = res.gan_reg_real_weighted
        self._disc_flagged = flags is not None
        if stepped:
            assert apply and flags is not None
            self._disc_total = total
            return var, loss
        if apply:
            self.disc_apply(None if flags is not None else total)
        else:
            self._disc_total = total
        return var, loss

    def _disc_gate_flags(self):
        """ops.step_flags arguments of the discriminator gate when the loss total's launch should carry them (captured step)."""
        return None

    def disc_apply(self, total):
        self._guard_disc(total)
        self.red_disc.average()
        self.optim_disc.step()

    def set_lr(self, nerf: float = None, disc: float = None):
        """Learning rates for the following iterations (eager: param groups; captured: the device scalars the graph reads)."""
        for value, name, optim in ((nerf, "lr_nerf", self.optim_nerf), (disc, "lr_disc", getattr(self, "optim_disc", None))):
            if value is None or optim is None:
                continue
            if self.capturable:
                getattr(self, name).fill_(float(value))
                getattr(self, name + "_used").fill_(float(value))
            else:
                setattr(self, name, float(value))
                setattr(self, name + "_used", float(value))
                for g in optim.param_groups:
                    g["lr"] = float(value)

    def load_optim_state(self, optim_nerf=None, optim_disc=None):
        """Resume: optimiser state dicts of a checkpoint (texpose_amd.checkpoint.restore_checkpoint passes them on).
        ``Optimizer.load_state_dict`` replaces each group's ``lr`` by the saved value; a captured step reads the rate
        from ``lr_*_used``, so the loaded value is copied there and the groups are pointed back at it."""
        for sd, name, optim in ((optim_nerf, "lr_nerf", self.optim_nerf), (optim_disc, "lr_disc", getattr(self, "optim_disc", None))):
            if sd is None or optim is None:
                continue
            optim.load_state_dict(sd)
            self._adopt_group_lr(name, optim)
            self._adopt_capturable(optim)

    def _adopt_capturable(self, optim):
        """``Optimizer.load_state_dict`` takes ``capturable`` and the step counters from the SAVED groups / state: a checkpoint
        of an eager trainer (or of the reference) carries capturable=False and CPU ``step`` tensors, one of a captured trainer
        the opposite.  Put both back to what THIS trainer needs, so that state loads in either direction."""
        for g in optim.param_groups:
            if "capturable" in g:
                g["capturable"] = bool(self.capturable)
            for p in g["params"]:
                st = optim.state.get(p)
                if not st or "step" not in st:
                    continue
                step = st["step"]
                value = float(step)
                if self.capturable:
                    st["step"] = torch.tensor(value, dtype=torch.float32, device=p.device)
                else:
                    st["step"] = torch.tensor(value, dtype=torch.float32)

    def _adopt_group_lr(self, name, optim):
        used = getattr(self, name + "_used")
        for g in optim.param_groups:
            if g["lr"] is not used:
                value = float(g["lr"])
                if self.capturable:
                    getattr(self, name).fill_(value)
                    used.fill_(value)
                    g["lr"] = used
                else:
                    setattr(self, name, value)
                    setattr(self, name + "_used", value)

    def flush_flags(self):
        """The eager trainer reads its guards before every optimiser step: nothing is pending (see GraphedGanTrainer.flush_flags)."""
        return [0, 0, 0]

    finish = flush_flags

    def train_iteration(self, var: AttrDict):
        var = self.graph.get_ray_idx(self.opt, var)
        var, loss = self.nerf_step(var)
        if self.has_disc:
            var, dloss = self.disc_step(var)
            loss.update({k: v for k, v in dloss.items() if k != "all"})
            self.graph.discriminator.progress.data.fill_(self.it / self.max_iter)
        self.it += 1
        self.graph.patch_sampler.iterations = self.it
        return var, loss


def __getattr__(name):
    """`GraphedGanTrainer` and the stream probe live in modules of their own (graphed_trainer.py, queue_probe.py) and stay importable
    from here: ``from texpose_amd.trainer import GraphedGanTrainer`` (resolved lazily: graphed_trainer imports this module)."""
    if name in ("GraphedGanTrainer", "_FormUnavailable"):
        from . import graphed_trainer
        return getattr(graphed_trainer, name)
    if name in ("distinct_queue_streams", "streams_after_collectives", "LAST_QUEUE_PROBE"):
        from . import queue_probe
        return getattr(queue_probe, name)
    raise AttributeError("module %r has no attribute %r" % (__name__, name))
