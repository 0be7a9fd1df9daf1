"""Modules that sit NEXT to the hot path in a training iteration: the PatchGAN discriminator
and the VGG perceptual loss.  On the GPU every layer runs on this repo's HIP kernels, differentiable to the order the
step needs (the R1 penalty is a double backward): spectral normalisation (K7: ~600 of the ~1,350 launches of an iteration in
stock PyTorch), stride-2 4x4 convolutions (K11), InstanceNorm + LeakyReLU (K9), the full-map convolution (K15), the
scale-conditioned head (K14); the feature network's 3x3 convolutions (K12) and 2x2 max pools (K13).  CPU tensors (goldens,
contract tests) go through the stock torch modules.  The discriminator STEP of a training iteration does not go through
autograd at all on the GPU: texpose_amd/disc_step.py (K16).  Written from the reference's architecture description
(layers/discriminator.py:8-173, layers/perceptual_loss.py:8-45), state-dict compatible with it:
``main.{0,3,6}.weight_{orig,u,v}``, ``final.{1,3,5}.weight_{orig,u,v}``, ``progress`` for patch_size 16.

PerceptualLoss needs torchvision's ImageNet VGG19 weights, which are not available offline: with
``pretrained_state=None`` it is randomly initialised (same structure and cost; parity unpinned).
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import knobs


def _pow2_map(n: int) -> bool:
    return n >= 8 and (n & (n - 1)) == 0


class SNConv2d(nn.Module):
    """Conv2d (no bias) under spectral normalisation with the state-dict layout of
    ``torch.nn.utils.spectral_norm(nn.Conv2d(...))``: ``weight_orig`` (parameter), ``weight_u`` / ``weight_v``
    (buffers).  The normalised weight is NOT computed here: ``spectral_weights`` does it for all convolutions of a
    module at once (five launches instead of ~15 per convolution)."""

    def __init__(self, c_in, c_out, k, stride, pad):
        super().__init__()
        conv = nn.Conv2d(c_in, c_out, (k, k), (stride, stride), (pad, pad), bias=False)   # same init + RNG order as stock
        self.weight_orig = nn.Parameter(conv.weight.detach().clone())
        w2 = self.weight_orig.detach().reshape(c_out, -1)
        self.register_buffer("weight_u", F.normalize(w2.new_empty(w2.shape[0]).normal_(0, 1), dim=0, eps=1e-12))
        self.register_buffer("weight_v", F.normalize(w2.new_empty(w2.shape[1]).normal_(0, 1), dim=0, eps=1e-12))
        self.stride, self.padding = (stride, stride), (pad, pad)

    def forward(self, x, weight):
        if x.is_cuda and self.padding == (0, 0) and tuple(x.shape[-2:]) == tuple(weight.shape[-2:]):
            # the kernel covers the whole map (the three 1x1 layers of ``final`` on a 1x1 map, and main's last 4x4 conv
            # on its 4x4 map): a plain GEMM -- one launch in each derivative order instead of MIOpen's conv + layout
            # transposes (and its naive double-backward fallbacks)
            x2, w2 = x.flatten(1), weight.flatten(1)
            if x2.shape[0] <= 256 and x2.shape[1] >= 1024:              # K15: one pass over the weight, one launch per node
                from . import autograd_ops
                return autograd_ops.skinny_linear(x2, w2)[:, :, None, None]
            _library_kernel_note("a full-map convolution of %d x %d (K15 takes at most 256 rows of at least 1024 columns)" % tuple(x2.shape))
            return F.linear(x2, w2)[:, :, None, None]
        if (x.is_cuda and tuple(weight.shape[-2:]) == (4, 4) and self.stride == (2, 2) and self.padding == (1, 1)
                and _pow2_map(x.shape[-2]) and _pow2_map(x.shape[-1])):
            from . import autograd_ops                                 # K11: one launch per derivative node
            return autograd_ops.conv4s2(x, weight)
        if x.is_cuda:
            _library_kernel_note("a %s convolution, stride %s, on a %d x %d map (K11 takes 4x4 stride-2 convolutions on power-of-two maps)"
                                 % (tuple(weight.shape[-2:]), self.stride, x.shape[-2], x.shape[-1]))
        return F.conv2d(x, weight, None, self.stride, self.padding)


_library_notes = set()


def _library_kernel_note(what):
    """Said once per shape: a PatchGAN geometry outside the HIP kernels' coverage runs this layer through the stock library kernel
    (MIOpen / rocBLAS) -- same values; none of BASELINE's configurations gets here."""
    if what not in _library_notes:
        _library_notes.add(what)
        import warnings
        warnings.warn("texpose_amd: %s runs through the stock library kernel, not a HIP kernel of this package" % what)


class _SpectralWeightsHip(torch.autograd.Function):
    """W_sn = W / sigma for a list of SNConv2d modules through tp_sn_fwd / tp_sn_bwd (u, v constants in the backward,
    as in torch).  First-order only: the R1 double backward reaches W_sn through the convolutions, never through here."""

    @staticmethod
    def forward(ctx, convs, training, *weights):
        from . import ops
        us, vs = [c.weight_u for c in convs], [c.weight_v for c in convs]
        # later forwards of the same iteration advance u / v in place: keep this forward's copies for its backward (written
        # by the normalisation launch itself; none when no weight wants a gradient, e.g. the nerf step)
        if training and any(ctx.needs_input_grad):
            outs, sigmas, ctx.us, ctx.vs = ops.spectral_norm_fwd([w.detach() for w in weights], us, vs, training, keep_uv=True)
        else:
            outs, sigmas = ops.spectral_norm_fwd([w.detach() for w in weights], us, vs, training)
            ctx.us, ctx.vs = us, vs
        ctx.sigmas = sigmas
        ctx.save_for_backward(*outs)
        return tuple(outs)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, *grads):
        from . import ops
        outs = ctx.saved_tensors
        grads = [g if g is not None else torch.zeros_like(o) for g, o in zip(grads, outs)]
        gw = ops.spectral_norm_bwd(grads, outs, ctx.us, ctx.vs, ctx.sigmas)
        return (None, None) + tuple(gw)


def spectral_weights(convs, training: bool):
    """Normalised weights of ``convs`` (SNConv2d list) with torch.nn.utils.spectral_norm semantics: in training mode
    one power iteration per call updates weight_u / weight_v in place.  GPU tensors go through the fused HIP kernels;
    CPU tensors (state-dict / contract tests, golden generation) use the same arithmetic in plain torch ops."""
    ws = [c.weight_orig for c in convs]
    if ws[0].is_cuda:
        return list(_SpectralWeightsHip.apply(convs, bool(training), *ws))
    outs = []
    for c in convs:
        w2 = c.weight_orig.reshape(c.weight_orig.shape[0], -1)
        u, v = c.weight_u, c.weight_v
        if training:
            with torch.no_grad():
                v.copy_(F.normalize(torch.mv(w2.t(), u), dim=0, eps=1e-12))
                u.copy_(F.normalize(torch.mv(w2, v), dim=0, eps=1e-12))
            u, v = u.clone(), v.clone()
        sigma = torch.dot(u, torch.mv(w2, v))
        outs.append(c.weight_orig / sigma)
    return outs


class Discriminator(nn.Module):
    """patch [B, 3(+6 geo), p, p] (+ patch scale) -> logit [B] (reference layers/discriminator.py:7-141).
    ``gan.L_nocs`` / ``gan.L_normal`` (empty in the shipped yaml): positional encodings of the nocs / normal channels appended to the
    input (:128-136), ``gan.geo_c2f``: coarse-to-fine weights on them from ``progress`` (:157-165).  Formed with torch element-wise
    ops in front of the ladder (differentiable twice: the R1 penalty goes through them); the discriminator step then runs through
    autograd over the HIP kernels instead of the explicit schedule K16 (disc_step.DiscStepSchedule covers the shipped configuration)."""

    def __init__(self, opt, ndf: int = 64):
        super().__init__()
        g = opt.gan
        self.scale_conditional, self.geo_conditional = bool(g.scale_conditional), bool(g.geo_conditional)
        self.L_scale = g.L_scale
        self.L_nocs, self.L_normal = g.L_nocs, g.L_normal
        self.c2f_range = None if g.geo_c2f is None else (float(g.geo_c2f[0]), float(g.geo_c2f[1]))
        if (self.L_nocs or self.L_normal or self.c2f_range is not None) and not (self.geo_conditional and (self.L_nocs or self.L_normal)):
            raise ValueError("gan.L_nocs / L_normal / geo_c2f need gan.geo_conditional and at least one encoding (reference :18-20)")
        if self.L_normal and self.L_normal != self.L_nocs:
            # the reference encodes the normals with L = L_nocs (layers/discriminator.py:134) while it sizes the first convolution
            # with L_normal (:24): any other combination fails there with a channel mismatch
            raise ValueError("gan.L_normal must equal gan.L_nocs (the reference encodes the normals with L_nocs)")
        nc = 3 + (6 if self.geo_conditional else 0) + 6 * ((self.L_nocs or 0) + (self.L_normal or 0))
        p = opt.patch_size
        if p not in (16, 32, 64, 128):
            raise ValueError("patch_size must be 16, 32, 64 or 128")
        self.progress = nn.Parameter(torch.tensor(0.))
        self._sn_queue = []                      # prefetched normalised weight sets (prefetch_spectral_weights)
        conv = SNConv2d
        final_dim = ndf if self.scale_conditional else 1
        if self.scale_conditional:
            # registered BEFORE ``main``, as in the reference (layers/discriminator.py:30-39): parameters() -- and with it the
            # parameter numbering inside a saved ``optim_disc`` state -- runs progress, final.{1,3,5}, main.{0,3,6}
            # 2^l pi, l < L_scale (a constant: kept as a non-persistent buffer instead of three launches per forward)
            self.register_buffer("scale_freq", (2 ** torch.arange(self.L_scale, dtype=torch.float32)) * math.pi,
                                 persistent=False)
            c = ndf + 2 * self.L_scale + 1
            self.final = nn.Sequential(nn.LeakyReLU(0.2), conv(c, ndf, 1, 1, 0), nn.LeakyReLU(0.2, inplace=True),
                                       conv(ndf, ndf, 1, 1, 0), nn.LeakyReLU(0.2, inplace=True), conv(ndf, 1, 1, 1, 0))
        # stride-2 ladder down to 8x8 with 256 channels; the first stage of the 64/128 ladders has no norm
        widths = {16: [ndf * 4], 32: [ndf * 2, ndf * 4], 64: [ndf, ndf * 2, ndf * 4],
                  128: [ndf // 2, ndf, ndf * 2, ndf * 4]}[p]
        blocks, c_in = [], nc
        for i, w in enumerate(widths):
            blocks.append(conv(c_in, w, 4, 2, 1))
            if not (i == 0 and p >= 64):
                blocks.append(nn.InstanceNorm2d(w))
            blocks.append(nn.LeakyReLU(0.2, inplace=True))
            c_in = w
        blocks += [conv(c_in, ndf * 8, 4, 2, 1), nn.InstanceNorm2d(ndf * 8), nn.LeakyReLU(0.2, inplace=True),
                   conv(ndf * 8, final_dim, 4, 1, 0)]
        self.main = nn.Sequential(*blocks)

    @staticmethod
    def _run(seq, x, weights):
        mods, i = list(seq), 0
        while i < len(mods):
            m = mods[i]
            if (isinstance(m, SNConv2d) and x.is_cuda and i + 2 < len(mods) and isinstance(mods[i + 1], nn.InstanceNorm2d)
                    and isinstance(mods[i + 2], nn.LeakyReLU) and not mods[i + 1].affine and not mods[i + 1].track_running_stats
                    and tuple(m.weight_orig.shape[-2:]) == (4, 4) and m.stride == (2, 2) and m.padding == (1, 1)
                    and not weights[0].requires_grad and Discriminator._first_order() and Discriminator._fused_stage_ok(x)):
                # constant weight (the nerf step's pass): convolution + InstanceNorm + LeakyReLU as one launch (K11 epilogue)
                from . import autograd_ops
                x = autograd_ops.conv4s2_inorm(x, weights.pop(0), mods[i + 1].eps, mods[i + 2].negative_slope)
                i += 3
                continue
            if isinstance(m, SNConv2d):
                x = m(x, weights.pop(0))
            elif (x.is_cuda and isinstance(m, nn.InstanceNorm2d) and i + 1 < len(mods) and isinstance(mods[i + 1], nn.LeakyReLU)
                  and not m.affine and not m.track_running_stats):
                # InstanceNorm + LeakyReLU as one kernel per derivative order (K9, csrc/inorm_lrelu.hip); CPU tensors
                # (contract tests, goldens) keep the stock modules
                from . import autograd_ops
                x = autograd_ops.inorm_lrelu(x, m.eps, mods[i + 1].negative_slope)
                i += 1
            elif isinstance(m, nn.LeakyReLU):
                # out of place even where the module says inplace: the input is a view of a GEMM output, and an in-place
                # op on a view costs autograd a CopySlices + AsStridedBackward pair (6 launches) per derivative order
                x = F.leaky_relu(x, m.negative_slope)
            else:
                x = m(x)
            i += 1
        return x

    @staticmethod
    def _first_order():
        """The caller declared that this pass is differentiated once, wrt its input (autograd_ops.first_order_only): the fused
        frozen-weight nodes are `once_differentiable`; without the declaration the differentiable K11 / K9 / K15 / K14 nodes run."""
        from . import autograd_ops
        return autograd_ops.first_order_declared() or not torch.is_grad_enabled()

    @staticmethod
    def _fused_stage_ok(x):
        from . import ops
        return ops.conv4s2_fwd_inorm_supported(x)

    def _tail_fusable(self, x, weights):
        """GPU, scale-conditioned plain head, constant weights (no gradient wrt them is being recorded), a last ladder convolution
        that covers its whole map, at most 16 patches."""
        from . import ops
        if not (x.is_cuda and self.scale_conditional and self._plain_head()) or any(w.requires_grad for w in weights):
            return False
        if not self._first_order():
            return False
        last = list(self.main)[-1]
        if not (isinstance(last, SNConv2d) and last.padding == (0, 0) and last.stride == (1, 1)):
            return False
        n_down = len([m for m in self.main if isinstance(m, SNConv2d)]) - 1
        kh, kw = last.weight_orig.shape[-2:]
        if (x.shape[-2] >> n_down, x.shape[-1] >> n_down) != (kh, kw):
            return False
        k_in = last.weight_orig.shape[1] * kh * kw
        return x.shape[0] <= ops.DISC_TAIL_MAX_ROWS and k_in % 4 == 0 and k_in >= 1024 and not knobs.K.no_disc_tail

    def _plain_head(self):
        """``final`` is LeakyReLU, 1x1, LeakyReLU, 1x1, LeakyReLU, 1x1 with one slope (what __init__ builds)."""
        mods = list(self.final)
        return (len(mods) == 6 and all(isinstance(m, nn.LeakyReLU) for m in mods[0::2])
                and len({m.negative_slope for m in mods[0::2]}) == 1
                and all(isinstance(m, SNConv2d) and tuple(m.weight_orig.shape[-2:]) == (1, 1) for m in mods[1::2]))

    def sn_convs(self):
        """The spectrally normalised convolutions in the order `forward` uses their weights: ladder, then head."""
        return [m for m in list(self.main) + (list(self.final) if self.scale_conditional else []) if isinstance(m, SNConv2d)]

    def prefetch_spectral_weights(self, n_calls: int, append: bool = False):
        """Run the power iterations / normalisations of the NEXT ``n_calls`` training-mode forwards now, in order (each advances
        weight_u / weight_v once, exactly as those forwards would), and queue the results.  They depend on the weights only, not on
        any input: the captured training step issues the three of an iteration (nerf step's D(fake), D(real), D(fake)) on a side
        stream while the render's MLP kernel runs, which takes 5 launches off each discriminator pass.  Consumers that do not
        differentiate through the normalisation take them with `take_prefetched_weights`; an unconsumed queue is an error."""
        from . import ops
        if not self.training:
            raise RuntimeError("prefetch_spectral_weights: training mode only")
        if self._sn_queue and not append:          # (`append`: the caller issues an iteration's sets in two parts, trainer._seg_sn)
            raise RuntimeError("prefetch_spectral_weights: %d prefetched weight sets were never used" % len(self._sn_queue))
        convs = self.sn_convs()
        W, U, V = [c.weight_orig.detach() for c in convs], [c.weight_u for c in convs], [c.weight_v for c in convs]
        if 1 < n_calls <= ops.SN_MAX_SETS and not knobs.K.no_sn_sets:
            # all power iterations first (2 launches each), ONE normalisation launch for the n_calls sets: 2 n + 1 launches instead of 3 n
            for outs, sigmas, us, vs in ops.spectral_norm_fwd_sets(W, U, V, n_calls):
                self._sn_queue.append((outs, sigmas, us, vs, torch.cuda.current_stream(outs[0].device)))
            return
        for _ in range(n_calls):
            outs, sigmas, us, vs = ops.spectral_norm_fwd(W, U, V, True, keep_uv=True)
            self._sn_queue.append((outs, sigmas, us, vs, torch.cuda.current_stream(outs[0].device)))

    def take_prefetched_weights(self):
        """(W_sn list, sigma list, u copies, v copies) of the oldest prefetched set, or None.  A calling stream other than the one
        the prefetch was issued on is made to wait for that stream."""
        if not self._sn_queue:
            return None
        outs, sigmas, us, vs, issued_on = self._sn_queue.pop(0)
        cur = torch.cuda.current_stream(outs[0].device)
        if issued_on is not None and cur != issued_on:         # (None: the replay loop's events order the streams)
            cur.wait_stream(issued_on)
        return outs, sigmas, us, vs

    def geo_encoding(self, x, L):
        """[B,3,h,w] -> [B,6L,h,w], channel c 2L + s L + l = sin / cos (2^l pi x_c) (reference positional_encoding :145-168 with
        reshape=True), times the coarse-to-fine weight of band l when gan.geo_c2f is set."""
        B, C, h, w = x.shape
        freq = (2 ** torch.arange(L, dtype=torch.float32, device=x.device)) * math.pi
        spec = x[:, :, None] * freq.view(1, 1, L, 1, 1)                       # [B,C,L,h,w]
        enc = torch.stack([spec.sin(), spec.cos()], dim=2)                     # [B,C,2,L,h,w]
        if self.c2f_range is not None:
            lo, hi = self.c2f_range
            alpha = (self.progress.detach() - lo) / (hi - lo) * L
            k = torch.arange(L, dtype=torch.float32, device=x.device)
            enc = enc * ((1 - (alpha - k).clamp(min=0, max=1).mul(math.pi).cos()) / 2).view(1, 1, 1, L, 1, 1)
        return enc.reshape(B, 2 * C * L, h, w)

    def forward(self, opt, x, scale=None):
        if self.L_nocs or self.L_normal:
            _image, nocs, normal = x.split(3, dim=1)
            parts = [x]
            if self.L_nocs is not None:
                parts.append(self.geo_encoding(nocs, self.L_nocs))
            if self.L_normal is not None:
                parts.append(self.geo_encoding(normal, self.L_nocs))          # (L_nocs: as the reference, :134)
            x = torch.cat(parts, dim=1)
        convs = self.sn_convs()
        pre = None
        if self._sn_queue:
            if self.training and not (torch.is_grad_enabled() and any(c.weight_orig.requires_grad for c in convs)):
                pre = self.take_prefetched_weights()
            else:
                raise RuntimeError("Discriminator.forward: prefetched spectral weights pending, but this call differentiates "
                                   "through the normalisation (or runs in eval mode)")
        weights = list(pre[0]) if pre is not None else spectral_weights(convs, self.training)   # all power iterations at once
        if self._tail_fusable(x, weights):
            # K17: the ladder's full-map convolution + the scale-conditioned head as ONE launch each way (frozen weights: the nerf
            # step's pass, whose backward needs the data gradient only)
            from . import autograd_ops
            n_main = len([m for m in self.main if isinstance(m, SNConv2d)])
            a = self._run(nn.Sequential(*list(self.main)[:-1]), x, weights)
            w0, (w1, w2, w3) = weights[0], weights[1:4]
            assert len(weights) == 4 and n_main >= 1
            return autograd_ops.disc_tail(a.flatten(1), w0.flatten(1), scale.reshape(-1), w1.flatten(1), w2.flatten(1), w3.flatten(1),
                                          self.L_scale, self.final[0].negative_slope)
        out = self._run(self.main, x, weights)                        # [B, c, 1, 1]
        if self.scale_conditional and out.is_cuda and self._plain_head():
            # K14: encoding, concatenation and the three 1x1 layers in one launch per derivative order
            from . import autograd_ops
            w1, w2, w3 = weights
            return autograd_ops.disc_head(out.flatten(1), scale.reshape(-1), w1.flatten(1), w2.flatten(1), w3.flatten(1),
                                          self.L_scale, self.final[0].negative_slope)
        if self.scale_conditional:
            spec = scale.view(-1, 1) * self.scale_freq                # [B, L]
            enc = torch.cat([spec.sin(), spec.cos()], dim=1)[:, :, None, None]
            out = self._run(self.final, torch.cat((out, enc, scale), 1), weights).flatten()
        return out

    __call__ = forward


class PerceptualLoss(nn.Module):
    """MSE between VGG19 features[:15] (conv3_3) of two images (reference layers/perceptual_loss.py)."""

    CFG = (64, 64, "M", 128, 128, "M", 256, 256, 256)

    def __init__(self, pretrained_state=None):
        super().__init__()
        layers, c = [], 3
        for v in self.CFG:
            if v == "M":
                layers.append(nn.MaxPool2d(2, 2))
            else:
                layers += [nn.Conv2d(c, v, 3, padding=1), nn.ReLU(inplace=True)]
                c = v
        self.model = nn.Sequential(*layers[:15])                      # ends at conv3_3 (no trailing ReLU)
        if pretrained_state is not None:
            self.model.load_state_dict(pretrained_state)
        for q in self.model.parameters():
            q.requires_grad = False
        self.model.eval()
        # ImageNet normalisation constants: NOT part of the state dict (the reference uses transforms.Normalize and has
        # no such keys, layers/perceptual_loss.py:19-20), otherwise reference checkpoints would not load strictly
        self.register_buffer("mean", torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1), persistent=False)
        self.register_buffer("std", torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1), persistent=False)
        # host copies of the constants (fp32-rounded, as the buffers hold them) for the fused input kernel: no device read
        self._mean_host = self.mean.flatten().tolist()
        self._std_host = self.std.flatten().tolist()

    def features(self, x):
        """``self.model(x)``; on the GPU every convolution (+ bias + ReLU) of the frozen network is one K12 launch in each
        direction (csrc/patch_conv.hip) instead of MIOpen's conv / layout transposes + bias + ReLU kernels."""
        from . import autograd_ops, ops
        mods, i = list(self.model), 0
        while i < len(mods):
            m = mods[i]
            if (isinstance(m, nn.Conv2d) and ops.conv3s1_supported(x) and not m.weight.requires_grad
                    and m.kernel_size == (3, 3) and m.stride == (1, 1) and m.padding == (1, 1)):
                relu = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
                x = autograd_ops.conv3s1_bias_relu(x, m.weight, m.bias, relu)
                i += 2 if relu else 1
            elif (isinstance(m, nn.MaxPool2d) and x.is_cuda and x.dtype == torch.float32 and m.kernel_size in (2, (2, 2))
                  and m.stride in (2, (2, 2)) and m.padding in (0, (0, 0)) and m.dilation in (1, (1, 1)) and not m.ceil_mode
                  and x.shape[-1] % 2 == 0 and x.shape[-2] % 2 == 0):
                x = autograd_ops.maxpool2(x)                       # K13: the backward writes every element (no zero fill)
                i += 1
            else:
                x = m(x)
                i += 1
        return x

    def forward(self, fake, real):
        f = self.features((fake - self.mean) / self.std)
        r = self.features((real - self.mean) / self.std)
        return F.mse_loss(f, r.detach())

    def pairs_from_patches(self, rgb, gathered, hw):
        """The two feature-loss terms of the generator step (reference model/nerf_adapt_st_gan.py:758-766) from the render
        output rgb [B,P,3] and the gathered patches [B,14,h,w]: masking, concatenation and ImageNet normalisation of the
        four image batches in one launch (K13 tp_feat_inputs), then one pass through the feature network."""
        from . import autograd_ops
        B = rgb.shape[0]
        x = autograd_ops.feat_inputs(rgb, gathered, self._mean_host, self._std_host, hw)
        feat = self.features(x)
        f1, f2, r1, r2 = torch.split(feat, B, dim=0)
        return F.mse_loss(f1, r1.detach()), F.mse_loss(f2, r2.detach())

    def loss_from_patches(self, rgb, gathered, hw, w2: float = 5.0):
        """P(fake1, real1) + w2 P(fake2, real2) of the generator step (reference model/nerf_adapt_st_gan.py:762-766) as three
        launch groups: inputs (K13 tp_feat_inputs), one pass through the feature network (K12), the two mean squared
        differences and their weighted sum (K13 tp_feat_pair_loss)."""
        from . import autograd_ops, ops
        if self.chain_eligible(rgb, gathered, hw):
            # K18: the chain written out as 17 launches, value and d / d rgb from ONE call (pools, ReLU derivatives and un-pooling in the
            # convolutions' epilogues, backward through the 2B fake images only)
            packed, bs = self._chain_params()
            loss, _parts = autograd_ops.feat_chain_loss(rgb, gathered, self._mean_host, self._std_host, hw, w2, packed, bs)
            return loss
        x = autograd_ops.feat_inputs(rgb, gathered, self._mean_host, self._std_host, hw)
        loss, _parts = autograd_ops.feat_pair_loss(self.features(x), w2)
        return loss

    def _chain_params(self):
        """(packed weights, biases) of K18.  The packed image is this module's: made on first use, re-made IN PLACE (a captured step
        keeps its address) when a weight tensor was replaced or written (tensor identity + version counter)."""
        from . import ops
        convs = [m for m in self.model if isinstance(m, nn.Conv2d)]
        ws = [c.weight for c in convs]
        # (a captured step replays the launches recorded with THIS image: weights written afterwards -- a load_state_dict of the
        # perceptual network -- are picked up by the next EAGER call or capture only; load them before the trainer captures)
        stamp = tuple((id(w), w.data_ptr(), w._version) for w in ws)
        cached = self.__dict__.get("_chain_packed")
        if cached is None or cached[1] != stamp or cached[0].device != ws[0].device:
            keep = cached[0] if cached is not None and cached[0].device == ws[0].device else None
            self.__dict__["_chain_packed"] = cached = (ops.feat_chain_pack([w.detach() for w in ws], out=keep), stamp)
        return cached[0], [c.bias for c in convs]

    def chain_eligible(self, rgb, gathered, hw):
        """tp_feat_chain covers the stock configuration: the seven frozen 3x3 convolutions of `CFG`, float32 CUDA tensors, 16 x 16
        patches; anything else (an injected network, other patch sizes, trainable weights) takes the general pieces."""
        from . import ops
        mods = list(self.model)
        convs = [m for m in mods if isinstance(m, nn.Conv2d)]
        # the hard-wired chain is VGG19.features[:15]: Conv, ReLU, Conv, ReLU, MaxPool(2), Conv, ReLU, Conv, ReLU, MaxPool(2), Conv, ReLU,
        # Conv, ReLU, Conv -- an injected network with the same convolution shapes but other activations / pooling is NOT it
        kinds = [nn.Conv2d, nn.ReLU] * 2 + [nn.MaxPool2d] + [nn.Conv2d, nn.ReLU] * 2 + [nn.MaxPool2d] + [nn.Conv2d, nn.ReLU] * 2 + [nn.Conv2d]
        if len(mods) != 15 or any(type(m) is not k for m, k in zip(mods, kinds)):
            return False
        if any(not (m.kernel_size in (2, (2, 2)) and m.stride in (2, (2, 2)) and m.padding in (0, (0, 0)) and m.dilation in (1, (1, 1))
                    and not m.ceil_mode) for m in mods if isinstance(m, nn.MaxPool2d)):
            return False
        return (ops.feat_chain_supported(rgb, gathered, hw) and len(convs) == 7
                and all(c.kernel_size == (3, 3) and c.stride == (1, 1) and c.padding == (1, 1) and c.bias is not None
                        and not c.weight.requires_grad and not c.bias.requires_grad and c.dilation == (1, 1) and c.groups == 1
                        and c.weight.is_cuda and c.weight.is_contiguous() for c in convs)
                and [tuple(c.weight.shape[:2]) for c in convs] == [(64, 3), (64, 64), (128, 64), (128, 128), (256, 128), (256, 256), (256, 256)])

    def loss_and_grad_from_patches(self, rgb, gathered, hw, w2: float = 5.0, scale: float = 1.0):
        """(loss, scale * d loss / d rgb) without autograd (the captured training step's feature graph): K18 when eligible, else None."""
        from . import ops
        if not self.chain_eligible(rgb, gathered, hw):
            return None
        packed, bs = self._chain_params()
        loss3, g_rgb = ops.feat_chain(rgb.detach(), gathered, packed, bs, self._mean_host, self._std_host, hw, w2, scale)
        return loss3[0], g_rgb

    def pairs(self, *fake_real):
        """[MSE(feat(fake_i), feat(real_i))] for several (fake, real) pairs through ONE pass over the feature network
        (the two terms of the reference's feature loss are four passes, :763-766): the same per-sample arithmetic --
        convolutions do not mix samples -- in a quarter of the launches.  Targets carry no gradient."""
        fakes, reals = [p[0] for p in fake_real], [p[1].detach() for p in fake_real]
        n = [t.shape[0] for t in fakes]
        x = torch.cat(fakes + reals, dim=0)
        feat = self.features((x - self.mean) / self.std)
        parts = torch.split(feat, n + n, dim=0)
        k = len(fakes)
        return [F.mse_loss(parts[i], parts[k + i].detach()) for i in range(k)]
