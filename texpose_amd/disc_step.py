"""K16: the discriminator step as an EXPLICIT launch schedule.

Reference: model/nerf_adapt_st_gan.py:129-171 (`disc_trainstep`: D(real) with the BCE term, the R1 penalty of
`compute_grad2` :794-807 -- a double backward through the discriminator --, D(fake) with its BCE term, one RMSprop step)
over layers/discriminator.py:45-115 (spectral-normalised stride-2 ladder with InstanceNorm + LeakyReLU, full-map
convolution, scale-conditioned head).

`trainer.GanTrainer.disc_step` used to build this step with autograd over the K7 / K9 / K11 / K14 / K15 Functions.  The
derivative structure of the step is fixed, so it is written out here as the list of kernel launches it is, with three
consequences autograd could not have:
  * a weight that receives a first-order AND a second-order (R1) contribution gets both from ONE weight-gradient launch: a
    weight gradient is a sum over samples, so the two (cotangent, input) pairs are stacked along the sample axis in buffers the
    producing kernels write into directly -- no second launch, no `add` per weight;
  * the two cotangents of a normalised activation (BCE path, R1 path) are summed inside the InstanceNorm backward launch
    (`addend`), the head's two weight-gradient shares inside the head kernel, the two spectral-norm instances of a weight
    (real pass, fake pass: the power iteration advances between them) inside `tp_sn_bwd` (`accumulate`): no engine `add`s, no
    zero fills, no AccumulateGrad;
  * nothing is computed that nobody reads (data gradients wrt the real / fake patches, the head's weight gradients in the R1
    penalty's first pass), both BCE terms and their cotangents are one launch, the R1 value and its cotangent one launch.
Same arithmetic as the autograd form up to the order of two-term sums (tests/test_gpu_parity.py::test_disc_step_schedule_*:
gradients agree to 1e-6 relative, losses bit for bit); the golden two-iteration fixture G13 pins it to the reference.

Only the HIP kernels are called (texpose_amd.ops); there is no CPU path: `eligible()` is False for CPU tensors and the caller
keeps its autograd form for those (golden generation, contract tests).
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.nn as nn

from . import knobs, ops
from .gan_modules import Discriminator, SNConv2d, _pow2_map
from .options import AttrDict


class DiscStepSchedule:
    def __init__(self, disc: Discriminator):
        self.disc = disc
        self.stages: List[tuple] = []           # (SNConv2d stride-2 4x4, InstanceNorm eps) of the ladder
        self.full: Optional[SNConv2d] = None    # the full-map convolution that ends the ladder
        self.slope: Optional[float] = None
        self.reason = self._parse()
        self._ones = {}

    # ------------------------------------------------------------------ structure
    def _parse(self) -> Optional[str]:
        d = self.disc
        if not isinstance(d, Discriminator) or not d.scale_conditional or not d._plain_head():
            return "needs the scale-conditioned three-layer head"
        if d.L_nocs or d.L_normal:
            return "geometry encodings in front of the ladder (the R1 penalty differentiates through them): autograd form"
        mods, i, slopes = list(d.main), 0, {d.final[0].negative_slope}
        while i + 2 < len(mods):
            c, n, a = mods[i], mods[i + 1], mods[i + 2]
            if not (isinstance(c, SNConv2d) and tuple(c.weight_orig.shape[-2:]) == (4, 4) and c.stride == (2, 2) and c.padding == (1, 1)
                    and isinstance(n, nn.InstanceNorm2d) and not n.affine and not n.track_running_stats and isinstance(a, nn.LeakyReLU)):
                break
            self.stages.append((c, float(n.eps)))
            slopes.add(a.negative_slope)
            i += 3
        if i != len(mods) - 1 or not self.stages:
            return "ladder is not [conv4s2, InstanceNorm, LeakyReLU]* + full-map convolution (patch_size 16 / 32)"
        full = mods[-1]
        if not (isinstance(full, SNConv2d) and full.padding == (0, 0) and full.stride == (1, 1)):
            return "last ladder layer is not an unpadded convolution"
        if len(slopes) != 1:
            return "different LeakyReLU slopes"
        self.full, self.slope = full, float(slopes.pop())
        return None

    def convs(self) -> List[SNConv2d]:
        """In the order of `Discriminator.forward`: ladder convolutions, then the head's."""
        d = self.disc
        return [m for m in list(d.main) + list(d.final) if isinstance(m, SNConv2d)]

    def eligible(self, opt, x: torch.Tensor) -> bool:
        """The schedule covers: GPU tensors, 'standard' GAN loss, a ladder whose maps the K11 kernels take."""
        if self.reason is not None or not x.is_cuda or x.dtype != torch.float32 or opt.gan.type != "standard":
            return False
        if opt.loss_weight.get("gan_disc_real") is None or opt.loss_weight.get("gan_disc_fake") is None:
            return False
        h, w = x.shape[-2:]
        for _ in self.stages:
            if not (_pow2_map(h) and _pow2_map(w)):
                return False
            h, w = h // 2, w // 2
        kh, kw = self.full.weight_orig.shape[-2:]
        k_in = self.full.weight_orig.shape[1] * kh * kw
        return (h, w) == (kh, kw) and 2 * x.shape[0] <= ops.SKINNY_MAX_ROWS and k_in >= 1024 and self.disc.training

    def _ones_like(self, t):
        key = (t.device, t.numel())
        if key not in self._ones:
            self._ones[key] = torch.ones(t.numel(), device=t.device)
        return self._ones[key]

    # ------------------------------------------------------------------ pieces
    def _normalised_weights(self):
        pre = self.disc.take_prefetched_weights()
        if pre is not None:
            return AttrDict(w=pre[0], sigma=pre[1], u=pre[2], v=pre[3])
        convs = self.convs()
        outs, sigmas, us, vs = ops.spectral_norm_fwd([c.weight_orig.detach() for c in convs], [c.weight_u for c in convs],
                                                     [c.weight_v for c in convs], True, keep_uv=True)
        return AttrDict(w=outs, sigma=sigmas, u=us, v=vs)

    def _forward(self, x, W, scale, stacks=None):
        """Ladder + head.  ``stacks[l]`` ([2B, ...], l >= 1): where the input of stage l (the full-map convolution for l = K)
        goes -- its first half."""
        K, B, sl = len(self.stages), x.shape[0], self.slope
        a, saved = x, []
        for l, (_conv, eps) in enumerate(self.stages):
            y_out = None if stacks is None else stacks[l + 1][:B]
            if ops.conv4s2_fwd_inorm_supported(a):                # convolution + InstanceNorm + LeakyReLU in one launch
                y, xhat, rstd = ops.conv4s2_fwd_inorm(a, W[l], eps, sl, y_out=y_out)
            else:
                z = ops.conv4s2_fwd(a, W[l])
                y, xhat, rstd = ops.inorm_lrelu_fwd(z, eps, sl, y_out=y_out)
            saved.append(AttrDict(x=a, xhat=xhat, rstd=rstd))
            a = y
        a2d, W0 = a.reshape(B, -1), W[K].flatten(1)
        if ops.disc_tail_eligible(a2d, W0, extra_rows=B):
            # K17: full-map convolution + head in one launch (the split-K partial sums of z meet in the last workgroup, which runs the head)
            out, t0, t1, t2 = ops.disc_tail_fwd(a2d, W0, scale, *self._head_w(W), self.disc.L_scale, sl)
            return AttrDict(out=out, stages=saved, a_full=a, head=(t0, t1, t2), C_z=W0.shape[0], tail=True)
        z3 = ops.skinny_linear_fwd(a2d, W0)
        out, t0, t1, t2 = ops.disc_head_fwd(z3, scale, W[K + 1].flatten(1), W[K + 2].flatten(1), W[K + 3].flatten(1),
                                            self.disc.L_scale, sl)
        return AttrDict(out=out, stages=saved, a_full=a, head=(t0, t1, t2), C_z=z3.shape[1], tail=False)

    def _head_w(self, W):
        K = len(self.stages)
        return W[K + 1].flatten(1), W[K + 2].flatten(1), W[K + 3].flatten(1)

    def _backward_plain(self, f, W, g_out):
        """First-order weight gradients of one pass (no R1): [ladder..., full-map, head 1..3] in the order of `convs()`."""
        K, B, sl = len(self.stages), g_out.shape[0], self.slope
        t0, t1, t2 = f.head
        gw = [None] * (K + 1)
        c_z_last = None
        if f.tail:
            # (the InstanceNorm + LeakyReLU backward of the last ladder stage rides in the same launch: no c_a round trip)
            last = f.stages[K - 1]
            r = ops.disc_tail_bwd(g_out, t0, t1, t2, W[K].flatten(1), *self._head_w(W), self.disc.L_scale, sl, a=f.a_full.reshape(B, -1),
                                  want_c_a=False, inorm=dict(xhat=last.xhat, rstd=last.rstd))
            gw[K], gW1, gW2, gW3, c_a, c_z_last = r["gW0"], r["gW1"], r["gW2"], r["gW3"], None, r["c_z"]
        else:
            c_z3, gW1, gW2, gW3, _, _ = ops.disc_head_bwd(g_out, t0, t1, t2, *self._head_w(W), f.C_z, self.disc.L_scale, sl)
            gw[K] = ops.skinny_linear_wgrad(c_z3, f.a_full.reshape(B, -1))
            c_a = ops.skinny_linear_dgrad(c_z3, W[K].flatten(1)).view_as(f.a_full)
        for l in range(K - 1, -1, -1):
            st = f.stages[l]
            c_z = c_z_last if (l == K - 1 and c_z_last is not None) else ops.inorm_lrelu_bwd(st.xhat, st.rstd, c_a, sl)
            gw[l] = ops.conv4s2_wgrad(c_z, st.x)
            if l > 0:
                c_a = ops.conv4s2_dgrad(c_z, W[l])
        return gw + [gW1, gW2, gW3]

    def _real_pass_with_r1(self, real_stack, W, scale, w_reg, make_g_out):
        """D(real), the R1 penalty (value, first and second pass) and ALL weight gradients of the real pass.  ``make_g_out(d_real)``
        returns the weighted BCE cotangent of D(real) (it may need D(fake): the caller runs the fake forward inside it)."""
        K, B, sl, L = len(self.stages), real_stack.shape[0] // 2, self.slope, self.disc.L_scale
        dev = real_stack.device
        x = real_stack[:B]
        # stacked operands of the weight gradients: xs[l] = [input of layer l | its R1 cotangent], gs[l] = [BCE-path cotangent of
        # layer l's output | that output's first-pass gradient]   (layer K = the full-map convolution, flattened)
        xs, gs, shp = [real_stack], [], x.shape
        for l, (conv, _eps) in enumerate(self.stages):
            co = conv.weight_orig.shape[0]
            shp = (B, co, shp[2] // 2, shp[3] // 2)
            xs.append(torch.empty((2 * B,) + shp[1:], device=dev))
            gs.append(torch.empty((2 * B,) + shp[1:], device=dev))
        n_out = self.full.weight_orig.shape[0]
        gs.append(torch.empty(2 * B, n_out, device=dev))
        f = self._forward(x, W, scale, stacks=xs)
        t0, t1, t2 = f.head
        Wh = self._head_w(W)
        ones = self._ones_like(f.out)
        # ---- R1, first pass: g = d D(real).sum() / d real  (reference :796-801)
        W0 = W[K].flatten(1)
        gz_last = None
        if f.tail:
            last = f.stages[K - 1]
            r = ops.disc_tail_bwd(ones, t0, t1, t2, W0, *Wh, L, sl, want_gW0=False, head_weight_grads=False, want_e=True, gz_out=gs[K][B:],
                                  inorm=dict(xhat=last.xhat, rstd=last.rstd, out=gs[K - 1][B:]))
            e1, e2, ga, ga_in, gz_last = r["e1"], r["e2"], r["c_a"].view_as(f.a_full), [None] * K, r["c_z"]
        else:
            gz3, _, _, _, e1, e2 = ops.disc_head_bwd(ones, t0, t1, t2, *Wh, f.C_z, L, sl, weight_grads=False, gz_out=gs[K][B:])
            ga, ga_in = ops.skinny_linear_dgrad(gz3, W0).view_as(f.a_full), [None] * K
        for l in range(K - 1, -1, -1):
            st = f.stages[l]
            ga_in[l] = ga
            gz = gz_last if (l == K - 1 and gz_last is not None) else ops.inorm_lrelu_bwd(st.xhat, st.rstd, ga, sl, out=gs[l][B:])
            ga = ops.conv4s2_dgrad(gz, W[l])
        # ---- value and weighted cotangent of the penalty (reference :802-806 and the .mean() of :149)
        r1, c = ops.sumsq_mean_fwd_bwd(ga, w_reg, out_g=xs[0][B:])
        # ---- R1, second pass: back through the first pass (every node linear in its cotangent)
        c_zr = [None] * K
        for l in range(K):
            st = f.stages[l]
            c_gz = ops.conv4s2_fwd(c, W[l])
            c, c_zr[l] = ops.inorm_lrelu_bwd_bwd(st.xhat, st.rstd, ga_in[l], c_gz, sl, out_gy=xs[l + 1][B:])
        gw = [None] * (K + 1)
        if f.tail:
            gW1, gW2, gW3 = ops.disc_tail_bwd_bwd(c.reshape(B, -1), ones, t0, t1, t2, e1, e2, W0, *Wh, L, sl)
            # ---- BCE path of the real pass, its cotangents joined with the R1 path's on the way: the R1 pair of the full-map weight
            # ((first-pass gz, second-pass c) = the second halves of the stacks) joins the weight-gradient sum inside the launch
            g_out = make_g_out(f.out)
            last = f.stages[K - 1]
            r = ops.disc_tail_bwd(g_out, t0, t1, t2, W0, *Wh, L, sl, a=xs[K][:B].reshape(B, -1), accumulate_into=(gW1, gW2, gW3),
                                  gy2=gs[K][B:], a2=xs[K][B:].reshape(B, -1), want_c_a=False,
                                  inorm=dict(xhat=last.xhat, rstd=last.rstd, addend=c_zr[K - 1], out=gs[K - 1][:B]))
            gw[K], gW1, gW2, gW3, c_a, c_z_last = r["gW0"], r["gW1"], r["gW2"], r["gW3"], None, r["c_z"]
        else:
            c_gz3 = ops.skinny_linear_fwd(c.reshape(B, -1), W0)
            _gg, gW1, gW2, gW3 = ops.disc_head_bwd_bwd(c_gz3, ones, t0, t1, t2, e1, e2, *Wh, L, sl)
            # ---- BCE path of the real pass, its cotangents joined with the R1 path's on the way
            g_out = make_g_out(f.out)
            c_z3, gW1, gW2, gW3, _, _ = ops.disc_head_bwd(g_out, t0, t1, t2, *Wh, f.C_z, L, sl, accumulate_into=(gW1, gW2, gW3),
                                                          gz_out=gs[K][:B])
            gw[K] = ops.skinny_linear_wgrad(gs[K], xs[K].reshape(2 * B, -1))
            c_a = ops.skinny_linear_dgrad(c_z3, W0).view_as(f.a_full)
            c_z_last = None
        for l in range(K - 1, -1, -1):
            st = f.stages[l]
            if l == K - 1 and f.tail:
                c_z = c_z_last
            else:
                c_z = ops.inorm_lrelu_bwd(st.xhat, st.rstd, c_a, sl, addend=c_zr[l], out=gs[l][:B])
            gw[l] = ops.conv4s2_wgrad(gs[l], xs[l])
            if l > 0:
                c_a = ops.conv4s2_dgrad(c_z, W[l])
        return f, r1, gw + [gW1, gW2, gW3]

    # ------------------------------------------------------------------ the nerf step's pass through the frozen discriminator
    def generator_pass(self, fake, scale, w_gan: float):
        """D(fake) with target 1 and its gradient wrt the rendered colours (reference model/nerf_adapt_st_gan.py:108-127 `nerf_trainstep`,
        :771-773 the `gan_nerf` term of compute_loss) as an explicit schedule: the weights are constants of this pass, so nothing but the
        data gradient is formed -- forward (3 launches), loss value + weighted cotangent (1), the tail's backward with the last stage's
        InstanceNorm backward inside (1), [data gradient, InstanceNorm backward]* (2 K - 1), the transposition to the render's layout (1).
        -> (bce(D(fake), 1), d (w_gan bce) / d rgb [B, P, 3], D(fake) [B]).  `fake` [B, nc, h, w] is the stack `ops.disc_inputs` built."""
        K, B, sl, L = len(self.stages), fake.shape[0], self.slope, self.disc.L_scale
        W = self._normalised_weights().w
        f = self._forward(fake, W, scale)
        if not f.tail:
            raise RuntimeError("generator_pass: the fused tail does not take this discriminator (check generator_pass_eligible)")
        out2, g_out, _ = ops.gan_disc_losses(f.out, f.out, w_gan, 0.0)         # (term 0 = target 1; the second term is not used)
        last = f.stages[K - 1]
        r = ops.disc_tail_bwd(g_out, *f.head, W[K].flatten(1), *self._head_w(W), L, sl, want_gW0=False, head_weight_grads=False,
                              want_c_a=False, inorm=dict(xhat=last.xhat, rstd=last.rstd))
        c_z = r["c_z"]
        for l in range(K - 1, -1, -1):
            if l > 0 and ops.conv4s2_dgrad_inorm_supported(c_z):        # (the InstanceNorm backward of stage l - 1 in the same launch)
                st = f.stages[l - 1]
                _, c_z = ops.conv4s2_dgrad(c_z, W[l], inorm=dict(xhat=st.xhat, rstd=st.rstd, slope=sl, keep=False))
                continue
            c_a = ops.conv4s2_dgrad(c_z, W[l])
            if l > 0:
                st = f.stages[l - 1]
                c_z = ops.inorm_lrelu_bwd(st.xhat, st.rstd, c_a, sl)
        g_rgb = ops.fake_patch_bwd(c_a, B, fake.shape[-2] * fake.shape[-1])
        return out2[0], g_rgb, f.out

    def generator_pass_eligible(self, opt, fake) -> bool:
        if knobs.K.no_gen_schedule or not self.eligible(opt, fake) or opt.loss_weight.get("gan_nerf") is None:
            return False
        return self.pairs_eligible(fake) or knobs.K.no_disc_pairs and self._tail_ok(fake)

    def _tail_ok(self, x) -> bool:
        B, _, h, w = x.shape
        for _ in self.stages:
            h, w = h // 2, w // 2
        K_in = self.full.weight_orig.shape[1] * h * w
        return K_in % 4 == 0 and B <= ops.DISC_TAIL_MAX_ROWS and not knobs.K.no_disc_tail

    # ------------------------------------------------------------------ the step with the real and the fake pass as PAIRS of launches
    # D(real) and D(fake) go through the same kernels one after the other, and so do their backward passes; every one of those launches
    # is latency-sized (10-15 us for MFLOPs), so the two problems of a pair run in ONE launch (ops.paired / tp_*_pair) in the time of
    # one: 8 launches and ~75 us off the discriminator chain of the B=4 iteration, the chain that bounds it.  Same
    # kernels on the same operands: values identical to the sequential schedule.
    def pairs_eligible(self, x) -> bool:
        if knobs.K.no_disc_pairs:
            return False
        B, _, h, w = x.shape
        for _ in self.stages:
            if (h // 2) * (w // 2) not in (16, 64) or h != w or knobs.K.no_conv_inorm:
                return False
            h, w = h // 2, w // 2
        K_in = self.full.weight_orig.shape[1] * h * w
        return K_in % 4 == 0 and B <= ops.DISC_TAIL_MAX_ROWS and not knobs.K.no_disc_tail

    def _forward_pair(self, xr, Wr, xf, Wf, scale, stacks, copies=None):
        """`_forward` of the real patches (weights Wr, inputs of the later stages into `stacks`) and of the fake patches (Wf), in pairs.
        ``copies`` = (real copy, fake copy): the first pair of launches also copies its inputs there, and everything later (the weight
        gradients of the first stage) reads the copies."""
        K, B, sl = len(self.stages), xr.shape[0], self.slope
        ar, af, sr, sf = xr, xf, [], []
        for l, (_conv, eps) in enumerate(self.stages):
            cr, cf = copies if (copies is not None and l == 0) else (None, None)
            with ops.paired():
                yr, xhr, rsr = ops.conv4s2_fwd_inorm(ar, Wr[l], eps, sl, y_out=stacks[l + 1][:B], copy_to=cr)
                yf, xhf, rsf = ops.conv4s2_fwd_inorm(af, Wf[l], eps, sl, copy_to=cf)
            sr.append(AttrDict(x=ar if cr is None else cr, xhat=xhr, rstd=rsr))
            sf.append(AttrDict(x=af if cf is None else cf, xhat=xhf, rstd=rsf))
            ar, af = yr, yf
        L = self.disc.L_scale
        with ops.paired():
            outr = ops.disc_tail_fwd(ar.reshape(B, -1), Wr[K].flatten(1), scale, *self._head_w(Wr), L, sl)
            outf = ops.disc_tail_fwd(af.reshape(B, -1), Wf[K].flatten(1), scale, *self._head_w(Wf), L, sl)
        fr = AttrDict(out=outr[0], stages=sr, a_full=ar, head=outr[1:], C_z=Wr[K].shape[0], tail=True)
        ff = AttrDict(out=outf[0], stages=sf, a_full=af, head=outf[1:], C_z=Wf[K].shape[0], tail=True)
        return fr, ff

    def _r1_passes(self, f, W, xs, gs, w_reg):
        """The R1 penalty on the real pass `f` (reference :794-807 and the .mean() of :149): first pass (d D(real).sum() / d real), value +
        weighted cotangent, second pass back through the first.  Fills the second halves of the stacks xs / gs (the R1 pairs of the weight
        gradients); returns (r1 = (value, weighted value), c_zr per stage, (gW1, gW2, gW3) of the head from the second pass)."""
        K, B, sl, L = len(self.stages), f.out.shape[0], self.slope, self.disc.L_scale
        t0, t1, t2 = f.head
        Wh = self._head_w(W)
        ones = self._ones_like(f.out)
        W0 = W[K].flatten(1)
        last = f.stages[K - 1]
        r = ops.disc_tail_bwd(ones, t0, t1, t2, W0, *Wh, L, sl, want_gW0=False, head_weight_grads=False, want_e=True, gz_out=gs[K][B:],
                              inorm=dict(xhat=last.xhat, rstd=last.rstd, out=gs[K - 1][B:]))
        e1, e2, ga, ga_in, gz_last = r["e1"], r["e2"], r["c_a"].view_as(f.a_full), [None] * K, r["c_z"]
        gz_next = None                       # (the InstanceNorm backward of stage l, formed by the data-gradient launch of stage l + 1)
        for l in range(K - 1, -1, -1):
            st = f.stages[l]
            ga_in[l] = ga
            gz = gz_last if l == K - 1 else gz_next if gz_next is not None else ops.inorm_lrelu_bwd(st.xhat, st.rstd, ga, sl, out=gs[l][B:])
            gz_next = None
            if l > 0 and ops.conv4s2_dgrad_inorm_supported(gz):
                prev = f.stages[l - 1]
                ga, gz_next = ops.conv4s2_dgrad(gz, W[l], inorm=dict(xhat=prev.xhat, rstd=prev.rstd, slope=sl, out=gs[l - 1][B:]))
            else:
                ga = ops.conv4s2_dgrad(gz, W[l])
        r1, c = ops.sumsq_mean_fwd_bwd(ga, w_reg, out_g=xs[0][B:])
        c_zr = [None] * K
        for l in range(K):
            st = f.stages[l]
            c_gz = ops.conv4s2_fwd(c, W[l])
            c, c_zr[l] = ops.inorm_lrelu_bwd_bwd(st.xhat, st.rstd, ga_in[l], c_gz, sl, out_gy=xs[l + 1][B:])
        gWh = ops.disc_tail_bwd_bwd(c.reshape(B, -1), ones, t0, t1, t2, e1, e2, W0, *Wh, L, sl)
        return r1, c_zr, gWh

    def _backward_pair(self, fr, Wr, g_real, xs, gs, c_zr, gWh, ff, Wf, g_fake):
        """All weight gradients of both passes, in pairs: the real pass' BCE path joined with its R1 path on the way (stacks xs / gs, the
        second-pass cotangents c_zr as addends, the head's second-pass gradients gWh accumulated into), the fake pass' plain backward."""
        K, B, sl, L = len(self.stages), g_real.shape[0], self.slope, self.disc.L_scale
        lr, lf = fr.stages[K - 1], ff.stages[K - 1]
        with ops.paired():
            rr = ops.disc_tail_bwd(g_real, *fr.head, Wr[K].flatten(1), *self._head_w(Wr), L, sl, a=xs[K][:B].reshape(B, -1), accumulate_into=gWh,
                                   gy2=gs[K][B:], a2=xs[K][B:].reshape(B, -1), want_c_a=False,
                                   inorm=dict(xhat=lr.xhat, rstd=lr.rstd, addend=c_zr[K - 1], out=gs[K - 1][:B]))
            rf = ops.disc_tail_bwd(g_fake, *ff.head, Wf[K].flatten(1), *self._head_w(Wf), L, sl, a=ff.a_full.reshape(B, -1), want_c_a=False,
                                   inorm=dict(xhat=lf.xhat, rstd=lf.rstd))
        gwr, gwf = [None] * (K + 1), [None] * (K + 1)
        gwr[K], gwf[K] = rr["gW0"], rf["gW0"]
        czr, czf, car, caf = rr["c_z"], rf["c_z"], None, None
        fused_next = False                   # (stage l's InstanceNorm backward came out of stage l + 1's data-gradient pair)
        for l in range(K - 1, -1, -1):
            sr, sf = fr.stages[l], ff.stages[l]
            if l < K - 1 and not fused_next:
                with ops.paired():
                    czr = ops.inorm_lrelu_bwd(sr.xhat, sr.rstd, car, sl, addend=c_zr[l], out=gs[l][:B])
                    czf = ops.inorm_lrelu_bwd(sf.xhat, sf.rstd, caf, sl)
            fused_next = False
            with ops.paired():
                gwr[l] = ops.conv4s2_wgrad(gs[l], xs[l])
                gwf[l] = ops.conv4s2_wgrad(czf, sf.x)
            if l > 0 and ops.conv4s2_dgrad_inorm_supported(czr):
                pr, pf = fr.stages[l - 1], ff.stages[l - 1]
                with ops.paired():
                    _, czr = ops.conv4s2_dgrad(czr, Wr[l], inorm=dict(xhat=pr.xhat, rstd=pr.rstd, slope=sl, addend=c_zr[l - 1], out=gs[l - 1][:B],
                                                                      keep=False))
                    _, czf = ops.conv4s2_dgrad(czf, Wf[l], inorm=dict(xhat=pf.xhat, rstd=pf.rstd, slope=sl, keep=False))
                fused_next = True
            elif l > 0:
                with ops.paired():
                    car = ops.conv4s2_dgrad(czr, Wr[l])
                    caf = ops.conv4s2_dgrad(czf, Wf[l])
        return gwr + [rr["gW1"], rr["gW2"], rr["gW3"]], gwf + [rf["gW1"], rf["gW2"], rf["gW3"]]

    def run_paired_a(self, real_stack, fake, scale, w_real, w_fake, own_inputs: bool = False):
        """First half of the paired step: both normalised weight sets, the forward pairs, both BCE terms and their cotangents -- and, with
        ``own_inputs``, private copies of the two patch stacks (one launch) so that NOTHING behind this half reads a buffer of the render
        (the captured trainer replays the two halves as two graphs and lets the next iteration's render start behind the first one).
        Returns the context `run_paired_b` continues from."""
        B, dev = fake.shape[0], fake.device
        scale = scale.reshape(-1).contiguous()
        res = AttrDict(gan_reg_real=None)
        n_real = self._normalised_weights()
        n_fake = self._normalised_weights()          # (the reference's order of power iterations: D(real)'s, then D(fake)'s)
        fake = fake.contiguous()
        copies, src_real, src_fake = None, real_stack[:B], fake
        if own_inputs:
            stack_own, fake_own = torch.empty_like(real_stack), torch.empty_like(fake)
            if knobs.K.no_conv_copy or real_stack[:B].numel() % 4 or not real_stack.is_contiguous():
                ops.step_inputs([(stack_own[:B], real_stack[:B]), (fake_own, fake)])
                src_real, src_fake = stack_own[:B], fake_own
            else:
                copies = (stack_own[:B], fake_own)         # (the first pair of convolution launches copies its inputs: no launch for it)
            real_stack, fake = stack_own, fake_own
        xs, gs, shp = [real_stack], [], real_stack[:B].shape
        for conv, _eps in self.stages:
            shp = (B, conv.weight_orig.shape[0], shp[2] // 2, shp[3] // 2)
            xs.append(torch.empty((2 * B,) + shp[1:], device=dev))
            gs.append(torch.empty((2 * B,) + shp[1:], device=dev))
        gs.append(torch.empty(2 * B, self.full.weight_orig.shape[0], device=dev))
        fr, ff = self._forward_pair(src_real, n_real.w, src_fake, n_fake.w, scale, xs, copies=copies)
        out2, g_real, g_fake = ops.gan_disc_losses(fr.out, ff.out, w_real, w_fake)
        res.gan_disc_real, res.gan_disc_fake, res.d_real, res.d_fake = out2[0], out2[1], fr.out, ff.out
        return AttrDict(res=res, n_real=n_real, n_fake=n_fake, xs=xs, gs=gs, fr=fr, ff=ff, g_real=g_real, g_fake=g_fake,
                        real=real_stack[:B], fake=fake)

    def run_paired_b(self, c, w_reg, step=None):
        """Second half: the R1 passes, the backward pairs, the spectral-norm backward; gradients into ``.grad``.  Returns the result.
        ``step`` = dict(w_real, w_fake, flags, square_avgs, steps, lr, alpha, eps): the step ENDS in the spectral-norm backward's two
        launches -- loss total + gate in the first, RMSprop of the six weights in the second (ops.spectral_norm_bwd(step=)); the
        result then carries ``total``."""
        res = c.res
        r1, c_zr, gWh = self._r1_passes(c.fr, c.n_real.w, c.xs, c.gs, w_reg)
        res.gan_reg_real, res.gan_reg_real_weighted = r1[0], r1[1]
        gw_real, gw_fake = self._backward_pair(c.fr, c.n_real.w, c.g_real, c.xs, c.gs, c_zr, gWh, c.ff, c.n_fake.w, c.g_fake)
        if step is not None:
            # (terms in the order of the autograd form's total: real, R1, fake)
            step = dict(step, terms=[res.gan_disc_real, res.gan_reg_real, res.gan_disc_fake], weights=[step["w_real"], w_reg, step["w_fake"]],
                        params=[conv.weight_orig.data for conv in self.convs()])
        grads = ops.spectral_norm_bwd(gw_real, c.n_real.w, c.n_real.u, c.n_real.v, c.n_real.sigma,
                                      second=(gw_fake, c.n_fake.w, c.n_fake.u, c.n_fake.v, c.n_fake.sigma), step=step)
        if step is not None:
            res.total = step["total"]
        for conv, g in zip(self.convs(), grads):
            conv.weight_orig.grad = g
        return res

    def _run_paired(self, real_stack, fake, scale, w_real, w_fake, w_reg, res):
        return self.run_paired_b(self.run_paired_a(real_stack, fake, scale, w_real, w_fake), w_reg)

    # ------------------------------------------------------------------ the step
    def run(self, real, fake, scale, w_real: float, w_fake: float, w_reg: Optional[float], real_stack=None):
        """Gradients of  w_real BCE(D(real), 1) + w_reg R1(real) + w_fake BCE(D(fake), 0)  wrt the discriminator's weights, written
        to ``.grad`` of the `weight_orig` parameters (replacing what was there), in the reference's order of side effects: power
        iteration + D(real), [R1], power iteration + D(fake).  ``real_stack`` ([2B, ...], optional): a buffer whose first half IS
        `real` (ops.disc_inputs(stacked=True)); without it `real` is copied into one.
        Returns AttrDict(d_real, d_fake, gan_disc_real, gan_disc_fake, gan_reg_real (unweighted, or None), gan_reg_real_weighted)."""
        B = real.shape[0]
        scale = scale.reshape(-1).contiguous()
        res = AttrDict(gan_reg_real=None)
        if w_reg is not None and self.pairs_eligible(real):
            if real_stack is None or real_stack.shape[0] != 2 * B or real_stack.data_ptr() != real.data_ptr():
                real_stack = torch.empty((2 * B,) + tuple(real.shape[1:]), device=real.device)
                real_stack[:B].copy_(real)
            return self._run_paired(real_stack, fake, scale, w_real, w_fake, w_reg, res)
        n_real = self._normalised_weights()
        state = {}

        def fake_forward_and_losses(d_real):
            # (the reference's order: the power iteration of the fake pass follows the whole real pass' forward work)
            n_fake = self._normalised_weights()
            f_fake = self._forward(fake.contiguous(), n_fake.w, scale)
            out2, g_real, g_fake = ops.gan_disc_losses(d_real, f_fake.out, w_real, w_fake)
            state.update(n_fake=n_fake, f_fake=f_fake, g_fake=g_fake)
            res.gan_disc_real, res.gan_disc_fake, res.d_fake = out2[0], out2[1], f_fake.out
            return g_real

        if w_reg is not None:
            if real_stack is None or real_stack.shape[0] != 2 * B or real_stack.data_ptr() != real.data_ptr():
                real_stack = torch.empty((2 * B,) + tuple(real.shape[1:]), device=real.device)
                real_stack[:B].copy_(real)
            f_real, r1, gw_real = self._real_pass_with_r1(real_stack, n_real.w, scale, w_reg, fake_forward_and_losses)
            res.gan_reg_real, res.gan_reg_real_weighted = r1[0], r1[1]
        else:
            f_real = self._forward(real.contiguous(), n_real.w, scale)
            gw_real = self._backward_plain(f_real, n_real.w, fake_forward_and_losses(f_real.out))
        res.d_real = f_real.out
        gw_fake = self._backward_plain(state["f_fake"], state["n_fake"].w, state["g_fake"])
        n_fake = state["n_fake"]
        # both normalised instances of the weights (real pass, fake pass) in one pair of launches: their terms are added per element
        grads = ops.spectral_norm_bwd(gw_real, n_real.w, n_real.u, n_real.v, n_real.sigma,
                                      second=(gw_fake, n_fake.w, n_fake.u, n_fake.v, n_fake.sigma))
        for conv, g in zip(self.convs(), grads):
            conv.weight_orig.grad = g
        return res
