"""torch.autograd glue between the module API and the HIP kernels.

Gradient topology follows the reference: the trunk runs under no_grad
(layers/nerf_static_transient_light.py:87-100), so only mlp_rgb / mlp_trans parameters and the two
latent rows receive gradients; sample positions and view directions come from the no_grad ray
sampler and get none.
"""
from __future__ import annotations

import torch

from . import ops

# The R1 penalty first differentiates D wrt its INPUT with create_graph=True (Graph.compute_grad2); PyTorch's
# ctx.needs_input_grad only reflects requires_grad, so the weight-gradient kernels of every convolution would run in that pass
# although nothing consumes them (the penalty's graph hangs off the data gradients alone).  Graph.compute_grad2_mean sets this
# flag around that call; the weight gradients proper come from the backward traversal of the optimiser step.
SKIP_WEIGHT_GRADS = False


class _Composite(torch.autograd.Function):
    """out [..,14], alpha_static, alpha_transient, prob, rgb_ray [..,3], uncert_ray [..,1] (+ fan-out aliases).  rgb_ray /
    uncert_ray are compact copies of columns 0..2 / 13 of `out` written by the same launch: a training step consumes ONLY those
    two, as contiguous tensors with their own cotangents -- slices of `out` cost a copy per consumer and a zero-fill + copy +
    add per slice backward.
    ``fan_out``: three more outputs -- two aliases of rgb_ray and one of the density input.  A tensor with several consumers
    costs autograd one `add` launch per extra consumer; a consumer that reads an alias instead sends its cotangent to THIS
    node, and the composite backward sums all of them while it loads them (tp_composite_bwd_args.g_rgb_ray2/3, g_density_add)."""

    @staticmethod
    def forward(ctx, ray, rgb, density, depth, uncert, min_uncert, per_sample, want_prob, fan_out):
        out, a_s, a_t, prob, rgb_ray, unc_ray = ops.composite_fwd(ray, rgb, density, depth, uncert, min_uncert,
                                                                  per_sample=per_sample, want_prob=want_prob, compact=True)
        ctx.save_for_backward(ray, rgb, density, depth, uncert)
        ctx.min_uncert = min_uncert
        ctx.set_materialize_grads(False)
        if not fan_out:
            return out, a_s, a_t, prob, rgb_ray, unc_ray, None, None, None
        return out, a_s, a_t, prob, rgb_ray, unc_ray, rgb_ray.view_as(rgb_ray), rgb_ray.view_as(rgb_ray), density.view_as(density)

    @staticmethod
    def backward(ctx, g_out, g_as, g_at, g_prob, g_rgb_ray, g_unc_ray, g_rgb_ray2, g_rgb_ray3, g_density_alias):
        ray, rgb, density, depth, uncert = ctx.saved_tensors
        g_rgb, g_den, g_unc = ops.composite_bwd(ray, rgb, density, depth, uncert, g_out, g_as, g_at, g_prob, ctx.min_uncert,
                                                g_rgb_ray=g_rgb_ray, g_uncert_ray=g_unc_ray, g_rgb_ray2=g_rgb_ray2,
                                                g_rgb_ray3=g_rgb_ray3, g_density_add=g_density_alias)
        return None, g_rgb, g_den, None, g_unc.view_as(uncert), None, None, None, None  # (ray, rgb, density, depth, uncert, ...)


def composite(ray, rgb, density, depth, uncert, min_uncert, per_sample=True, want_prob=True, fan_out=False):
    """``per_sample`` / ``want_prob`` = False skip writing alpha_static / alpha_transient / prob ([B,R,N] each; the
    per-ray sums do not need them): the returned entries are then None.
    -> (out [..,14], alpha_static, alpha_transient, prob, rgb_ray [..,3], uncert_ray [..,1], rgb_ray alias, rgb_ray alias,
        density alias)   (the aliases None unless ``fan_out``)"""
    return _Composite.apply(ray, rgb, density, depth, uncert, float(min_uncert), bool(per_sample), bool(want_prob), bool(fan_out))


class _Mlp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, nerf, need_grad, density_noise, lat_trans, lat_light, center, ray, depth, points, ray_unit, *head_params):
        precision = nerf.train_precision if need_grad else nerf.precision
        # evaluation renders with whole tiles inside one ray: view / light / transient inputs as a per-ray bias (ops.ray_bias_applies)
        rb = ops.ray_bias_applies(precision, depth.numel() // (center.shape[0] * center.shape[1]) if center is not None else 0,
                                  bool(need_grad), center is not None)
        packed = nerf.packed_weights(precision, for_training=bool(need_grad), ray_bias=rb)   # (grad mode is off in here: the caller decided)
        res = ops.mlp_forward(packed, lat_trans, lat_light, center=center, ray=ray, depth=depth, points=points,
                              ray_unit=ray_unit, save=need_grad, precision=precision, ray_bias=rb, density_noise=density_noise)
        if need_grad:
            rgb, density, uncert, saved = res
            ctx.nerf = nerf
            ctx.grad_scale = nerf.c2f_grad_scale()       # (coarse-to-fine weights folded into the packed columns: see NeRF._state_for_pack)
            # the split-fp16 weight-gradient GEMM needs activations inside the fp16 range: guaranteed (flagged) by
            # the f16x3 recording forward only
            ctx.wgrad_precision = precision
            ctx.save_for_backward(lat_trans, lat_light, saved, rgb, density, uncert)
        else:
            rgb, density, uncert = res
        return rgb, density, uncert

    @staticmethod
    def backward(ctx, g_rgb, g_density, g_uncert):
        lat_trans, lat_light, saved, rgb, density, uncert = ctx.saved_tensors
        grads = ops.mlp_backward(ctx.nerf, lat_trans, lat_light, saved, rgb, density, uncert, g_rgb, g_density,
                                 g_uncert, wgrad_precision=ctx.wgrad_precision)
        if ctx.grad_scale is not None:
            # d / dW[:, c] of W (w (.) enc) = w_c times the gradient the kernels formed from the UNWEIGHTED recorded encoding
            idx, cols, scale = ctx.grad_scale
            grads["params"][idx][:, cols[0]:cols[1]].mul_(scale)
        return (None, None, None, grads["lat_trans"], grads["lat_light"], None, None, None, None, None) + tuple(grads["params"])


MAX_IMAGES_PER_BACKWARD = 32     # tp_mlp_bwd: the one-hot "image id" tile of the weight-gradient GEMM has 32 columns


def mlp(nerf, lat_trans, lat_light, center=None, ray=None, depth=None, points=None, ray_unit=None, density_noise=None):
    head_params = [p for _, p in nerf.head_parameters()]
    # decided here: inside Function.forward grad mode is always off and needs_input_grad ignores no_grad()
    need_grad = torch.is_grad_enabled() and (lat_trans.requires_grad or lat_light.requires_grad
                                             or any(p.requires_grad for p in head_params))
    B = lat_trans.shape[0]
    if need_grad and B > MAX_IMAGES_PER_BACKWARD:
        # more images than one backward call takes: run the recording forward / backward per group of 32 images (autograd
        # sums the head gradients of the groups)
        outs = []
        for b0 in range(0, B, MAX_IMAGES_PER_BACKWARD):
            sl = slice(b0, min(B, b0 + MAX_IMAGES_PER_BACKWARD))
            part = lambda t: None if t is None else t[sl]
            outs.append(_Mlp.apply(nerf, True, part(density_noise), lat_trans[sl], lat_light[sl], part(center), part(ray), part(depth),
                                   part(points), part(ray_unit), *head_params))
        return tuple(torch.cat(o, dim=0) for o in zip(*outs))
    return _Mlp.apply(nerf, need_grad, density_noise, lat_trans, lat_light, center, ray, depth, points, ray_unit, *head_params)


class _NerfLosses(torch.autograd.Function):
    """(render, uncert, trans_reg) of the generator step from the render outputs and the gathered patches: one forward
    and one backward launch instead of ~60 elementwise / reduction kernels (reference compute_loss :747-760)."""

    @staticmethod
    def forward(ctx, rgb, uncert, density, gathered):
        sums, losses = ops.nerf_losses_fwd(rgb, uncert, density, gathered, want_losses=True)
        ctx.save_for_backward(rgb, uncert, density, gathered, sums)
        ctx.set_materialize_grads(False)                    # (a term nobody back-propagates arrives as None, not as a zero fill)
        return losses[0], losses[1], losses[2]

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_render, g_unc, g_trans):
        rgb, uncert, density, gathered, sums = ctx.saved_tensors
        if g_render is None and g_unc is None and g_trans is None:
            return None, None, None, None
        g_rgb, g_u, g_d = ops.nerf_losses_bwd(rgb, uncert, density, gathered, sums, (g_render, g_unc, g_trans))
        return g_rgb, g_u.view_as(uncert), g_d, None


def nerf_losses(rgb, uncert, density, gathered):
    return _NerfLosses.apply(rgb, uncert, density, gathered)


class _InormLreluBackward(torch.autograd.Function):
    """gx of the fused InstanceNorm + LeakyReLU, itself differentiable (the R1 penalty back-propagates through the
    gradient wrt the discriminator input, reference model/nerf_adapt_st_gan.py:794-807).  ``x`` is only the handle
    autograd routes the second-order gradient to; the arithmetic uses xhat / rstd of the forward."""

    @staticmethod
    def forward(ctx, x, xhat, rstd, gy, slope):
        ctx.save_for_backward(xhat, rstd, gy)
        ctx.slope = slope
        return ops.inorm_lrelu_bwd(xhat, rstd, gy, slope)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, ggx):
        xhat, rstd, gy = ctx.saved_tensors
        g_gy, g_x = ops.inorm_lrelu_bwd_bwd(xhat, rstd, gy, ggx.contiguous(), ctx.slope)
        return g_x, None, None, g_gy, None


class _InormLrelu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, eps, slope):
        y, xhat, rstd = ops.inorm_lrelu_fwd(x, eps, slope)
        ctx.save_for_backward(x, xhat, rstd)
        ctx.slope = slope
        return y

    @staticmethod
    def backward(ctx, gy):
        x, xhat, rstd = ctx.saved_tensors
        return _InormLreluBackward.apply(x, xhat, rstd, gy.contiguous(), ctx.slope), None, None


def inorm_lrelu(x, eps: float = 1e-5, slope: float = 0.2):
    """LeakyReLU(InstanceNorm2d(x)) (affine = False) with first and second derivatives as single launches (K9)."""
    return _InormLrelu.apply(x.contiguous(), float(eps), float(slope))


# ---- K11: the stride-2 4x4 convolutions of the PatchGAN ladder.  A convolution is bilinear in (x, w): its three kernels
# (forward F, data gradient D, weight gradient G) are closed under differentiation, so these three Functions differentiate
# to any order (the R1 penalty needs the second) with one launch per node and no MIOpen layout transposes.
class _Conv4s2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        return ops.conv4s2_fwd(x, w)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gy = gy.contiguous()
        gx = _Conv4s2Dgrad.apply(gy, w) if ctx.needs_input_grad[0] else None
        gw = _Conv4s2Wgrad.apply(gy, x) if ctx.needs_input_grad[1] and not SKIP_WEIGHT_GRADS else None
        return gx, gw


class _Conv4s2Dgrad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gy, w):
        ctx.save_for_backward(gy, w)
        return ops.conv4s2_dgrad(gy, w)

    @staticmethod
    def backward(ctx, ggx):
        gy, w = ctx.saved_tensors
        ggx = ggx.contiguous()
        g_gy = _Conv4s2.apply(ggx, w) if ctx.needs_input_grad[0] else None
        g_w = _Conv4s2Wgrad.apply(gy, ggx) if ctx.needs_input_grad[1] else None
        return g_gy, g_w


class _Conv4s2Wgrad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gy, x):
        ctx.save_for_backward(gy, x)
        return ops.conv4s2_wgrad(gy, x)

    @staticmethod
    def backward(ctx, ggw):
        gy, x = ctx.saved_tensors
        ggw = ggw.contiguous()
        g_gy = _Conv4s2.apply(x, ggw) if ctx.needs_input_grad[0] else None
        g_x = _Conv4s2Dgrad.apply(gy, ggw) if ctx.needs_input_grad[1] else None
        return g_gy, g_x


def conv4s2(x, w):
    """conv2d(x, w, stride 2, padding 1) for a [Co,C,4,4] weight (K11), differentiable to any order."""
    return _Conv4s2.apply(x.contiguous(), w.contiguous())


# ---- K12: 3x3 convolution + bias + ReLU of the frozen perceptual-loss network: one launch forward, one backward (data
# gradient with the ReLU derivative applied while the cotangent is gathered).  No weight gradient: callers check that.
class _Conv3s1BiasRelu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, bias, relu):
        y = ops.conv3s1_fwd(x, w, bias, relu)
        ctx.relu = relu
        ctx.save_for_backward(w, y if relu else None)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gy):
        w, y = ctx.saved_tensors
        return ops.conv3s1_dgrad(gy.contiguous(), w, y if ctx.relu else None), None, None, None


def conv3s1_bias_relu(x, w, bias, relu: bool):
    """relu?(conv2d(x, w, bias, padding 1)) for a frozen [Co,C,3,3] weight (K12)."""
    return _Conv3s1BiasRelu.apply(x.contiguous(), w, bias, bool(relu))


class _MaxPool2(torch.autograd.Function):
    """MaxPool2d(2, 2) with a backward that writes every input element (torch: zero fill + scatter), first order only."""

    @staticmethod
    def forward(ctx, x):
        y, arg = ops.maxpool2_fwd(x)
        ctx.save_for_backward(arg)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gy):
        (arg,) = ctx.saved_tensors
        return ops.maxpool2_bwd(gy.contiguous(), arg)


def maxpool2(x):
    return _MaxPool2.apply(x.contiguous())


# ---- K13 pieces
class _BceLogitsMean(torch.autograd.Function):
    """mean binary_cross_entropy_with_logits(x, constant target): one launch each way (first order only: the R1 penalty
    differentiates d_out.sum(), never this loss)."""

    @staticmethod
    def forward(ctx, x, target):
        ctx.save_for_backward(x)
        ctx.target = target
        return ops.bce_logits_fwd(x, target)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        x, = ctx.saved_tensors
        return ops.bce_logits_bwd(x, ctx.target, g).view_as(x), None


def bce_logits_mean(x, target: float):
    return _BceLogitsMean.apply(x.contiguous(), float(target))


class _FeatInputs(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rgb, gathered, mean, std, hw):
        ctx.save_for_backward(rgb, gathered)
        ctx.consts = (mean, std)
        return ops.feat_inputs_fwd(rgb, gathered, mean, std, hw)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        rgb, gathered = ctx.saved_tensors
        return ops.feat_inputs_bwd(rgb, gathered, ctx.consts[0], ctx.consts[1], g.contiguous()), None, None, None, None


def feat_inputs(rgb, gathered, mean, std, hw):
    """rgb [B,P,3] + the gathered patches -> the normalised [4B,3,h,w] input of the feature network; gradient wrt rgb only."""
    return _FeatInputs.apply(rgb.contiguous(), gathered, tuple(mean), tuple(std), tuple(hw))


# ---- K18: the whole feature-loss chain (inputs, frozen network, pair loss, data gradient) as one call
class _FeatChainLoss(torch.autograd.Function):
    """loss = P(fake1, real1) + w2 P(fake2, real2) of the generator step from rgb [B,P,3] and the gathered patches; the gradient wrt rgb is
    computed together with the value (tp_feat_chain) and scaled by the incoming cotangent in backward.  First order only."""

    @staticmethod
    def forward(ctx, rgb, gathered, mean, std, hw, w2, packed_and_biases):
        packed, bs = packed_and_biases
        loss3, g_rgb = ops.feat_chain(rgb, gathered, packed, bs, mean, std, hw, w2, 1.0)
        ctx.save_for_backward(g_rgb)
        return loss3[0].clone(), loss3

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g, _g3):
        (g_rgb,) = ctx.saved_tensors
        return g_rgb * g, None, None, None, None, None, None


def feat_chain_loss(rgb, gathered, mean, std, hw, w2, packed, biases):
    """-> (loss, parts [3] = {loss, l1, l2} detached); ``packed`` = ops.feat_chain_pack of the seven weights."""
    loss, parts = _FeatChainLoss.apply(rgb.contiguous(), gathered, tuple(mean), tuple(std), tuple(hw), float(w2), (packed, list(biases)))
    return loss, parts.detach()


# ---- K14: the scale-conditioned head of the PatchGAN, one launch per derivative order
class _DiscHeadBackward(torch.autograd.Function):
    """(gz, gW1, gW2, gW3) of the head; differentiable once more for the R1 penalty, whose cotangent reaches gz only (the
    weight gradients are leaves of the step).  LeakyReLU masks are constants almost everywhere: nothing flows to t0..t2."""

    @staticmethod
    def forward(ctx, g_out, t0, t1, t2, W1, W2, W3, C_z, L, slope):
        # weight gradients only when somebody takes them (not for the frozen discriminator of the generator step, not in the R1
        # penalty's first pass): a third of this launch
        want_w = any(ctx.needs_input_grad[4:7]) and not SKIP_WEIGHT_GRADS
        gz, gW1, gW2, gW3, e1, e2 = ops.disc_head_bwd(g_out, t0, t1, t2, W1, W2, W3, C_z, L, slope, weight_grads=want_w)
        ctx.save_for_backward(g_out, t0, t1, t2, e1, e2, W1, W2, W3)
        ctx.consts = (L, slope)
        ctx.set_materialize_grads(False)
        return gz, gW1, gW2, gW3

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, c_gz, c_W1, c_W2, c_W3):
        if c_W1 is not None or c_W2 is not None or c_W3 is not None:
            raise NotImplementedError("second-order terms through the head's weight gradients")
        if c_gz is None:
            return (None,) * 10
        g_out, t0, t1, t2, e1, e2, W1, W2, W3 = ctx.saved_tensors
        gg, gW1, gW2, gW3 = ops.disc_head_bwd_bwd(c_gz.contiguous(), g_out, t0, t1, t2, e1, e2, W1, W2, W3, *ctx.consts)
        return gg, None, None, None, gW1, gW2, gW3, None, None, None


class _DiscHead(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, scale, W1, W2, W3, L, slope):
        out, t0, t1, t2 = ops.disc_head_fwd(z, scale, W1, W2, W3, L, slope)
        ctx.save_for_backward(t0, t1, t2, W1, W2, W3)
        ctx.consts = (z.shape[1], L, slope)
        return out

    @staticmethod
    def backward(ctx, g_out):
        t0, t1, t2, W1, W2, W3 = ctx.saved_tensors
        gz, gW1, gW2, gW3 = _DiscHeadBackward.apply(g_out.contiguous(), t0, t1, t2, W1, W2, W3, *ctx.consts)
        return gz, None, gW1, gW2, gW3, None, None


def disc_head(z, scale, W1, W2, W3, L: int, slope: float = 0.2):
    """out [B] = W3 lrelu(W2 lrelu(W1 lrelu([z, enc(scale), scale]))) with W1 [H,C+2L+1], W2 [H,H], W3 [1,H] (K14)."""
    return _DiscHead.apply(z.contiguous(), scale.contiguous(), W1.contiguous(), W2.contiguous(), W3.contiguous(), int(L), float(slope))


# The fused frozen-weight nodes below (_Conv4s2Inorm, _DiscTail) are first order in their input.  A frozen weight alone does not say
# that first order suffices -- a gradient penalty wrt the input with create_graph=True under frozen weights needs the differentiable
# K11 / K9 / K15 / K14 nodes -- so the caller that knows (the nerf step's pass through the discriminator, texpose_amd/graph.py)
# says so explicitly:
_FIRST_ORDER = [0]


class first_order_only:
    """``with first_order_only():`` -- passes through the discriminator issued inside need the data gradient once and nothing else."""

    def __enter__(self):
        _FIRST_ORDER[0] += 1

    def __exit__(self, *exc):
        _FIRST_ORDER[0] -= 1


def first_order_declared():
    return _FIRST_ORDER[0] > 0


# ---- K11 + K9 fused forward for FROZEN weights (the nerf step's pass through the discriminator): one launch instead of two
class _Conv4s2Inorm(torch.autograd.Function):
    """y = lrelu(instance_norm(conv4s2(x, w))) with a constant weight: forward tp_conv4s2_fwd_inorm, backward K9's first-order
    kernel followed by K11's data gradient (two launches, as the unfused nodes).  First order only."""

    @staticmethod
    def forward(ctx, x, w, eps, slope):
        y, xhat, rstd = ops.conv4s2_fwd_inorm(x, w, eps, slope)
        ctx.save_for_backward(xhat, rstd, w)
        ctx.slope = slope
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gy):
        xhat, rstd, w = ctx.saved_tensors
        gz = ops.inorm_lrelu_bwd(xhat, rstd, gy.contiguous(), ctx.slope)
        gx = ops.conv4s2_dgrad(gz, w) if ctx.needs_input_grad[0] else None
        return gx, None, None, None


def conv4s2_inorm(x, w, eps: float, slope: float):
    return _Conv4s2Inorm.apply(x.contiguous(), w.contiguous(), float(eps), float(slope))


# ---- K17: full-map convolution + head as one launch each way (frozen discriminator of the nerf step: data gradient only)
class _DiscTail(torch.autograd.Function):
    """out [B] of the PatchGAN's tail for FROZEN weights: forward tp_disc_tail_fwd, backward tp_disc_tail_bwd (c_a only).  First
    order in the input; anything else (weight gradients through autograd, the R1 double backward) takes the K15 + K14 nodes."""

    @staticmethod
    def forward(ctx, a, W0, scale, W1, W2, W3, L, slope):
        out, t0, t1, t2 = ops.disc_tail_fwd(a, W0, scale, W1, W2, W3, L, slope)
        ctx.save_for_backward(t0, t1, t2, W0, W1, W2, W3)
        ctx.consts = (L, slope)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_out):
        t0, t1, t2, W0, W1, W2, W3 = ctx.saved_tensors
        r = ops.disc_tail_bwd(g_out.contiguous(), t0, t1, t2, W0, W1, W2, W3, *ctx.consts, want_gW0=False, head_weight_grads=False)
        return r["c_a"], None, None, None, None, None, None, None


def disc_tail(a, W0, scale, W1, W2, W3, L: int, slope: float = 0.2):
    """a [B,K] (the ladder's last activation, flattened) -> logits [B]; weights are constants (see _DiscTail)."""
    return _DiscTail.apply(a.contiguous(), W0.contiguous(), scale.contiguous(), W1.contiguous(), W2.contiguous(), W3.contiguous(),
                           int(L), float(slope))


# ---- K15: y = x W^T for a handful of rows (the PatchGAN's full-map convolution), closed under differentiation like K11
class _SkinnyLinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        return ops.skinny_linear_fwd(x, w)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gy = gy.contiguous()
        gx = _SkinnyLinearDgrad.apply(gy, w) if ctx.needs_input_grad[0] else None
        gw = _SkinnyLinearWgrad.apply(gy, x) if ctx.needs_input_grad[1] and not SKIP_WEIGHT_GRADS else None
        return gx, gw


class _SkinnyLinearDgrad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gy, w):
        ctx.save_for_backward(gy, w)
        return ops.skinny_linear_dgrad(gy, w)

    @staticmethod
    def backward(ctx, ggx):
        gy, w = ctx.saved_tensors
        ggx = ggx.contiguous()
        g_gy = _SkinnyLinear.apply(ggx, w) if ctx.needs_input_grad[0] else None
        g_w = _SkinnyLinearWgrad.apply(gy, ggx) if ctx.needs_input_grad[1] else None
        return g_gy, g_w


class _SkinnyLinearWgrad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gy, x):
        ctx.save_for_backward(gy, x)
        return ops.skinny_linear_wgrad(gy, x)

    @staticmethod
    def backward(ctx, ggw):
        gy, x = ctx.saved_tensors
        ggw = ggw.contiguous()
        g_gy = _SkinnyLinear.apply(x, ggw) if ctx.needs_input_grad[0] else None
        g_x = _SkinnyLinearDgrad.apply(gy, ggw) if ctx.needs_input_grad[1] else None
        return g_gy, g_x


def skinny_linear(x, w):
    """x [M,K] @ w [N,K]^T for M <= 256 rows (K15), differentiable to any order."""
    return _SkinnyLinear.apply(x.contiguous(), w.contiguous())


# ---- K13 (round 3): the remaining small chains of the GAN step, one launch per direction
class _DiscPatches(torch.autograd.Function):
    """(real stack, fake) of the PatchGAN from the rendered colours and the gathered patches (tp_disc_inputs).  `fake` [B,nc,h,w]
    carries the gradient back to rgb (the nerf step's D(fake) term); the real stack [2B,nc,h,w] holds `real` in its first half, a
    constant of the step (the second half belongs to the discriminator step's schedule, texpose_amd/disc_step.py)."""

    @staticmethod
    def forward(ctx, rgb, gathered, hw, geo, pre_real=None, pre_fake=None):
        if pre_real is not None:                     # (formed by the gather's launch from this very rgb: ops.patch_gather(disc_rgb=))
            real, fake = pre_real, pre_fake
        else:
            real, fake = ops.disc_inputs(rgb, gathered, hw, geo, stacked=True)
        ctx.dims = (rgb.shape[0], rgb.shape[1])
        ctx.mark_non_differentiable(real)
        ctx.set_materialize_grads(False)
        return real, fake

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_real, g_fake):
        if g_fake is None:
            return None, None, None, None, None, None
        return ops.fake_patch_bwd(g_fake.contiguous(), *ctx.dims), None, None, None, None, None


def disc_patches(rgb, gathered, hw, geo: bool, pre=None):
    """-> (real [B,nc,h,w], fake [B,nc,h,w], real stack [2B,nc,h,w]).  ``pre`` = (real stack, fake, ...): already formed from this rgb."""
    pr, pf = (pre[0], pre[1]) if pre is not None else (None, None)
    stack, fake = _DiscPatches.apply(rgb, gathered, tuple(hw), bool(geo), pr, pf)
    return stack[:rgb.shape[0]], fake, stack


class _FeatPairLoss(torch.autograd.Function):
    """l1 + w2 l2 with l_i = mse(feat(fake_i), feat(real_i).detach()) for feat = features of [fake1 | fake2 | real1 | real2]
    (reference model/nerf_adapt_st_gan.py:762-766): two mse_loss + mul + add and their backward chains as one launch each."""

    @staticmethod
    def forward(ctx, feat, w2):
        out = ops.feat_pair_loss_fwd(feat, w2)
        ctx.save_for_backward(feat)
        ctx.w2 = w2
        ctx.set_materialize_grads(False)                    # (no zero fill for the cotangent of the logged parts)
        parts = out[1:]
        ctx.mark_non_differentiable(parts)
        return out[0], parts

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g, _g_parts):
        (feat,) = ctx.saved_tensors
        if g is None:
            return None, None
        return ops.feat_pair_loss_bwd(feat, ctx.w2, g.contiguous()), None


def feat_pair_loss(feat, w2: float = 5.0):
    """-> (loss, [l1, l2]) ."""
    return _FeatPairLoss.apply(feat.contiguous(), float(w2))


class _SumsqMean(torch.autograd.Function):
    """sum(g^2) / B: the R1 penalty value of a batch (compute_grad2(...).mean() with one discriminator output).  Its backward
    (2 g cot / B) feeds the double backward of the discriminator; it is itself only needed at first order."""

    @staticmethod
    def forward(ctx, g):
        ctx.save_for_backward(g)
        return ops.sumsq_mean_fwd(g)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, cot):
        (g,) = ctx.saved_tensors
        return ops.sumsq_mean_bwd(g, cot.contiguous())


def sumsq_mean(g):
    return _SumsqMean.apply(g.contiguous())


class _LatentRows(torch.autograd.Function):
    """Rows idx of the transient / light latent tables in one launch; dense table gradients in one launch (no zero fill)."""

    @staticmethod
    def forward(ctx, w_trans, w_light, idx, job_out=None):
        # the kernels read int64 rows: save what the forward actually passed down (an int32 or strided idx is accepted, like
        # index_select's, and converted ONCE here -- the backward must see the same buffer)
        idx = idx.to(torch.int64).contiguous()
        # ... a PRIVATE copy, written by the forward's own launch: the captured training step refills the caller's idx buffer for the
        # next iteration while this one's backward is still to run (trainer.GraphedGanTrainer, `defer_results`)
        own = torch.empty_like(idx)
        ctx.save_for_backward(own)
        ctx.n_rows = w_trans.shape[0]
        if job_out is not None:
            # nothing launched here: the ray-generation launch of the training step carries the gather (ops.raygen(..., rows=job))
            ot, ol, job = ops.latent_rows_fwd(w_trans, w_light, idx, idx_copy=own, defer=True)
            job_out.append(job)
            return ot, ol
        return ops.latent_rows_fwd(w_trans, w_light, idx, idx_copy=own)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_trans, g_light):
        (idx,) = ctx.saved_tensors
        gwt, gwl = ops.latent_rows_bwd(g_trans.contiguous(), g_light.contiguous(), idx, ctx.n_rows)
        return gwt, gwl, None, None


def latent_rows(w_trans, w_light, idx, defer=False):
    """``defer``: -> (rows_trans, rows_light, job); the tensors are filled by the ray-generation launch that is given the job."""
    if not defer:
        return _LatentRows.apply(w_trans, w_light, idx)
    job = []
    ot, ol = _LatentRows.apply(w_trans, w_light, idx, job)
    return ot, ol, job[0]
