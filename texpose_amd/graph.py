"""Graph mirror: the reference's model/nerf_adapt_st_gan.py:412-835 computation graph for the
ray-marching path, with the same method names / signatures / returned keys, on the HIP kernels.

What differs by design (documented in DESIGN.md):
  * eval rays are generated only for the requested pixels (the reference builds all H*W rays per
    2048-ray chunk and gathers, :573-579), and render_by_slices may use larger slices than
    nerf.rand_rays since results do not depend on the slice size;
  * ray-gen + bounds + stratified samples are one kernel, positions + encodings + 16 layers another,
    the composite a third; per-sample intermediates other than the 9 MLP outputs never reach HBM;
  * the NaN-retry host syncs (:554,569) are gone (the kernels produce NaN only from NaN inputs);
  * stratified jitter comes from an in-kernel Philox stream seeded from torch's seed (the reference's
    own CPU and CUDA torch.rand streams already differ); ``rand=`` injects a tensor for parity tests.
The discriminator / perceptual / Lab modules are the reference's stock PyTorch modules and are
injected, not re-implemented.
"""
from __future__ import annotations

import itertools
import warnings

import torch
import torch.nn.functional as torch_F

from . import autograd_ops, knobs, ops
from .geometry import FlexPatchSampler, RaySampler, rotation_distance
from .nerf import NeRF
from .options import AttrDict as edict

RENDER_KEYS = ("rgb", "rgb_static", "rgb_transient", "opacity", "opacity_static", "opacity_transient", "uncert",
               "depth", "alpha_static", "alpha_transient", "density")

_philox_calls = itertools.count()
_ones_cache = {}


def _ones_like_cached(t):
    """A constant tensor of ones shaped like ``t`` (made once per shape / device: no fill launch per step, capturable)."""
    key = (tuple(t.shape), t.dtype, str(t.device))
    if key not in _ones_cache:
        _ones_cache[key] = torch.ones_like(t).detach()
    return _ones_cache[key]


class Graph(torch.nn.Module):

    def __init__(self, opt, discriminator=None, perceptual_loss=None, lab_loss=None):
        super().__init__()
        self.nerf = NeRF(opt)
        if discriminator is not None:
            self.discriminator = discriminator
        if perceptual_loss is not None:
            self.perceptual_loss = perceptual_loss
        self.lab_loss = lab_loss
        self.ray_sampler = RaySampler(opt)
        # the reference passes `opt` into the random_shift slot (:424); truthy => default behaviour
        self.patch_sampler = FlexPatchSampler(True, scale_anneal=0.0002)

    def attach_latents(self, n_train, opt):
        """Per-image appearance embeddings (Model.build_networks, reference :56-59)."""
        self.latent_vars_trans = torch.nn.Embedding(n_train, opt.nerf.N_latent_trans).to(opt.device)
        self.latent_vars_light = torch.nn.Embedding(n_train, opt.nerf.N_latent_light).to(opt.device)
        torch.nn.init.normal_(self.latent_vars_trans.weight)
        torch.nn.init.normal_(self.latent_vars_light.weight)

    # ------------------------------------------------------------------ ray / sample selection
    def get_ray_idx(self, opt, var):
        # var.patch_u (optional, [3,B,1,1,1] uniforms) pins the scale / shift draw for parity tests
        # (`fuse_prologue`, set by the captured training step around its render: the coordinates are drawn by the ray-generation launch)
        defer = dict(defer=True) if getattr(self, "fuse_prologue", False) and var.get("patch_u") is None else {}
        var.ray_idx, var.ray_scales = self.patch_sampler(nbatch=opt.batch_size, patch_size=opt.patch_size,
                                                         device=opt.device, u=var.get("patch_u"), **defer)
        return var

    @staticmethod
    def get_pose(opt, var, mode=None):
        if mode == "train":
            return dict(gt=var.pose, predicted=var.pose_init)[opt.data.pose_source]
        return var.pose

    @staticmethod
    def _jitter(opt, rand, step_counter=None):
        """ops.raygen / ops.sample_depth arguments of the stratified draw.  ``step_counter`` (int64 [1] on the device, set by the
        captured training step): the Philox offset is that word, read inside the kernel -- one recorded launch, new numbers at
        every replay."""
        if rand is not None:
            return dict(rand=rand)
        if opt.nerf.sample_stratified and step_counter is not None:
            return dict(jitter=ops.JITTER_PHILOX, seed=torch.initial_seed() & (2 ** 64 - 1), offset=0, offset_dev=step_counter)
        if opt.nerf.sample_stratified:
            return dict(jitter=ops.JITTER_PHILOX, seed=torch.initial_seed() & (2 ** 64 - 1), offset=next(_philox_calls))
        return dict(jitter=ops.JITTER_MID)

    @staticmethod
    def sample_depth(opt, batch_size, depth_range, num_rays=None, rand=None):
        """(near, far) [B,R] -> stratified depths [B,R,N,1] (reference :683-700)."""
        near, far = depth_range
        return ops.sample_depth(near, far, opt.nerf.sample_intvs, depth_param=Graph._depth_param(opt), **Graph._jitter(opt, rand))[..., None]

    @staticmethod
    def _depth_param(opt):
        param = opt.nerf.depth.param
        if param not in ops.DEPTH_PARAMS:
            raise KeyError(param)                                    # (the reference indexes a dict of the two, :699)
        return param

    @staticmethod
    def ray_batch_sample(ray_identity, ray_idx):
        """[B,HW,C] rows at ray_idx [B,R] (reference :702-710)."""
        assert ray_identity.shape[0] == ray_idx.shape[0]
        C = ray_identity.shape[-1]
        return torch.gather(ray_identity, 1, ray_idx[..., None].expand(-1, -1, C))

    # ------------------------------------------------------------------ rendering (the hot path)
    def render(self, opt, pose, intr=None, ray_idx=None, depth_range=None, sample_idx=None, mode=None, rand=None):
        N = opt.nerf.sample_intvs
        z_near, z_far = depth_range
        batch_size = len(pose)
        src = dict(coords=ray_idx) if mode == "train" else dict(ray_idx=ray_idx)
        rows_k13 = (mode == "train" and torch.is_tensor(sample_idx) and sample_idx.dim() == 1 and sample_idx.is_cuda
                    and self.latent_vars_trans.weight.shape[0] == self.latent_vars_light.weight.shape[0])
        fused = {}
        if mode == "train" and getattr(self, "fuse_prologue", False):
            # captured training step: the patch coordinates (a deferred sampler call, `get_ray_idx`) and the latent rows ride in the
            # ray-generation launch (tp_raygen_train) -- one launch instead of three in front of the MLP forward
            job = ray_idx.__dict__.pop("_tp_sampler_job", None)
            if job is not None:
                fused["sampler"] = job
            if rows_k13:
                lat_t, lat_l, fused["rows"] = autograd_ops.latent_rows(self.latent_vars_trans.weight, self.latent_vars_light.weight, sample_idx,
                                                                       defer=True)
        elif torch.is_tensor(ray_idx) and "_tp_sampler_job" in ray_idx.__dict__:
            raise RuntimeError("render: ray_idx comes from a deferred sampler call but this render does not fill it (fuse_prologue is off)")
        # camera.ndc (reference :581-583) and nerf.depth.param (:699) are flags of the same launch
        center, ray, _, _, depth = ops.raygen(intr, pose, H=opt.H, W=opt.W, n_samples=N, z_near=z_near, z_far=z_far,
                                              ndc=bool(opt.camera.ndc), depth_param=self._depth_param(opt), **src, **fused,
                                              **self._jitter(opt, rand, getattr(self, "step_counter", None) if mode == "train" else None))
        depth_samples = depth[..., None]                                     # [B,R,N,1]
        if "rows" in fused:
            pass                                                             # (lat_t / lat_l: filled by the launch above)
        elif rows_k13:
            # K13: the rows of both tables in one launch, their dense gradients in one launch (index_select: two launches
            # forward, zeros + index_add_ per table backward; indexing's backward is a seven-launch index_put_)
            lat_t, lat_l = autograd_ops.latent_rows(self.latent_vars_trans.weight, self.latent_vars_light.weight, sample_idx)
        elif mode == "train" and torch.is_tensor(sample_idx) and sample_idx.dim() == 1:
            lat_t = self.latent_vars_trans.weight.index_select(0, sample_idx)
            lat_l = self.latent_vars_light.weight.index_select(0, sample_idx)
        elif mode == "train":
            lat_t = self.latent_vars_trans.weight[sample_idx]
            lat_l = self.latent_vars_light.weight[sample_idx]
        elif mode == "val":
            lat_t = self.latent_vars_trans.weight[0][None]
            lat_l = self.latent_vars_light.weight[0][None]
        else:
            if opt.render.transient == "zero":
                lat_t = torch.zeros(batch_size, opt.nerf.N_latent_trans, device=pose.device)
            elif opt.render.transient == "sample":
                lat_t = self.latent_vars_trans.weight[sample_idx][None]
            else:
                raise NotImplementedError
            lat_l = self.latent_vars_light.weight[sample_idx][None]
        rgb_s, density_s, uncert_s = self.nerf.forward_samples(opt, center=center, ray=ray, depth_samples=depth_samples,
                                                               latent_variable_trans=lat_t, latent_variable_light=lat_l,
                                                               mode=mode)
        # `prob` is never written (the reference computes and discards it, :599-606); with opt.render.per_sample = False
        # (not a reference option; default True keeps the reference's returned keys populated) alpha_static /
        # alpha_transient are not materialised either -- evaluate_full and validate only read per-ray maps
        per_sample = bool(opt.render.get("per_sample", True)) or torch.is_grad_enabled()
        fan = dict(fan_out={}) if mode == "train" else {}
        (rgb, rgb_static, rgb_transient, depth_map, opacity, opacity_static, opacity_transient, _prob, uncert,
         alpha_static, alpha_transient) = self.nerf.composite(opt, ray, rgb_s, density_s, depth_samples, uncert_s,
                                                              per_sample=per_sample, want_prob=False, **fan)
        ret = edict(rgb=rgb, rgb_static=rgb_static, rgb_transient=rgb_transient, opacity=opacity,
                    opacity_static=opacity_static, opacity_transient=opacity_transient, uncert=uncert,
                    depth=depth_map, alpha_static=alpha_static, alpha_transient=alpha_transient, density=density_s)
        # training: aliases of rgb / density for the feature loss, the discriminator patches and the transient regulariser --
        # same values, own cotangents (summed inside the composite backward instead of one `add` launch per extra consumer)
        ret.update(fan.get("fan_out") or {})
        return ret

    def _range_guarded(self, opt, device, render_image):
        """``render_image()`` renders one whole image with the MLP kernel currently selected.  The f16x3 kernel raises a
        device flag if an activation left the fp16 range; the flag is read (and cleared) once per image -- one host
        sync per ~0.2 s image -- and a flagged image is rendered again, in the same call, with the exact-fp32 kernel.
        The caller never has to catch and retry.  ``opt.arch.mlp_range_check = 'off'`` skips the read (the flag then
        stays set for ops.check_mlp_status)."""
        out = render_image()
        nerf = self.nerf
        if "f16x3" not in (nerf.precision, nerf.train_precision) or opt.arch.get("mlp_range_check", "sync") == "off" \
                or torch.cuda.is_current_stream_capturing():
            return out
        if ops.take_mlp_status(device) & 1:
            self.range_fallbacks = getattr(self, "range_fallbacks", 0) + 1
            if self.range_fallbacks == 1:
                warnings.warn("texpose_amd: an activation left the fp16 range of the f16x3 MLP kernel; the image was "
                              "re-rendered with the exact-fp32 kernel (arch.mlp_precision='fp32' avoids the double work)")
            keep = nerf.precision, nerf.train_precision
            nerf.precision = nerf.train_precision = "fp32"
            try:
                out = render_image()
            finally:
                nerf.precision, nerf.train_precision = keep
        return out

    @staticmethod
    def _slice_rays(opt):
        # result-invariant chunk size: at least the reference's rand_rays, by default a whole image per launch
        return int(opt.nerf.get("slice_rays") or max(opt.nerf.rand_rays, min(opt.H * opt.W, 1 << 20)))

    def render_by_slices(self, opt, pose, intr=None, depth_range=None, object_mask=None, sample_idx=None, mode=None):
        return self._range_guarded(opt, pose.device, lambda: self._render_by_slices(
            opt, pose, intr=intr, depth_range=depth_range, object_mask=object_mask, sample_idx=sample_idx, mode=mode))

    def _render_by_slices(self, opt, pose, intr=None, depth_range=None, object_mask=None, sample_idx=None, mode=None):
        HW = opt.H * opt.W
        step = self._slice_rays(opt)
        if mode == "val":
            parts = {k: [] for k in RENDER_KEYS}
            for c in range(0, HW, step):
                idx = torch.arange(c, min(c + step, HW), device=pose.device)[None]
                ret = self.render(opt, pose, intr=intr, ray_idx=idx, depth_range=depth_range, sample_idx=sample_idx,
                                  mode=mode)
                for k in RENDER_KEYS:
                    parts[k].append(ret[k])
            return edict({k: (None if v[0] is None else (v[0] if len(v) == 1 else torch.cat(v, dim=1))) for k, v in parts.items()})
        # eval: only object pixels are rendered and scattered into default-filled maps (reference :652-680; B == 1)
        dev, N = pose.device, opt.nerf.sample_intvs
        obj = (object_mask.reshape(HW) > 0).nonzero(as_tuple=True)[0]
        out = edict()
        lazy = not bool(opt.render.get("per_sample", True))
        for k in RENDER_KEYS:
            if lazy and "alpha" in k:
                out[k] = None                                   # (per_sample = False: not materialised)
            elif k == "uncert":
                out[k] = torch.full((1, HW, 1), float(opt.nerf.min_uncert), device=dev)
            elif k == "density":
                out[k] = torch.ones(1, HW, N, 2, device=dev)
            elif "rgb" in k:
                out[k] = torch.zeros(1, HW, 3, device=dev)
            elif "alpha" in k:
                out[k] = torch.ones(1, HW, N, device=dev)
            else:
                out[k] = torch.zeros(1, HW, 1, device=dev)
        for c in range(0, len(obj), step):
            idx = obj[c:c + step][None]
            ret = self.render(opt, pose, intr=intr, ray_idx=idx, depth_range=depth_range, sample_idx=sample_idx,
                              mode=mode)
            for k in RENDER_KEYS:
                if out[k] is not None:
                    out[k][:, idx[0]] = ret[k][0]
        return out

    # ------------------------------------------------------------------ consumers of render()
    def gather_patches(self, opt, var, disc_rgb=None):
        """One fused gather for everything compute_loss / sample_geometry / disc_forward sample at var.ray_idx.
        ``disc_rgb`` (the rendered colours the PatchGAN's fake stack is built from): the same launch also forms the discriminator's
        real / fake stacks (K13 tp_disc_inputs' values) -> var.disc_stacks = (real stack, fake, the rgb tensor they belong to)."""
        B = len(var.idx)
        if var.get("gathered_for") is var.ray_idx:          # same coordinates, same images: the gather is pure
            return var
        args = (var.ray_idx, var.image, var.get("image_syn", var.image),
                var.get("nocs_pred", var.image), var.get("normal_pred", var.image),
                var.obj_mask.view(B, opt.H, opt.W), var.get("mask_syn", var.obj_mask).view(B, opt.H, opt.W))
        if disc_rgb is not None and disc_rgb.is_cuda and not knobs.K.no_gather_disc:
            g, real, fake = ops.patch_gather(*args, disc_rgb=disc_rgb, disc_geo=bool(opt.gan.geo_conditional))
            var.disc_stacks = (real, fake, disc_rgb)
        else:
            g = ops.patch_gather(*args)
        var.gathered, var.gathered_for = g, var.ray_idx
        var.image_sample, var.image_syn_sample = g[:, 0:3], g[:, 3:6]
        var.nocs_sample, var.normal_sample = g[:, 6:9], g[:, 9:12]
        var.mask_sample, var.mask_syn_sample = g[:, 12:13], g[:, 13:14]
        return var

    def sample_geometry(self, opt, var, mode=None):
        """nocs / normal patches masked by the synthetic mask (reference :444-461)."""
        if mode == "train":
            if "nocs_sample" not in var:
                var = self.gather_patches(opt, var)
            return var
        B = len(var.idx)
        ms = (var.mask_syn > 0).float().view(B, 1, opt.H, opt.W)
        var.nocs_sample, var.normal_sample = var.nocs_pred * ms, var.normal_pred * ms
        return var

    @staticmethod
    def eval_light_index(opt, var):
        """Row of ``latent_vars_light`` an evaluation render uses (reference model/nerf_adapt_st_gan.py:487-494): one of the
        ``opt.render.N_candidate`` anchor (training) poses nearest in rotation to the test pose (camera.py:345-350), drawn with
        ``torch.randperm`` from the global CPU generator exactly as the reference draws it.  0-dim index tensor on the poses' device."""
        R_dist = rotation_distance(var.pose[..., :3, :3], var.pose_anchor[..., :3, :3]).unsqueeze(-1)
        cand = torch.topk(R_dist, k=int(opt.render.N_candidate), dim=0, largest=False, sorted=True)[1]
        return cand[torch.randperm(len(cand))[0]][0]

    def nerf_forward(self, opt, var, mode=None, stage=None):
        """``stage`` (the captured training step that runs as several hipGraphs, trainer.GraphedGanTrainer): "render" = everything up
        to the discriminator's patch stacks (render, gathers, K13 stacks), "consume" = the rest (feature chain, the discriminator's
        pass for the generator) on a ``var`` that went through "render"; None = both, as the reference's nerf_forward."""
        if stage != "consume":
            pose = self.get_pose(opt, var, mode=mode)
            depth_range = (var.z_near[:, :, None], var.z_far[:, :, None])
            if opt.nerf.rand_rays and mode == "train":
                # var.jitter_rand (optional, [B,R,N,1]) pins the stratified draw for parity tests
                ret = self.render(opt, pose, intr=var.intr, ray_idx=var.ray_idx, depth_range=depth_range,
                                  sample_idx=var.idx, mode=mode, rand=var.get("jitter_rand"))
            elif mode == "val":
                ret = self.render_by_slices(opt, pose, intr=var.intr, depth_range=depth_range, object_mask=var.obj_mask,
                                            sample_idx=None, mode=mode)
            else:
                light_idx = self.eval_light_index(opt, var)
                ret = self.render_by_slices(opt, pose, intr=var.intr, depth_range=depth_range, object_mask=var.obj_mask,
                                            sample_idx=light_idx, mode=mode)
            var.update(ret)
        if mode == "train" and opt.gan is not None and hasattr(self, "discriminator"):
            B, h, w, _ = var.ray_idx.shape
            if stage != "consume":
                if opt.gan.geo_conditional:
                    if var.rgb.is_cuda and var.get("gathered_for") is not var.ray_idx:
                        # (the gather of this iteration's patches also forms the PatchGAN's stacks: one launch instead of two)
                        var = self.gather_patches(opt, var, disc_rgb=var.get("rgb_disc", var.rgb))
                    var = self.sample_geometry(opt, var, mode)
                if "gathered" in var and var.rgb.is_cuda and var.get("gathered_for") is var.ray_idx:
                    # K13: real / fake stacks in one launch (fake differentiable wrt rgb); the discriminator step of the same
                    # iteration re-uses them (same values: it detaches the very same render)
                    pre = var.pop("disc_stacks", None)
                    if pre is not None and pre[2] is not var.get("rgb_disc", var.rgb):
                        pre = None
                    var.patch_real_nerf, patch_fake, var.patch_real_stack = autograd_ops.disc_patches(
                        var.get("rgb_disc", var.rgb), var.gathered, (h, w), bool(opt.gan.geo_conditional), pre=pre)
                    var.patch_fake_nerf, var.disc_patches_for = patch_fake, var.ray_idx
                else:
                    patch_fake = var.rgb.view(B, h, w, 3).permute(0, 3, 1, 2)
                    if opt.gan.geo_conditional:
                        patch_fake = torch.cat([patch_fake, var.nocs_sample, var.normal_sample], dim=1)
                    var.patch_fake_nerf = patch_fake
                if stage == "render":
                    return var
            # (the feature chain is issued BEFORE the discriminator's pass: with the opposite order the replayed hipGraph put both chains
            # on one hardware queue, one after the other -- 1.406 vs 1.331 ms per B=4 iteration on one box, profiles/r4)
            if var.get("feat_early_for") is not var.ray_idx:      # (else: a graph of its own already ran it, trainer._seg_feat)
                self._feature_loss_early(opt, var, (h, w), mode)
            # (the nerf step differentiates this pass once, wrt the patch: the fused frozen-weight kernels may serve it)
            if "gan_nerf_precomputed" not in var:       # (else: the trainer ran this pass as an explicit schedule, disc_step.generator_pass)
                with autograd_ops.first_order_only():
                    var.d_fake_nerf = self.discriminator(opt, var.patch_fake_nerf, var.ray_scales)
        return var

    def _feature_loss_early(self, opt, var, hw, mode):
        """With `feat_stream` set (the captured training step does that), the feature loss of the nerf step -- its inputs, the pass
        through the feature network, the two mean squared differences -- is enqueued on that stream BEFORE the discriminator's
        pass for the generator: the two chains of small dependent launches only share the render, so they run side by side in
        the replayed graph (and autograd runs each backward on its forward's stream).  `compute_loss` picks the value up and
        makes the calling stream wait for it."""
        fs = getattr(self, "feat_stream", None)
        if (fs is None or opt.loss_weight.feat is None or not hasattr(self, "perceptual_loss") or "gathered" not in var
                or not var.rgb.is_cuda or not hasattr(self.perceptual_loss, "loss_from_patches")):
            return
        main = torch.cuda.current_stream(var.rgb.device)
        fs.wait_stream(main)
        with torch.cuda.stream(fs):
            var.feat_early = self.perceptual_loss.loss_from_patches(var.get("rgb_feat", var.rgb), var.gathered, hw, 5.0)
        var.feat_early_for = var.ray_idx

    def disc_patch_stacks(self, opt, var):
        """(real, fake, real stack) of the discriminator step on the GPU, all detached: the stacks the nerf step of this iteration
        built (same render), else one K13 launch.  `real` is the first half of the [2B, ...] stack (texpose_amd/disc_step.py)."""
        B, h, w, _ = var.ray_idx.shape
        if var.get("disc_patches_for") is var.ray_idx and "patch_real_stack" in var:
            stack, fake = var.patch_real_stack.detach(), var.patch_fake_nerf.detach()
        else:
            stack, fake = ops.disc_inputs(var.rgb, var.gathered, (h, w), bool(opt.gan.geo_conditional), stacked=True)
        return stack[:B], fake, stack

    def disc_forward(self, opt, var, mode):
        if mode != "train":
            raise Exception("No use of discriminator in val/testing phase of NeRF!")
        B, h, w, _ = var.ray_idx.shape
        if var.get("disc_patches_for") is var.ray_idx and var.rgb.is_cuda:
            real, fake = var.patch_real_nerf.detach(), var.patch_fake_nerf.detach()        # built by the nerf step of this iteration
        elif "gathered" in var and var.rgb.is_cuda and var.get("gathered_for") is var.ray_idx:
            real, fake = ops.disc_inputs(var.rgb, var.gathered, (h, w), bool(opt.gan.geo_conditional))     # K13: one launch
        else:
            rgb = var.rgb.view(B, h, w, 3).permute(0, 3, 1, 2).contiguous()
            mask_pad = torch.logical_and(var.mask_syn_sample == 1, var.mask_sample == 0).float()
            real = (var.image_sample * var.mask_sample + rgb * mask_pad).detach()
            fake = rgb.detach()
            if opt.gan.geo_conditional:
                var = self.sample_geometry(opt, var, mode)
                real = torch.cat([real, var.nocs_sample, var.normal_sample], dim=1)
                fake = torch.cat([fake, var.nocs_sample, var.normal_sample], dim=1)
        var.patch_real, var.patch_fake = real.requires_grad_(), fake.requires_grad_()
        var.d_real_disc = self.discriminator(opt, var.patch_real, var.ray_scales)
        var.d_fake_disc = self.discriminator(opt, var.patch_fake, var.ray_scales)
        return var

    def evaluate_metrics(self, opt, var, lpips_module=None):
        """PSNR / SSIM of ``Model.evaluate_full`` (reference :340-362) for a rendered ``var`` (mode 'eval_*'): static
        render vs masked image, resized to 480x640 when the data is not the 128x128 crop.  Returns 0-dim device tensors.

        ``lpips_module``: the STOCK perceptual metric, injected (the reference builds ``lpips.LPIPS(net='alex')`` in
        ``Model.__init__`` (:31) and calls ``self.lpips_loss(rgb_map * 2 - 1, image_masked * 2 - 1).item()`` (:363-364);
        SURVEY 8 f4: "VGG / LPIPS stay stock").  When given, it is called exactly like that on the same two images
        PSNR / SSIM see -- [B,3,h,w], the static render and ``image * obj_mask``, after the reference's optional resize
        (bilinear ``align_corners=False`` for the images, nearest for the mask) -- and ``lpips`` is added to the result.  The
        AlexNet weights are not available offline, so the value itself is unpinned; the call contract is tested."""
        out_hw = None if list(opt.data.image_size) == [128, 128] else (480, 640)
        psnr, ssim, _ = ops.eval_metrics(var.rgb_static, var.image, var.obj_mask, opt.H, opt.W, out_hw=out_hw)
        out = edict(psnr=psnr, ssim=ssim)
        if lpips_module is not None:
            B = var.image.shape[0]
            rgb_map = var.rgb_static.view(B, opt.H, opt.W, 3).permute(0, 3, 1, 2)
            image, mask = var.image.view(B, 3, opt.H, opt.W), var.obj_mask.view(B, 1, opt.H, opt.W).float()
            if out_hw is not None:                  # reference :344-349
                rgb_map = torch_F.interpolate(rgb_map, size=list(out_hw), mode="bilinear", align_corners=False)
                image = torch_F.interpolate(image, size=list(out_hw), mode="bilinear", align_corners=False)
                mask = torch_F.interpolate(mask, size=list(out_hw), mode="nearest")
            with torch.no_grad():
                out.lpips = lpips_module(rgb_map * 2 - 1, (image * mask) * 2 - 1).reshape(-1).mean()
        return out

    def _warn_once(self, message):
        seen = self.__dict__.setdefault("_warned", set())
        if message not in seen:
            seen.add(message)
            warnings.warn("texpose_amd: " + message)

    @staticmethod
    def MSE_loss(pred, label, mask=None):
        loss = (pred.contiguous() - label) ** 2
        return loss.mean() if mask is None else (loss * mask).sum() / (mask.sum() + 1e-5)

    def compute_loss(self, opt, var, mode=None, train_step="nerf"):
        """Photometric / uncertainty / transient-regulariser / feature / GAN terms (reference :712-776)."""
        loss = edict()
        B = len(var.idx)
        if opt.nerf.rand_rays and mode in ["train", "test-optim"]:
            _, h, w, _ = var.ray_idx.shape
            var = self.gather_patches(opt, var)
            image, obj_mask = var.image_sample, var.mask_sample
            image_syn, mask_syn = var.image_syn_sample, var.mask_syn_sample
            rgb = var.rgb.view(B, h, w, 3).permute(0, 3, 1, 2)
            uncert = var.uncert.view(B, h, w, 1).permute(0, 3, 1, 2)
        else:
            image = var.image.view(B, 3, opt.H, opt.W)
            obj_mask = (var.obj_mask > 0).float().view(B, 1, opt.H, opt.W)
            image_syn = var.get("image_syn", var.image).view(B, 3, opt.H, opt.W)
            mask_syn = (var.get("mask_syn", var.obj_mask) > 0).float().view(B, 1, opt.H, opt.W)
            rgb = var.rgb.view(B, opt.H, opt.W, 3).permute(0, 3, 1, 2)
            uncert = var.uncert.view(B, opt.H, opt.W, 1).permute(0, 3, 1, 2)
            var.image_syn_sample, var.image_sample, var.mask_sample, var.mask_syn_sample = (image_syn, image, obj_mask,
                                                                                            mask_syn)
        lw = opt.loss_weight
        if train_step == "nerf":
            # default configuration: the three render-consuming terms and their gradients in one launch each way (K8)
            fused = ("gathered" in var and opt.nerf.mask_obj and lw.render is not None and lw.uncert is not None
                     and lw.trans_reg is not None and lw.mask is None and var.rgb.is_cuda)
            if fused:
                loss.render, loss.uncert, loss.trans_reg = autograd_ops.nerf_losses(var.rgb, var.uncert,
                                                                                     var.get("density_losses", var.density), var.gathered)
            elif var.rgb.is_cuda:
                # non-reference option combinations (a term switched off, mask_obj = False, full-image losses): the terms are
                # formed with torch element-wise ops on the render outputs -- same values, many small launches.  Said once.
                self._warn_once("compute_loss: loss options differ from the reference configuration (render / uncert / "
                                "trans_reg on, mask off, mask_obj, patch mode): the render-consuming terms run as torch ops "
                                "instead of the fused K8 launch")
            if not fused and lw.render is not None:
                if opt.nerf.mask_obj:
                    loss.render = (obj_mask * ((image - rgb) ** 2 / uncert ** 2)).sum() / (obj_mask.sum() + 1e-5)
                else:
                    loss.render = self.MSE_loss(rgb, image)
            if lw.mask is not None:
                loss.mask = self.MSE_loss(obj_mask, var.opacity[..., None])
            if not fused and lw.uncert is not None:
                loss.uncert = 5 + torch.log(var.uncert ** 2).mean() / 2
            if not fused and lw.trans_reg is not None:
                loss.trans_reg = var.density[..., -1].mean()
            if lw.feat is not None:
                if not hasattr(self, "perceptual_loss"):
                    raise RuntimeError("loss_weight.feat is set but no perceptual_loss module was injected")
                fused_feat = ("gathered" in var and var.rgb.is_cuda and hasattr(self.perceptual_loss, "pairs_from_patches")
                              and opt.nerf.rand_rays and mode in ["train", "test-optim"])
                if fused_feat and var.get("feat_early_for") is var.ray_idx:
                    if not var.get("feat_early_joined"):
                        torch.cuda.current_stream(var.rgb.device).wait_stream(self.feat_stream)      # (enqueued by nerf_forward)
                    loss.feat, l1 = var.feat_early, None
                elif fused_feat and hasattr(self.perceptual_loss, "loss_from_patches"):
                    # K13 + K12: inputs of the four batches in one launch, one pass through the network, l1 + 5 l2 in one launch
                    loss.feat = self.perceptual_loss.loss_from_patches(var.get("rgb_feat", var.rgb), var.gathered, (h, w), 5.0)
                    l1 = None
                elif fused_feat:
                    l1, l2 = self.perceptual_loss.pairs_from_patches(var.rgb, var.gathered, (h, w))
                else:
                    mask_pad = torch.logical_and(mask_syn == 1, obj_mask == 0).float()
                    pair1 = (rgb, image * obj_mask + image_syn * mask_pad)
                    pair2 = (rgb * obj_mask + image * (1 - obj_mask), image)
                    if hasattr(self.perceptual_loss, "pairs"):        # both terms through one pass of the feature network
                        l1, l2 = self.perceptual_loss.pairs(pair1, pair2)
                    else:                                              # any injected module with the reference's call signature
                        self._warn_once("compute_loss: the injected perceptual_loss has no fused entry points; calling it "
                                        "twice like the reference (:762-764)")
                        l1, l2 = self.perceptual_loss(*pair1), self.perceptual_loss(*pair2)
                if l1 is not None:
                    loss.feat = l1 + 5 * l2
            if lw.lab is not None:
                loss.lab, var.rgb_lab, var.img_syn_lab = self.lab_loss(rgb, image_syn, mask=mask_syn)
            if opt.gan is not None and lw.gan_nerf is not None and mode == "train":
                loss.gan_nerf = (var.gan_nerf_precomputed if "gan_nerf_precomputed" in var
                                 else self.compute_gan_loss(opt, d_outs=var.d_fake_nerf, target=1))
        elif train_step == "disc":
            if lw.gan_disc_real is not None:
                loss.gan_disc_real = self.compute_gan_loss(opt, d_outs=var.d_real_disc, target=1)
            if lw.gan_disc_fake is not None:
                loss.gan_disc_fake = self.compute_gan_loss(opt, d_outs=var.d_fake_disc, target=0)
        else:
            raise NotImplementedError
        return loss

    @staticmethod
    def compute_grad2(opt, d_outs, x_in):
        """R1 penalty: squared gradient norm of D wrt its input (reference :794-807)."""
        d_outs = d_outs if isinstance(d_outs, list) else [d_outs]
        reg = 0
        for d_out in d_outs:
            g = torch.autograd.grad(outputs=d_out.sum(), inputs=x_in, create_graph=True, retain_graph=True,
                                    only_inputs=True)[0]
            reg = reg + g.pow(2).reshape(x_in.size(0), -1).sum(1)
        return reg / len(d_outs)

    @staticmethod
    def compute_grad2_mean(opt, d_out, x_in):
        """``compute_grad2(opt, d_out, x_in).mean()`` for ONE discriminator output (what the discriminator step takes,
        reference :146-149), with the glue fused on the GPU: the cotangent of d_out.sum() is a constant vector of ones (no
        reduction launch), and sum(g^2) / B with its backward is one launch each way (K13) instead of pow / sum / add / div /
        mean and their autograd chains."""
        if not (torch.is_tensor(d_out) and d_out.is_cuda):
            return Graph.compute_grad2(opt, d_out, x_in).mean()
        ones = _ones_like_cached(d_out)
        autograd_ops.SKIP_WEIGHT_GRADS = True          # this pass differentiates wrt the INPUT only
        try:
            g = torch.autograd.grad(outputs=d_out, inputs=x_in, grad_outputs=ones, create_graph=True, retain_graph=True,
                                    only_inputs=True)[0]
        finally:
            autograd_ops.SKIP_WEIGHT_GRADS = False
        return autograd_ops.sumsq_mean(g)

    @staticmethod
    def compute_gan_loss(opt, d_outs, target):
        d_outs = d_outs if isinstance(d_outs, list) else [d_outs]
        if opt.gan.type == "standard" and len(d_outs) == 1 and d_outs[0].is_cuda:
            return autograd_ops.bce_logits_mean(d_outs[0], float(target))      # K13: one launch each way
        loss = d_outs[0].new_zeros(())                       # (no host-to-device copy: the step is hipGraph-capturable)
        for d_out in d_outs:
            if opt.gan.type == "standard":
                loss = loss + torch_F.binary_cross_entropy_with_logits(d_out, torch.full_like(d_out, float(target)))
            elif opt.gan.type == "wgan":
                loss = loss + (2 * target - 1) * d_out.mean()
            else:
                raise NotImplementedError
        return loss / len(d_outs)


def summarize_loss(opt, loss, check_finite: bool = False):
    """total = sum 10^w * loss (reference model/base.py:145-157).  The reference asserts every weighted term finite
    (:153-154: one host sync per term); ``check_finite=True`` does it ONCE on the weighted sum (a non-finite term makes
    the sum non-finite) and raises FloatingPointError before anything is back-propagated.  The trainers leave it off and
    fold the check into the read they already do before the optimiser step (texpose_amd/trainer.py)."""
    assert "all" not in loss
    total = 0.
    for key in loss:
        assert key in opt.loss_weight and loss[key].shape == ()
        if opt.loss_weight[key] is not None:
            total = total + 10 ** float(opt.loss_weight[key]) * loss[key]
    if check_finite and torch.is_tensor(total) and not bool(torch.isfinite(total)):
        raise FloatingPointError("non-finite loss: " + ", ".join("%s=%g" % (k, float(v)) for k, v in loss.items()))
    loss.update(all=total)
    return loss
