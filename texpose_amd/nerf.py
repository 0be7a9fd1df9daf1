"""NeRF (static + transient + light) module mirroring reference layers/nerf_static_transient_light.py.

Same constructor, parameter names / shapes (state-dict compatible: ``mlp_feat.{0..7}``,
``mlp_rgb.{0..3}``, ``mlp_trans.{0..3}``, ``progress``) and method signatures; the arithmetic runs in
the fused HIP kernels (csrc/mlp_fwd.hip, mlp_bwd.hip, composite.hip).  Only the reference's default
architecture is compiled into the kernels; anything else raises instead of silently falling back.
"""
from __future__ import annotations

import math
from typing import Optional

import torch

from . import autograd_ops, knobs, ops


def _layer_dims(layers):
    return list(zip(layers[:-1], layers[1:]))


class NeRF(torch.nn.Module):
    WIDTH, L_3D, L_VIEW, N_TRANS, N_LIGHT = 256, 10, 4, 16, 48

    def __init__(self, opt):
        super().__init__()
        self._check_arch(opt)
        d3 = 3 + 6 * opt.arch.posenc.L_3D
        dv = 3 + 6 * opt.arch.posenc.L_view
        feat_dim = opt.arch.layers_feat[-1]

        def stack(layers, first_in, skip=(), extra_out_last=0, last_init="all"):
            mods = torch.nn.ModuleList()
            dims = _layer_dims(layers)
            for li, (k_in, k_out) in enumerate(dims):
                if li == 0:
                    k_in = first_in
                if li in skip:
                    k_in += d3
                last = li == len(dims) - 1
                lin = torch.nn.Linear(k_in, k_out + (extra_out_last if last else 0))
                if opt.arch.tf_init:
                    self.tensorflow_init_weights(lin, out=last_init if last else None)
                mods.append(lin)
            return mods

        # frozen geometry trunk: 63 -> 256 x8 (+63 skip at layer 4) -> 1 density + 256 feature
        self.mlp_feat = stack(opt.arch.layers_feat, d3, skip=opt.arch.skip, extra_out_last=1, last_init="first")
        for q in self.mlp_feat.parameters():
            q.requires_grad = False
        # static colour head: [feat, view enc, x, light latent] -> 3
        self.mlp_rgb = stack(opt.arch.layers_rgb, feat_dim + dv + 3 + opt.nerf.N_latent_light)
        # transient head: [feat, transient latent] -> rgb(3), sigma, uncertainty
        self.mlp_trans = stack(opt.arch.layers_trans, feat_dim + opt.nerf.N_latent_trans)
        if opt.c2f is not None:
            self.progress = torch.nn.Parameter(torch.tensor(0.))
        # forward arithmetic: "f16x3" (split-fp16 products on the f16 matrix cores, fp32-grade accuracy) or "fp32"
        # (exact fp32 MFMA).  `precision` is used without autograd, `train_precision` for the recording forward of a
        # training step (the backward kernels are fp32 MFMA either way and consume the same activation record)
        self.precision = opt.arch.get("mlp_precision", "f16x3")
        self.train_precision = opt.arch.get("mlp_train_precision", "f16x3")
        # compute units the weight-gradient launch of the backward fills (tp_mlp_bwd_args.wgrad_cus; 0 = all): a training step that runs
        # other streams beside the backward sets it (trainer.GraphedGanTrainer, linear graphs)
        self.wgrad_cus = 0
        # coarse-to-fine weights of the two positional encodings (reference :217-234; None: off, as in the shipped yaml).  The kernels
        # encode unweighted; the weight of band l is folded into the COLUMNS of the three weight matrices that read the encodings when
        # the stream is packed (`_state_for_pack`): W (w (.) enc) = (W (.) w^T) enc
        c = opt.c2f
        self._c2f = None if (c is None or c.range is None) else (float(c.range[0]), float(c.range[1]), 0 if c.start is None else int(c.start))
        self.density_noise_override = None       # tests: the [B,R,N] standard-normal draw of the next train-mode forward
        if self.precision not in ops.PRECISIONS or self.train_precision not in ops.PRECISIONS:
            raise ValueError("arch.mlp_precision / mlp_train_precision must be one of %s" % list(ops.PRECISIONS))
        self._packed = {}
        self._versions = {}
        self._packed_t, self._packed_t_ver = None, None     # transposed f16x3 head image of the backward (packed_weights)

    # ------------------------------------------------------------------ construction helpers
    @classmethod
    def _check_arch(cls, opt):
        a = opt.arch
        ok = (list(a.layers_feat[1:]) == [cls.WIDTH] * 8 and list(a.layers_rgb[1:]) == [cls.WIDTH] * 3 + [3]
              and list(a.layers_trans[1:]) == [cls.WIDTH] * 3 + [5] and list(a.skip) == [4]
              and a.posenc and a.posenc.L_3D == cls.L_3D and a.posenc.L_view == cls.L_VIEW
              and a.density_activ == "softplus" and opt.nerf.view_dep
              and opt.nerf.N_latent_trans == cls.N_TRANS and opt.nerf.N_latent_light == cls.N_LIGHT)
        if not ok:
            raise NotImplementedError("the gfx950 MLP kernels are built for the reference default architecture "
                                      "(options/nerf_lm_adapt_gan.yaml:9-17,37-40); got " + repr(dict(a)))

    @staticmethod
    def tensorflow_init_weights(linear, out=None):
        """Xavier-uniform as in the reference (layers/...light.py:63-74): ReLU gain for hidden layers, gain 1
        for output layers ("all") and for the density row of the trunk's last layer ("first")."""
        g = torch.nn.init.calculate_gain("relu")
        if out == "all":
            torch.nn.init.xavier_uniform_(linear.weight)
        elif out == "first":
            torch.nn.init.xavier_uniform_(linear.weight[:1])
            torch.nn.init.xavier_uniform_(linear.weight[1:], gain=g)
        else:
            torch.nn.init.xavier_uniform_(linear.weight, gain=g)
        torch.nn.init.zeros_(linear.bias)

    # ------------------------------------------------------------------ packed weight image
    def _state(self):
        return {k: v for k, v in self.named_parameters() if k.startswith("mlp_")}

    def c2f_weight(self, L: int):
        """Weight of frequency band l < L at the current ``progress`` (reference :225-231), a device tensor [L]; None: c2f off."""
        if self._c2f is None:
            return None
        lo, hi, start = self._c2f
        dev = self.progress.device
        alpha = (self.progress.detach() - lo) / (hi - lo) * L
        k = torch.arange(L, dtype=torch.float32, device=dev) - start
        return (1 - (alpha - k).clamp(min=0, max=1).mul(math.pi).cos()) / 2

    # columns of the encodings in the three matrices that read them ([x | PE(x)] = 3 + 60; mlp_rgb.0: feat 256 | view 3 | PE(view) 24 | x 3 | light 48)
    _VIEW_ENC_COLS = (WIDTH + 3, WIDTH + 3 + 6 * L_VIEW)

    def _state_for_pack(self):
        """`_state()` with the c2f weights folded into the encoding columns of mlp_feat.0 / mlp_feat.4 / mlp_rgb.0 (copies; the
        parameters keep the reference's meaning)."""
        st = self._state()
        if self._c2f is None:
            return st
        w3, wv = self.c2f_weight(self.L_3D), self.c2f_weight(self.L_VIEW)
        one = lambda n: torch.ones(n, device=w3.device)
        s3 = torch.cat([one(3), w3.repeat(6)])                       # (index c 2L + s L + l: the band is the fastest index)
        st = dict(st)
        with torch.no_grad():
            st["mlp_feat.0.weight"] = st["mlp_feat.0.weight"].detach() * s3
            st["mlp_feat.4.weight"] = st["mlp_feat.4.weight"].detach() * torch.cat([one(self.WIDTH), s3])
            st["mlp_rgb.0.weight"] = st["mlp_rgb.0.weight"].detach() * torch.cat([one(self.WIDTH + 3), wv.repeat(6), one(3 + self.N_LIGHT)])
        return st

    def c2f_grad_scale(self):
        """(index of mlp_rgb.0.weight among the head parameters, its view-encoding columns, their weights) for the backward: the kernels
        form dL/dW from the unweighted recorded encoding.  None: c2f off."""
        if self._c2f is None:
            return None
        names = [k for k, _ in self.head_parameters()]
        return names.index("mlp_rgb.0.weight"), self._VIEW_ENC_COLS, self.c2f_weight(self.L_VIEW).repeat(6)

    def set_progress(self, value: float):
        """``progress.data.fill_(value)`` (what the reference's pretrain stages do, model/nerf_pretrain.py:77) AND a re-pack of the
        weight streams at the next forward: with c2f on, the packed encoding columns depend on it, and a write through ``.data``
        does not bump the tensor version the lazy re-pack watches (a load_state_dict does).  Code that writes ``progress.data``
        itself calls `mark_heads_dirty()` afterwards."""
        with torch.no_grad():
            self.progress.fill_(float(value))
        self.mark_heads_dirty()

    def _version_keys(self):
        st = self._state()
        extra = ((self.progress.data_ptr(), self.progress._version),) if self._c2f is not None else ()
        vt = tuple((p.data_ptr(), p._version) for k, p in st.items() if k.startswith("mlp_feat")) + extra
        vh = tuple((p.data_ptr(), p._version) for k, p in st.items() if not k.startswith("mlp_feat")) + extra
        return vt, vh

    def density_noise(self, opt, mode, shape, device):
        """randn * nerf.density_noise_reg for a train-mode forward (reference :96-97), else None."""
        reg = opt.nerf.density_noise_reg
        if not reg or mode != "train":
            return None
        draw, self.density_noise_override = self.density_noise_override, None
        if draw is None:
            draw = torch.randn(shape, device=device)
        return draw.reshape(shape) * float(reg)

    def head_parameters(self):
        """(name, parameter) of the trainable heads, in a fixed order shared by forward and backward."""
        return [(k, p) for k, p in self.named_parameters() if k.startswith(("mlp_rgb", "mlp_trans"))]

    def packed_weights(self, precision: str = "fp32", for_training: bool = False, ray_bias: bool = False) -> torch.Tensor:
        """MFMA-ordered weight stream for ``precision``, re-packed lazily: trunk once (frozen), heads when an
        optimiser step or a load_state_dict bumped a parameter version.  ``for_training`` (a recording f16x3 forward whose
        backward will run): the same launch also writes the transposed head image of that backward.  ``ray_bias``: the f16x3 stream
        variant of ops.mlp_forward(..., ray_bias=True), kept beside the plain one."""
        assert not (ray_bias and (for_training or precision != "f16x3"))
        key = precision + ("+ray_bias" if ray_bias else "")
        vt, vh = self._version_keys()
        dev = next(self.parameters()).device
        buf = self._packed.get(key)
        if buf is None or buf.device != dev:
            buf = self._packed[key] = torch.empty(ops.packed_bytes() // 4, device=dev)
            self._versions[key] = [None, None]
        ver = self._versions[key]
        with torch.no_grad():
            st = self._state_for_pack() if (ver[0] != vt or ver[1] != vh) else None
            if ver[0] != vt:
                ops.pack_weights(st, packed=buf, parts=ops.PACK_TRUNK, precision=precision, ray_bias=ray_bias)
                ver[0] = vt
            if ver[1] != vh:
                if precision == "f16x3" and for_training and not knobs.K.no_pack_merge:
                    # a training step: forward chunks, biases AND the transposed image of this step's data gradient in one launch
                    if self._packed_t is None or self._packed_t.device != dev:
                        self._packed_t = torch.empty(ops.packed_t_bytes() // 4, device=dev)
                    ops.pack_heads_train(st, buf, self._packed_t)
                    self._packed_t_ver = vh
                else:
                    ops.pack_weights(st, packed=buf, parts=ops.PACK_HEADS, precision=precision, ray_bias=ray_bias)
                ver[1] = vh
        return buf

    def packed_t_current(self):
        """The transposed f16x3 head image if it was built from the head weights as they are now (by the pack launch of this
        step's forward), else None."""
        if self._packed_t is None or self._packed_t_ver is None:
            return None
        vh = self._version_keys()[1]
        return self._packed_t if vh == self._packed_t_ver else None

    def mark_heads_dirty(self):
        """Force a re-pack of the head weights at the next forward.  Needed after parameter updates that do not bump
        tensor versions: a replayed hipGraph step, and torch's FUSED optimisers (``Adam(fused=True)`` leaves
        ``p._version`` unchanged; the default foreach / single-tensor implementations bump it).  The trainers of
        texpose_amd.trainer call this after every optimiser step."""
        for ver in self._versions.values():
            ver[1] = None
            if self._c2f is not None:
                ver[0] = None               # (coarse-to-fine weights folded into the trunk's encoding columns: they follow `progress`)
        self._packed_t_ver = None

    # ------------------------------------------------------------------ reference API
    def forward(self, opt, points_3D, ray_unit=None, latent_variable_trans=None, latent_variable_light=None,
                mode=None):
        """points_3D, ray_unit [B,R,N,3] -> rgb [B,R,N,3,2], density [B,R,N,2], uncert [B,R,N,1]
        (reference layers/...light.py:76-145)."""
        assert ray_unit is not None, "view_dep=True needs ray_unit"
        lt, ll = self._per_image(latent_variable_trans, latent_variable_light, points_3D.shape[0])
        return autograd_ops.mlp(self, lt, ll, points=points_3D, ray_unit=ray_unit,
                                density_noise=self.density_noise(opt, mode, tuple(points_3D.shape[:3]), points_3D.device))

    def forward_samples(self, opt, center, ray, depth_samples, latent_variable_trans=None,
                        latent_variable_light=None, mode=None):
        """center, ray [B,R,3], depth_samples [B,R,N,1]: points and unit view directions are formed inside the
        kernel (reference layers/...light.py:147-166)."""
        lt, ll = self._per_image(latent_variable_trans, latent_variable_light, center.shape[0])
        return autograd_ops.mlp(self, lt, ll, center=center, ray=ray, depth=depth_samples,
                                density_noise=self.density_noise(opt, mode, tuple(depth_samples.shape[:3]), center.device))

    @staticmethod
    def _per_image(lat_trans, lat_light, B):
        """The reference broadcasts a single latent row over the batch (val mode passes weight[0][None], :594-596)."""
        if lat_trans.shape[0] == 1 and B > 1:
            lat_trans = lat_trans.expand(B, -1)
        if lat_light.shape[0] == 1 and B > 1:
            lat_light = lat_light.expand(B, -1)
        return lat_trans, lat_light

    @staticmethod
    def composite(opt, ray, rgb_samples, density_samples, depth_samples, uncert_samples=None, per_sample=True,
                  want_prob=True, fan_out=None):
        """11-tuple of the reference (layers/...light.py:168-212): rgb, rgb_static, rgb_transient, depth, opacity,
        opacity_static, opacity_transient, prob [B,R,N,1], uncert, alpha_static, alpha_transient [B,R,N].
        ``want_prob`` / ``per_sample`` = False (not in the reference signature) leave prob / the two alphas unwritten and
        return None in their place: Graph.render never uses ``prob`` (reference :599-606 discards it as well).
        ``fan_out`` (a dict, not in the reference signature): filled with aliases ``rgb_feat`` / ``rgb_disc`` of the rgb output and
        ``density_losses`` of the density input for the second / third consumer of a training step (autograd_ops._Composite)."""
        want_fan = fan_out is not None and torch.is_grad_enabled() and rgb_samples.requires_grad
        out, a_s, a_t, prob, rgb_ray, unc_ray, rgb_b, rgb_c, den_b = autograd_ops.composite(
            ray, rgb_samples, density_samples, depth_samples, uncert_samples, opt.nerf.min_uncert, per_sample, want_prob, want_fan)
        if want_fan:
            fan_out.update(rgb_feat=rgb_b, rgb_disc=rgb_c, density_losses=den_b)
        f = {name: out[..., lo:hi] for name, lo, hi in ops.COMPOSITE_RAY_FIELDS}
        # rgb / uncert: the compact tensors the kernel wrote next to `out` (same values; contiguous, own cotangents)
        return (rgb_ray, f["rgb_static"], f["rgb_transient"], f["depth"], f["opacity"], f["opacity_static"],
                f["opacity_transient"], None if prob is None else prob[..., None], unc_ray, a_s, a_t)

    def positional_encoding(self, opt, x, L, c2f=False):
        """[..., C] -> [..., 2 C L], index c*2L + s*L + l (reference layers/...light.py:217-234); ``c2f``: with the coarse-to-fine weights
        when opt.c2f.range is set."""
        enc = ops.posenc(x, L)
        w = self.c2f_weight(L) if c2f else None
        return enc if w is None else (enc.view(-1, L) * w).view(enc.shape)
