"""Minimal option container for the hot path.

The reference hands an ``easydict`` ``opt`` to every function (reference options.py:17-141); the
mirror only needs attribute access, so any attribute-dict works (easydict included).
``default_options()`` restates the values of options/nerf_lm_adapt_gan.yaml + options/base.yaml that
the ray-marching path reads; everything else of the reference's config system is out of scope.
"""
from __future__ import annotations


class AttrDict(dict):
    """dict with recursive attribute access (stand-in for easydict.EasyDict)."""

    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, AttrDict):
            v = AttrDict(v)
        super().__setitem__(k, v)

    __setattr__ = __setitem__

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def update(self, *a, **kw):
        for k, v in dict(*a, **kw).items():
            self[k] = v


def default_options(H: int = 128, W: int = 128, device: str = "cuda:0") -> AttrDict:
    return AttrDict(
        model="nerf_adapt_st_gan_amd", device=device, H=H, W=W, batch_size=8, patch_size=16, seed=0,
        c2f=dict(range=None, start=None),
        arch=dict(layers_feat=[None] + [256] * 8, layers_rgb=[None, 256, 256, 256, 3],
                  layers_trans=[None, 256, 256, 256, 5], skip=[4], posenc=dict(L_3D=10, L_view=4),
                  density_activ="softplus", tf_init=True),
        nerf=dict(view_dep=True, depth=dict(param="metric", range=[0, 3], scale=10, range_source="box"),
                  sample_intvs=64, sample_stratified=True, rand_rays=2048, density_noise_reg=None, mask_obj=True,
                  N_latent_trans=16, N_latent_light=48, min_uncert=0.05),
        data=dict(image_size=[H, W], pose_source="predicted"),
        camera=dict(model="perspective", ndc=False),
        loss_weight=dict(render=0, depth=None, mask=None, uncert=0, trans_reg=-2, feat=-2, gan_nerf=-1, lab=None,
                         gan_disc_real=0, gan_disc_fake=0, gan_reg_real=1, gan_reg_fake=None),
        gan=dict(type="standard", scale_conditional=True, geo_conditional=True, geo_c2f=None, L_nocs=None, L_scale=4,
                 L_normal=None),
        optim=dict(lr=1e-3, lr_end=1e-4, algo="Adam", sched=dict(type="ExponentialLR", gamma=0.9996163094458892)),
        optim_disc=dict(lr=1e-4, algo="RMSprop"),
        render=dict(N_candidate=3, transient="zero"),
        max_epoch=6000,
    )
