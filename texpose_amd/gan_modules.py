"""Stock PyTorch modules that sit NEXT to the hot path in a training iteration: the PatchGAN
discriminator and the VGG perceptual loss.  They run on MIOpen / rocBLAS through PyTorch-ROCm; only their
inputs (the patch gather) are hand-written kernels.  Written from the reference's architecture description
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


class Discriminator(nn.Module):
    """patch [B, 3(+6 geo), p, p] (+ patch scale) -> logit [B]."""

    def __init__(self, opt, ndf: int = 64):
        super().__init__()
        g = opt.gan
        self.scale_conditional, self.geo_conditional = bool(g.scale_conditional), bool(g.geo_conditional)
        self.L_scale = g.L_scale
        if g.L_nocs or g.L_normal or g.geo_c2f is not None:
            raise NotImplementedError("geometry positional encodings are off in the reference config")
        nc = 3 + (6 if self.geo_conditional else 0)
        p = opt.patch_size
        if p not in (16, 32, 64, 128):
            raise ValueError("patch_size must be 16, 32, 64 or 128")
        self.progress = nn.Parameter(torch.tensor(0.))
        SN = nn.utils.spectral_norm
        conv = lambda i, o, k, s, pad: SN(nn.Conv2d(i, o, (k, k), (s, s), (pad, pad), bias=False))
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
        final_dim = ndf if self.scale_conditional else 1
        blocks += [conv(c_in, ndf * 8, 4, 2, 1), nn.InstanceNorm2d(ndf * 8), nn.LeakyReLU(0.2, inplace=True),
                   conv(ndf * 8, final_dim, 4, 1, 0)]
        self.main = nn.Sequential(*blocks)
        if self.scale_conditional:
            c = ndf + 2 * self.L_scale + 1
            self.final = nn.Sequential(nn.LeakyReLU(0.2), conv(c, ndf, 1, 1, 0), nn.LeakyReLU(0.2, inplace=True),
                                       conv(ndf, ndf, 1, 1, 0), nn.LeakyReLU(0.2, inplace=True), conv(ndf, 1, 1, 1, 0))

    def forward(self, opt, x, scale=None):
        out = self.main(x)                                            # [B, c, 1, 1]
        if self.scale_conditional:
            freq = (2 ** torch.arange(self.L_scale, dtype=torch.float32, device=x.device)) * math.pi
            spec = scale.view(-1, 1) * freq                           # [B, L]
            enc = torch.cat([spec.sin(), spec.cos()], dim=1)[:, :, None, None]
            out = self.final(torch.cat((out, enc, scale), 1)).flatten()
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
        self.register_buffer("mean", torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1))
        self.register_buffer("std", torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1))

    def forward(self, fake, real):
        f = self.model((fake - self.mean) / self.std)
        r = self.model((real - self.mean) / self.std)
        return F.mse_loss(f, r.detach())
