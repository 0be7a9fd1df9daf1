"""GPU box: compare the training activation record written by the exact-fp32 and the f16x3 recording forwards, block by
block (mlp_layout.h "Training record"): which slot / tile / lane half differs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import texpose_oracle as O
from texpose_amd import ops
dev = torch.device("cuda:0")
params = {k: v.to(dev) for k, v in O.make_params(7).items()}
B, R, N = 2, 16, 8
torch.manual_seed(0)
pts = (torch.rand(B, R, N, 3, device=dev) * 2 - 1)
unit = torch.nn.functional.normalize(torch.randn(B, R, 1, 3, device=dev), dim=-1).expand(B, R, N, 3).contiguous()
lt, ll = torch.randn(B, 16, device=dev), torch.randn(B, 48, device=dev)
recs, outs = {}, {}
for prec in ("fp32", "f16x3"):
    packed = ops.pack_weights(params, precision=prec)
    rgb, den, unc, saved = ops.mlp_forward(packed, lt, ll, points=pts, ray_unit=unit, save=True, precision=prec)
    recs[prec], outs[prec] = saved.clone(), (rgb, den, unc)
torch.cuda.synchronize()
for a, b, n in zip(outs["fp32"], outs["f16x3"], ("rgb", "density", "uncert")):
    print("output", n, "max abs diff", float((a - b).abs().max()))
BLK, GROUP = 8192, 7 * 8192 + 1024 + 6 * 4 * 64
S = B * R * N
ng = (S + 31) // 32
a = recs["fp32"].view(-1)[:ng * GROUP].view(ng, GROUP)
b = recs["f16x3"].view(-1)[:ng * GROUP].view(ng, GROUP)
names = ["FEAT", "T0", "T1", "T2", "R0", "R1", "R2"]
for sl in range(7):
    x, y = a[:, sl * BLK:(sl + 1) * BLK], b[:, sl * BLK:(sl + 1) * BLK]
    d = (x - y).abs()
    print("slot", names[sl], "max abs diff %.3e" % float(d.max()), "ref max %.3e" % float(x.abs().max()), "frac differing > 1e-3: %.4f" % float((d > 1e-3).float().mean()))
    if float(d.max()) > 1e-3:
        per_tile = d.view(ng, 8, 1024).amax(dim=(0, 2))
        print("   per tile (32 features each):", [round(float(v), 4) for v in per_tile])
x, y = a[:, 7 * BLK:7 * BLK + 1024], b[:, 7 * BLK:7 * BLK + 1024]
print("slot EX max abs diff %.3e" % float((x - y).abs().max()))
ma = a[:, 7 * BLK + 1024:].contiguous().view(torch.int32)
mb = b[:, 7 * BLK + 1024:].contiguous().view(torch.int32)
print("mask words differing:", int((ma != mb).sum()), "of", ma.numel())
