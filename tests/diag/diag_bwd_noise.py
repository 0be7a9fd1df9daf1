"""Diagnostic (GPU box): is the HIP-vs-oracle gradient gap fp32 noise?  Compares HIP fp32, oracle fp32 and
oracle fp64 on the same inputs."""
import sys, os
_REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, _REPO)
sys.path.insert(0, os.path.join(_REPO, "tests"))
import numpy as np, torch
from oracle import texpose_oracle as O
from test_gpu_parity import _graph, cu, rel_l2

B, R, N = 3, 64, 64
rs = np.random.RandomState(7 * B + R)
params = O.make_params(21)
g, opt = _graph(params, N=N)
pts = torch.from_numpy(rs.uniform(-1.2, 1.2, size=(B, R, N, 3)).astype(np.float32))
unit = torch.nn.functional.normalize(torch.from_numpy(rs.normal(size=(B, R, 1, 3)).astype(np.float32)), dim=-1).expand(B, R, N, 3).contiguous()
lt = torch.from_numpy(rs.normal(size=(B, 16)).astype(np.float32))
ll = torch.from_numpy(rs.normal(size=(B, 48)).astype(np.float32))
cots = [torch.from_numpy(rs.normal(size=s).astype(np.float32)) for s in ((B, R, N, 3, 2), (B, R, N, 2), (B, R, N, 1))]

def oracle(dtype):
    po = {k: v.to(dtype).clone().requires_grad_(not k.startswith("mlp_feat")) for k, v in params.items()}
    lto, llo = lt.to(dtype).clone().requires_grad_(), ll.to(dtype).clone().requires_grad_()
    import oracle.texpose_oracle as OO
    # posenc in the requested dtype but with the fp32-rounded argument (as the reference computes it)
    out = OO.mlp_forward(po, pts.to(dtype), unit.to(dtype), lto, llo)
    sum((o * c.to(dtype)).sum() for o, c in zip(out, cots)).backward()
    return out, {k: v.grad for k, v in po.items() if v.grad is not None}, lto.grad, llo.grad

o32, g32, lt32, ll32 = oracle(torch.float32)
# fp64 oracle: freq tensor is float32 in posenc -> promote inside by monkeypatching posenc
def posenc64(x, L):
    freq = (2 ** torch.arange(L, dtype=torch.float32)) * np.pi
    spec = (x.float()[..., None] * freq).double()     # same fp32-rounded argument as the reference
    enc = torch.stack([spec.sin(), spec.cos()], dim=-2)
    return enc.reshape(*x.shape[:-1], -1)
O.posenc = posenc64
o64, g64, lt64, ll64 = oracle(torch.float64)

ltd, lld = cu(lt).requires_grad_(), cu(ll).requires_grad_()
outd = g.nerf.forward(opt, cu(pts), ray_unit=cu(unit), latent_variable_trans=ltd, latent_variable_light=lld, mode="train")
sum((o * cu(c)).sum() for o, c in zip(outd, cots)).backward()
print("forward rel-L2  hip-vs-f64 / f32-vs-f64 / hip-vs-f32")
for a, b, c, n in zip(outd, o32, o64, ("rgb", "density", "uncert")):
    print(f"  {n:8s} {rel_l2(a, c):.2e} {rel_l2(b, c):.2e} {rel_l2(a, b):.2e}   max-abs hip-f64 {float((a.cpu().double()-c).abs().max()):.2e} f32-f64 {float((b.double()-c).abs().max()):.2e}")
print("grad rel-L2  hip-vs-f64 / f32-vs-f64 / hip-vs-f32")
for k, p in g.nerf.named_parameters():
    if p.grad is None: continue
    print(f"  {k:20s} {rel_l2(p.grad, g64[k]):.2e} {rel_l2(g32[k], g64[k]):.2e} {rel_l2(p.grad, g32[k]):.2e}")
print(f"  lat_trans            {rel_l2(ltd.grad, lt64):.2e} {rel_l2(lt32, lt64):.2e} {rel_l2(ltd.grad, lt32):.2e}")
print(f"  lat_light            {rel_l2(lld.grad, ll64):.2e} {rel_l2(ll32, ll64):.2e} {rel_l2(lld.grad, ll32):.2e}")
