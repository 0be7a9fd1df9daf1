import sys, os
R_=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R_); sys.path.insert(0, os.path.join(R_, "tests"))
import numpy as np, torch
from oracle import texpose_oracle as O
from conftest import load_golden
from test_gpu_parity import _graph, cu, rel_l2
from texpose_amd import ops
g9 = load_golden("g9_render_train")
params = O.make_params(g9["seed"])
graph, opt = _graph(params, n_train=g9["n_train"], emb_seed=g9["emb_seed"], H=g9["H"], W=g9["W"], N=g9["N"])
c, r, zn, zf, depth = ops.raygen(cu(g9["intr"]), cu(g9["pose"]), H=g9["H"], W=g9["W"], n_samples=g9["N"], coords=cu(g9["coords"]), z_near=cu(g9["z_near"]), z_far=cu(g9["z_far"]), rand=cu(g9["rand"]))
idx = g9["sample_idx"]
po = {k: v.clone().requires_grad_(not k.startswith("mlp_feat")) for k, v in params.items()}
et = graph.latent_vars_trans.weight.detach().cpu().clone().requires_grad_()
el = graph.latent_vars_light.weight.detach().cpu().clone().requires_grad_()
rgb_o, den_o, unc_o = O.forward_samples(po, c.cpu(), r.cpu(), depth.cpu()[..., None], et[idx], el[idx])
for t in (rgb_o, den_o, unc_o): t.retain_grad()
ref = O.composite(r.cpu(), rgb_o, den_o, depth.cpu()[..., None], unc_o, 0.05)
ref = dict(rgb=ref[0], rgb_static=ref[1], rgb_transient=ref[2], depth=ref[3], uncert=ref[8], density=den_o)
cot = {k[4:]: v for k, v in g9.items() if k.startswith("cot_")}
sum((ref[k] * cot[k]).sum() for k in cot).backward()
# 1) composite bwd alone
g_out = torch.zeros(2, 16, 14)
for name, lo, hi in ops.COMPOSITE_RAY_FIELDS:
    if name in cot: g_out[..., lo:hi] = cot[name]
g_rgb, g_den, g_unc = ops.composite_bwd(r, cu(rgb_o.detach()), cu(den_o.detach()), depth[..., None], cu(unc_o.detach()), cu(g_out))
g_den = g_den + cu(cot["density"])
print("composite bwd: g_rgb", rel_l2(g_rgb, rgb_o.grad), "g_den_t", rel_l2(g_den[..., 1], den_o.grad[..., 1]), "g_unc", rel_l2(g_unc, unc_o.grad))
print("  max |g_den_t| ours/ref", float(g_den[...,1].abs().max()), float(den_o.grad[...,1].abs().max()))
d = (g_den[...,1].cpu()-den_o.grad[...,1]).abs()
print("  worst idx", np.unravel_index(int(d.argmax()), d.shape), float(d.max()))
# 2) MLP bwd with the oracle's cotangents
lt = graph.latent_vars_trans.weight[cu(idx)].detach().requires_grad_(); ll = graph.latent_vars_light.weight[cu(idx)].detach().requires_grad_()
out = graph.nerf.forward_samples(opt, c, r, depth[..., None], latent_variable_trans=lt, latent_variable_light=ll, mode="train")
(out[0]*cu(rgb_o.grad)).sum().add((out[1]*cu(den_o.grad)).sum()).add((out[2]*cu(unc_o.grad)).sum()).backward()
for k, p in graph.nerf.named_parameters():
    if p.grad is not None: print(f"  {k:20s} {rel_l2(p.grad, po[k].grad):.2e}")
print("fwd rel", rel_l2(out[0], rgb_o), rel_l2(out[1], den_o), rel_l2(out[2], unc_o))
print("max |cot den_t|", float(den_o.grad[...,1].abs().max()), "max unc cot", float(unc_o.grad.abs().max()), "max rgb cot", float(rgb_o.grad.abs().max()))
print("---- full graph")
graph.nerf.zero_grad()
stash = {}
orig = ops.mlp_backward
def spy(nerf, lt_, ll_, saved, rgb, density, uncert, g_rgb, g_density, g_uncert):
    stash.update(g_rgb=g_rgb, g_density=g_density, g_uncert=g_uncert, rgb=rgb, density=density, uncert=uncert, lt=lt_, ll=ll_)
    return orig(nerf, lt_, ll_, saved, rgb, density, uncert, g_rgb, g_density, g_uncert)
ops.mlp_backward = spy
dr = (cu(g9["z_near"])[:, :, None], cu(g9["z_far"])[:, :, None])
ret = graph.render(opt, cu(g9["pose"]), intr=cu(g9["intr"]), ray_idx=cu(g9["coords"]), depth_range=dr, sample_idx=cu(g9["sample_idx"]), mode="train", rand=cu(g9["rand"]))
sum((ret[k] * cu(cot[k])).sum() for k in cot).backward()
for k, ref_t in (("g_rgb", rgb_o.grad), ("g_density", den_o.grad), ("g_uncert", unc_o.grad), ("rgb", rgb_o), ("density", den_o), ("uncert", unc_o)):
    t = stash[k]
    print(k, None if t is None else (tuple(t.shape), t.is_contiguous(), rel_l2(t, ref_t.detach())))
print("g_density t-channel", rel_l2(stash["g_density"][..., 1], den_o.grad[..., 1]), "s-channel", rel_l2(stash["g_density"][..., 0], den_o.grad[..., 0]))
print("lat", rel_l2(stash["lt"], et[idx].detach()), rel_l2(stash["ll"], el[idx].detach()))
for k, p in graph.nerf.named_parameters():
    if p.grad is not None: print(f"  {k:20s} {rel_l2(p.grad, po[k].grad):.2e}")
