"""GPU box: one training iteration, eager-capturable vs hipGraph trainer: per-key update comparison."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import texpose_oracle as O
from texpose_amd.gan_modules import Discriminator
from texpose_amd.graph import Graph
from texpose_amd.options import default_options, AttrDict
from texpose_amd.synthetic import training_batch
from texpose_amd.trainer import GanTrainer, GraphedGanTrainer
dev = torch.device("cuda:0")
B, H, W, n_train, N = 2, 32, 32, 5, 8
def build(cls):
    opt = default_options(H=H, W=W, device="cuda:0")
    opt.batch_size, opt.patch_size, opt.nerf.sample_intvs = B, 16, N
    opt.loss_weight.feat = None
    graph = Graph(opt, discriminator=Discriminator(opt)).to(dev)
    graph.train(); graph.nerf.precision = "fp32"
    return cls(opt, graph, n_train=n_train), graph
eager, g_e = build(type("EagerCapturable", (GanTrainer,), dict(capturable=True)))
g_e.nerf.load_state_dict({**g_e.nerf.state_dict(), **{k: v.to(dev) for k, v in O.make_params(5).items()}})
dcpu = Discriminator(eager.opt); O.seed_spectral_module(dcpu, 9); g_e.discriminator.load_state_dict(dcpu.state_dict())
snap = {k: v.detach().clone() for k, v in g_e.state_dict().items()}
batch = training_batch(B, H, W, n_train=n_train, seed=1, device="cuda:0")
rnd = (torch.rand(3, B, 1, 1, 1, device=dev), torch.rand(B, 256, N, 1, device=dev))
graphed, g_g = build(GraphedGanTrainer)
g_g.load_state_dict(snap)
ex = AttrDict(dict(batch)); ex.patch_u, ex.jitter_rand = rnd
graphed.capture(ex, warmup=2)
print("after capture: params restored?", all(torch.equal(v, snap[k]) for k, v in g_g.state_dict().items()))
print("opt state after capture:", {n: [float(v.abs().max()) for v in list(st.values())] for n, st in list(enumerate(graphed.optim_nerf.state.values()))[:2]})
v = AttrDict(dict(batch)); v.patch_u, v.jitter_rand = rnd
eager.train_iteration(v)
v = AttrDict(dict(batch)); v.patch_u, v.jitter_rand = rnd
graphed.train_iteration(v)
torch.cuda.synchronize()
print("lr_used", float(graphed.lr_nerf_used), float(eager.lr_nerf_used), "bad", graphed._bad.tolist())
for k in snap:
    da, db = (g_e.state_dict()[k] - snap[k]).float(), (g_g.state_dict()[k] - snap[k]).float()
    if float(da.abs().max()) > 0 or float(db.abs().max()) > 0:
        print("%-40s eager |d| %.3e graphed |d| %.3e  diff %.3e" % (k, float(da.abs().max()), float(db.abs().max()), float((da - db).abs().max())))
