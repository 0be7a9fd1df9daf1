import os, sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import torch
from texpose_amd.gan_modules import Discriminator, PerceptualLoss
from texpose_amd.graph import Graph
from texpose_amd.options import default_options, AttrDict
from texpose_amd.synthetic import training_batch
from texpose_amd.trainer import GraphedGanTrainer
torch.manual_seed(0)
opt = default_options(H=128, W=128, device="cuda:0")
opt.batch_size, opt.patch_size, opt.nerf.sample_intvs = 4, 16, 64
graph = Graph(opt, discriminator=Discriminator(opt), perceptual_loss=PerceptualLoss()).to("cuda:0")
graph.nerf.train_precision = "f16x3"
tr = GraphedGanTrainer(opt, graph, n_train=189)
var = training_batch(4, 128, 128, device="cuda:0")
for _ in range(10):
    tr.train_iteration(AttrDict(dict(var)))
torch.cuda.synchronize()
# whole iteration host time without sync
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    tr.train_iteration(AttrDict(dict(var)))
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("50 iterations: host issue %.1f us/iter, with final sync %.1f us/iter" % ((t1 - t0) / 50 * 1e6, (t2 - t0) / 50 * 1e6))
# (single segments replayed out of order: garbage state afterwards, this goes last)
if tr._linear:
    print("mode:", "six linear graphs on three streams" if tr._linear else "four graphs on two streams")
    for name, g in tr._graphs.items():
        s = tr._side if name[0] == "D" else tr.graph.feat_stream if name == "F" else torch.cuda.current_stream()
        ts = []
        for _ in range(20):
            torch.cuda.synchronize()
            with torch.cuda.stream(s):
                t0 = time.perf_counter(); g.replay(); t1 = time.perf_counter()
            ts.append((t1 - t0) * 1e6)
        print(name, "host us per replay: min %.1f median %.1f" % (min(ts), sorted(ts)[10]))
else:
    ts = []
    for _ in range(20):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); tr._graph.replay(); t1 = time.perf_counter()
        ts.append((t1 - t0) * 1e6)
    print("one graph host us per replay: min %.1f median %.1f" % (min(ts), sorted(ts)[10]))
