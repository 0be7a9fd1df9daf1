"""GPU box: untraced timeline of the six-graph training iteration -- HIP events on the three streams around every graph replay,
averaged over iterations (a kernel trace slows the host's launches enough to change the picture).  linear_timeline.py [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from texpose_amd.gan_modules import Discriminator, PerceptualLoss
from texpose_amd.graph import Graph
from texpose_amd.options import default_options, AttrDict
from texpose_amd.synthetic import training_batch
from texpose_amd.trainer import GraphedGanTrainer

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
order = os.environ.get("TP_TIMELINE_ORDER", "product")
os.environ["TP_STAMPS"] = "1"
from texpose_amd import knobs
knobs.reload()
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
assert tr._linear
# the stamps of the LAST iteration (each is a captured one-thread launch at a segment boundary: ~5 us of perturbation apiece)
n = len(tr._stamp_names)
acc = torch.zeros(n, dtype=torch.float64)
per = []
prev = None
for _ in range(iters):
    tr.train_iteration(AttrDict(dict(var)))
    torch.cuda.synchronize()
    st = tr._stamps[:n].cpu().double()
    acc += (st - st[tr._stamp_names.index("G1.0")]) / 100.0
    if prev is not None:
        per.append(float(st[tr._stamp_names.index("G1.0")] - prev) / 100.0)
    prev = st[tr._stamp_names.index("G1.0")]
print("segment boundaries in us after the iteration's G1 start, mean of %d SYNCHRONISED iterations" % iters)
for name, v in zip(tr._stamp_names, (acc / iters).tolist()):
    print("  %-6s %8.1f" % (name, v))
# free-running: only the last iteration's stamps are read
for _ in range(50):
    tr.train_iteration(AttrDict(dict(var)))
torch.cuda.synchronize()
st = tr._stamps[:n].cpu().double()
print("free-running, last of 50 iterations:")
for name, v in zip(tr._stamp_names, ((st - st[tr._stamp_names.index("G1.0")]) / 100.0).tolist()):
    print("  %-6s %8.1f" % (name, v))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200):
    tr.train_iteration(AttrDict(dict(var)))
e1.record()
torch.cuda.synchronize()
print("period, free-running: %.1f us (pipeline_disc_tail %s, defer_results %s)" % (e0.elapsed_time(e1) / 200 * 1e3, tr._pipelined(), tr._defers_results()))
