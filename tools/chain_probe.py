"""GPU box: each graph of the six-graph training iteration replayed ALONE on an otherwise idle GPU (how long is a chain by itself,
without the other streams' kernels and without the MLP kernels holding every CU), next to its span inside the running iteration
(tools/linear_timeline.py).  chain_probe.py [reps]      -- run it under `rocprofv3 --kernel-trace --stats` for per-kernel spans."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from texpose_amd.gan_modules import Discriminator, PerceptualLoss
from texpose_amd.graph import Graph
from texpose_amd.options import default_options, AttrDict
from texpose_amd.synthetic import training_batch
from texpose_amd.trainer import GraphedGanTrainer

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
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
# the whole iteration, free-running, for reference
for _ in range(20):
    tr.train_iteration(AttrDict(dict(var)))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200):
    tr.train_iteration(AttrDict(dict(var)))
e1.record()
torch.cuda.synchronize()
print("iteration, free-running: %.1f us (%.0f it/s)" % (e0.elapsed_time(e1) / 200 * 1e3, 200 / e0.elapsed_time(e1) * 1e3))
tr.flush_flags()
streams = {"D1": tr._side, "G1": tr._capture_stream, "F": tr.graph.feat_stream, "G2a": tr._capture_stream, "G2b": tr._capture_stream,
           "D2": tr._side, "D2a": tr._side, "D2b": tr._side, "D1b": tr._side}
print("graph   us per replay, alone on the GPU (%d back-to-back replays between two HIP events)" % reps)
total = 0.0
for name in [n for n in ("G1", "F", "G2a", "G2b", "D1", "D1b", "D2", "D2a", "D2b") if n in tr._graphs]:
    g, st = tr._graphs[name], streams[name]
    with torch.cuda.stream(st):
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(reps):
            g.replay()
        e1.record(st)
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    total += us
    print("  %-4s %8.1f" % (name, us))
print("  sum  %8.1f   (generator loop G1 + F + G2b: the serial chain of an iteration)" % total)
