"""GPU box: which framework-level operators launch the kernels of ONE eager training iteration (tools/op_trace.py [B]).
Prints, in launch order, every top-level aten / custom operator that launched at least one kernel with its input shapes, the
number of kernels and whether it ran inside autograd's backward; the tiny launches to fuse next are read off this list."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity, record_function
from texpose_amd.gan_modules import Discriminator, PerceptualLoss
from texpose_amd.graph import Graph
from texpose_amd.options import default_options, AttrDict
from texpose_amd.synthetic import training_batch
from texpose_amd.trainer import GanTrainer

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
torch.manual_seed(0)
opt = default_options(H=128, W=128, device="cuda:0")
opt.batch_size, opt.patch_size, opt.nerf.sample_intvs = B, 16, 64
graph = Graph(opt, discriminator=Discriminator(opt), perceptual_loss=PerceptualLoss()).to("cuda:0")
graph.nerf.train_precision = "f16x3"
tr = GanTrainer(opt, graph, n_train=189)
tr.capturable = True
var = training_batch(B, 128, 128, device="cuda:0")
for _ in range(4):
    tr.train_iteration(AttrDict(dict(var)))
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    v = tr.graph.get_ray_idx(opt, AttrDict(dict(var)))
    with record_function("## nerf_step"):
        v, _ = tr.nerf_step(v)
    with record_function("## disc_step"):
        tr.disc_step(v)
    torch.cuda.synchronize()
evs = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CPU]
evs.sort(key=lambda e: e.time_range.start)
total = 0


def n_kernels(e):
    return len(e.kernels) + sum(n_kernels(c) for c in e.cpu_children)


def walk(e, depth):
    global total
    k = n_kernels(e)
    if k == 0 and not e.name.startswith("##"):
        return
    # descend through wrappers (autograd nodes, record_function ranges); print the first level that owns kernels directly
    kids = [c for c in e.cpu_children if n_kernels(c)]
    if e.name.startswith("##") or (not e.kernels and kids and (e.name.startswith("autograd::") or "Backward" in e.name or e.name.endswith("Function") or depth < 1)):
        print("%s%s  [%d kernels]" % ("  " * depth, e.name[:80], k))
        for c in e.cpu_children:
            walk(c, depth + 1)
        return
    shapes = str(e.input_shapes)[:90] if e.input_shapes else ""
    total += k
    print("%s%-44s %2d  %s" % ("  " * depth, e.name[:44], k, shapes))


for e in evs:
    if e.cpu_parent is None:
        walk(e, 0)
print("kernels attributed:", total)
