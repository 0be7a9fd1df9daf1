"""GPU box: boundaries of the captured graphs of the B=4 GAN iteration in the FREE-RUNNING benchmark configuration (pipeline_disc_tail +
defer_results): TP_STAMPS one-thread launches at every graph boundary plus two around the tp_step_inputs launch, read after every 40th
iteration, mean of 8 (profiles/r6/03-06).  python tools/free_running_timeline.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["TP_STAMPS"] = "1"
import torch
from texpose_amd import knobs
knobs.reload()
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
tr.pipeline_disc_tail = tr.defer_results = True
var = training_batch(4, 128, 128, device="cuda:0")
for _ in range(20):
    tr.train_iteration(AttrDict(dict(var)))
torch.cuda.synchronize()
from texpose_amd import ops as _ops
_orig_si = _ops.step_inputs
tr._stamp_names += ["SI.0", "SI.1"]
def _si(*a, **k):
    _ops.stamp(tr._stamps, tr._stamp_names.index("SI.0"))
    _orig_si(*a, **k)
    _ops.stamp(tr._stamps, tr._stamp_names.index("SI.1"))
_ops.step_inputs = _si
for _ in range(5):
    tr.train_iteration(AttrDict(dict(var)))
torch.cuda.synchronize()
n = len(tr._stamp_names)
names = tr._stamp_names
acc = None
K = 8
for rep in range(K):
    for _ in range(40):
        tr.train_iteration(AttrDict(dict(var)))
    torch.cuda.synchronize()
    st = tr._stamps[:n].cpu().double()
    g10 = st[names.index("G1.0")]
    rel = (st - g10) / 100.0
    acc = rel if acc is None else acc + rel
print("free-running (pipelined + deferred), mean of %d last-iterations: us after G1 start (negative: launched in the previous period)" % K)
for nm, v in sorted(zip(names, (acc / K).tolist()), key=lambda x: x[1]):
    print("  %-6s %8.1f" % (nm, v))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(300):
    tr.train_iteration(AttrDict(dict(var)))
e1.record()
torch.cuda.synchronize()
print("period %.1f us" % (e0.elapsed_time(e1) / 300 * 1e3))
tr.finish()
