"""GPU box: what each chain of the captured B=4 GAN iteration costs the others -- the free-running iteration with some graphs LEFT OUT of
the replay (timing only: such an iteration computes nothing meaningful), stamps at the graph boundaries (profiles/r6/40).
python tools/window_probe.py"""
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
class Skip:
    def replay(self): pass
def run(skip):
    torch.manual_seed(0)
    opt = default_options(H=128, W=128, device="cuda:0")
    opt.batch_size, opt.patch_size, opt.nerf.sample_intvs = 4, 16, 64
    graph = Graph(opt, discriminator=Discriminator(opt), perceptual_loss=PerceptualLoss()).to("cuda:0")
    graph.nerf.train_precision = "f16x3"
    tr = GraphedGanTrainer(opt, graph, n_train=189)
    tr.pipeline_disc_tail = tr.defer_results = True
    var = training_batch(4, 128, 128, device="cuda:0")
    for _ in range(10):
        tr.train_iteration(AttrDict(dict(var)))
    torch.cuda.synchronize()
    for k in skip:
        tr._graphs[k] = Skip()
    names = tr._stamp_names
    n = len(names)
    acc = None
    for rep in range(6):
        for _ in range(30):
            tr.train_iteration(AttrDict(dict(var)))
        torch.cuda.synchronize()
        st = tr._stamps[:n].cpu().double()
        rel = (st - st[names.index("G1.0")]) / 100.0
        acc = rel if acc is None else acc + rel
    d = dict(zip(names, (acc / 6).tolist()))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        tr.train_iteration(AttrDict(dict(var)))
    e1.record(); torch.cuda.synchronize()
    print("skip %-22s period %7.1f | G1.1 %6.1f F %6.1f -> %6.1f (%5.1f) G2a %6.1f -> %6.1f  G2b.0 %6.1f G2b.1 %6.1f" % (
        ",".join(skip) or "-", e0.elapsed_time(e1) / 200 * 1e3, d["G1.1"], d["F.0"], d["F.1"], d["F.1"] - d["F.0"], d.get("G2a.0", 0), d.get("G2a.1", 0), d["G2b.0"], d["G2b.1"]), flush=True)
    del tr, graph
for skip in ([], ["D2a", "D2b"], ["D1b", "D2a", "D2b"], ["G2a"], ["G2a", "D1b", "D2a", "D2b"], ["D1", "D1b", "G2a", "D2a", "D2b"]):
    run(skip)
