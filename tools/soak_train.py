"""GPU box: N (default 600) hipGraph-replayed GAN iterations (f16x3 recording forward, split-fp16 backward, explicit discriminator
schedule, prefetched spectral norms, in-kernel random draws): finite losses and parameters, range flag clear, no withheld step, the
step counter of the random draws in step with the iteration count.     python tools/soak_train.py [iterations]"""
import sys, os, json, torch
sys.path.insert(0, os.getcwd())
from texpose_amd.gan_modules import Discriminator, PerceptualLoss
from texpose_amd.graph import Graph
from texpose_amd.options import default_options, AttrDict
from texpose_amd.synthetic import training_batch
from texpose_amd.trainer import GraphedGanTrainer
from texpose_amd import ops
torch.manual_seed(0)
opt = default_options(H=128, W=128, device="cuda:0")
opt.batch_size, opt.patch_size, opt.nerf.sample_intvs = 4, 16, 64
graph = Graph(opt, discriminator=Discriminator(opt), perceptual_loss=PerceptualLoss()).to("cuda:0")
tr = GraphedGanTrainer(opt, graph, n_train=189)
if os.environ.get("TP_SOAK_STRICT") != "1":               # (the benchmark loops' configuration; results are read behind wait_all())
    tr.pipeline_disc_tail = tr.defer_results = True
var = training_batch(4, 128, 128, device="cuda:0")
hist = []
N_IT = int(sys.argv[1]) if len(sys.argv) > 1 else 600
for it in range(N_IT):
    _, loss = tr.train_iteration(AttrDict(dict(var)))
    if it % max(1, N_IT // 6) == 0 or it == N_IT - 1:
        tr.wait_all()
        hist.append({k: round(float(v), 5) for k, v in loss.items()})
        print(it, hist[-1], flush=True)
tr.flush_flags()                                          # (the last replay's gate words: a withheld final step raises here)
ops.check_mlp_status("cuda:0")
sd = graph.state_dict()
assert all(torch.isfinite(v).all() for v in sd.values() if v.dtype.is_floating_point)
assert all(all(abs(v) < 1e6 for v in h.values()) for h in hist)      # (random-noise target images: no loss trend to expect)
assert tr.skipped_steps == 0 and not graph.discriminator._sn_queue
counter = int(tr._rng_counter) if getattr(tr, "_rng_counter", None) is not None else None
assert counter is None or counter >= N_IT, counter            # (+ the warm-up iterations of the capture)
import hashlib
h = hashlib.sha256()
for k in sorted(sd):
    h.update(k.encode() + sd[k].detach().cpu().contiguous().numpy().tobytes())
for st in list(tr.optim_disc.state.values()) + list(tr.optim_nerf.state.values()):
    for k in sorted(st):
        if torch.is_tensor(st[k]):
            h.update(st[k].detach().cpu().contiguous().numpy().tobytes())
print("state sha256 (parameters, buffers, optimiser state):", h.hexdigest()[:32], "deferred / pipelined:", tr._defers_results(), tr._pipelined())
print("soak ok; step counter", counter, "patch sampler iterations", graph.patch_sampler.iterations, "progress", float(graph.discriminator.progress))
