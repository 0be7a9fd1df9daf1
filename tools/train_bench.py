"""GPU box: training iterations / s of the full GAN loop (config C3: B images, 16x16 patches of 128x128 crops, N=64)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from texpose_amd.gan_modules import Discriminator, PerceptualLoss
from texpose_amd.graph import Graph
from texpose_amd.options import default_options, AttrDict
from texpose_amd.synthetic import training_batch
from texpose_amd.trainer import GanTrainer, GraphedGanTrainer


def run(B=4, iters=20, warm=6, full=True, device="cuda:0", graphed=False, train_precision="fp32"):
    torch.manual_seed(0)
    torch.backends.cudnn.benchmark = os.environ.get("TP_MIOPEN_FIND", "1") == "1"   # let MIOpen search conv solvers
    opt = default_options(H=128, W=128, device=device)
    opt.batch_size, opt.patch_size, opt.nerf.sample_intvs = B, 16, 64
    if not full:
        opt.loss_weight.feat = None
        opt.loss_weight.gan_nerf = None
        opt.gan = None
    graph = Graph(opt, discriminator=Discriminator(opt) if full else None,
                  perceptual_loss=PerceptualLoss() if full else None).to(device)
    graph.nerf.train_precision = train_precision
    # TP_PRE_STREAMS=k: k streams made (and used once) before the trainer makes its own -- shifts the round-robin assignment of hardware
    # queues, the way earlier work in a larger process does (robustness check of the captured step's stream layout)
    pre = [torch.cuda.Stream(device=device) for _ in range(int(os.environ.get("TP_PRE_STREAMS", "0")))]
    for st in pre:
        with torch.cuda.stream(st):
            torch.zeros(1, device=device)
    tr = (GraphedGanTrainer if graphed else GanTrainer)(opt, graph, n_train=189)
    if graphed and os.environ.get("TP_NO_PIPELINE_DISC") != "1":
        tr.pipeline_disc_tail = True           # (losses / state are read behind the final synchronise only)
        tr.defer_results = os.environ.get("TP_NO_DEFER") != "1"
    var = training_batch(B, 128, 128, device=device)
    for _ in range(warm):
        tr.train_iteration(AttrDict(dict(var)))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        tr.train_iteration(AttrDict(dict(var)))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    return dict(iters_per_s=1 / dt, ms_per_iter=dt * 1e3, batch=B, launch="hipGraph replay" if graphed else "eager", recording_forward=train_precision, rays_per_iter=B * 256, samples_per_iter=B * 256 * 64,
                loop="full GAN (render fwd+bwd, gathers, random-init VGG19[:15] feature loss, PatchGAN + R1, Adam + RMSprop)"
                if full else "nerf step only (render fwd+bwd, photometric/uncert/trans_reg losses, Adam)")


if __name__ == "__main__":
    if len(sys.argv) > 1:                                   # train_bench.py <batch> <full 0|1> [iters] [graphed 0|1]
        print(json.dumps(run(B=int(sys.argv[1]), full=bool(int(sys.argv[2])), iters=int(sys.argv[3]) if len(sys.argv) > 3 else 20,
                             graphed=bool(int(sys.argv[4])) if len(sys.argv) > 4 else False,
                             train_precision=sys.argv[5] if len(sys.argv) > 5 else "fp32")))
    else:
        for B in (4, 32):
            for full in (True, False):
                for graphed in (False, True):
                    for tp in ("fp32", "f16x3"):
                        print(json.dumps(run(B=B, full=full, graphed=graphed, train_precision=tp)))
