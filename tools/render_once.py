"""GPU box: render N full images with a chosen MLP precision (profiling target)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda:0")
sc, params, emb_t, emb_l = bench.build_scene(dev, 0)
g, opt = bench.make_graph(dev, params, emb_t, emb_l)
g.nerf.precision = prec
if os.environ.get("TP_RANGE_CHECK_OFF"):            # timing experiments with deliberately wrong arithmetic
    opt.arch.mlp_range_check = "off"
pose, intr = sc["pose"].to(dev), sc["intr"].to(dev)
dr = (sc["z_near"].to(dev)[:, :, None], sc["z_far"].to(dev)[:, :, None])
mask = torch.ones(1, 480, 640, device=dev)
import time
with torch.no_grad():
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ret = g.render_by_slices(opt, pose, intr=intr, depth_range=dr, object_mask=mask, sample_idx=None, mode="val")
        torch.cuda.synchronize()
        print("image ms", (time.perf_counter() - t0) * 1e3)
print("done", float(ret.rgb.mean()))
