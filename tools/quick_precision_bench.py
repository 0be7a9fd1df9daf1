"""GPU box: full-image (480x640x128) render time with the exact-fp32 and the f16x3 MLP kernels."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from texpose_amd import ops
dev = torch.device("cuda:0")
sc, params, emb_t, emb_l = bench.build_scene(dev, 0)
outs = {}
for prec in ("fp32", "f16x3"):
    g, opt = bench.make_graph(dev, params, emb_t, emb_l)
    g.nerf.precision = prec
    pose, intr = sc["pose"].to(dev), sc["intr"].to(dev)
    dr = (sc["z_near"].to(dev)[:, :, None], sc["z_far"].to(dev)[:, :, None])
    mask = torch.ones(1, 480, 640, device=dev)
    opt.nerf.sample_stratified = False
    with torch.no_grad():
        ret = g.render_by_slices(opt, pose, intr=intr, depth_range=dr, object_mask=mask, sample_idx=None, mode="val")
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3):
            ret = g.render_by_slices(opt, pose, intr=intr, depth_range=dr, object_mask=mask, sample_idx=None, mode="val")
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    ops.check_mlp_status(dev)
    outs[prec] = {k: ret[k].clone() for k in ("rgb", "depth", "uncert", "density")}
    print(prec, "ms/image %.2f" % (dt * 1e3), "rays/s %.0f" % (307200 / dt))
for k in outs["fp32"]:
    a, b = outs["fp32"][k].double(), outs["f16x3"][k].double()
    print(k, "f16x3 vs fp32: rel-L2 %.2e  max-rel %.2e" % (float((a - b).norm() / b.norm()), float(((a - b).abs() / (b.abs() + 1e-6)).max())))
