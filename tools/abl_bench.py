import os, sys, time, subprocess
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = '''
import os, sys, time
sys.path.insert(0, %r)
import torch, bench
dev = torch.device("cuda:0")
sc, params, emb_t, emb_l = bench.build_scene(dev, 0)
g, opt = bench.make_graph(dev, params, emb_t, emb_l)
g.nerf.precision = "f16x3"
pose, intr = sc["pose"].to(dev), sc["intr"].to(dev)
dr = (sc["z_near"].to(dev)[:, :, None], sc["z_far"].to(dev)[:, :, None])
mask = torch.ones(1, 480, 640, device=dev)
with torch.no_grad():
    g.render_by_slices(opt, pose, intr=intr, depth_range=dr, object_mask=mask, sample_idx=None, mode="val")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(2): g.render_by_slices(opt, pose, intr=intr, depth_range=dr, object_mask=mask, sample_idx=None, mode="val")
    torch.cuda.synchronize(); print("TP_ABL=%%s ms/image %%.1f" %% (os.environ.get("TP_ABL"), (time.perf_counter()-t0)/2*1e3))
''' % R
for abl in (0, 1, 2, 3, 4, 7):
    subprocess.run([sys.executable, "-c", code], env=dict(os.environ, TP_ABL=str(abl)))
