import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tests.test_gpu_parity as T
O = T.O
B, R, N = [int(x) for x in sys.argv[1:4]]
rs = np.random.RandomState(7 * B + R)
params = O.make_params(21)
g, opt = T._graph(params, N=N)
g.nerf.train_precision = "f16x3"
cu = T.cu
pts = torch.from_numpy(rs.uniform(-1.2, 1.2, size=(B, R, N, 3)).astype(np.float32))
unit = torch.nn.functional.normalize(torch.from_numpy(rs.normal(size=(B, R, 1, 3)).astype(np.float32)), dim=-1).expand(B, R, N, 3).contiguous()
lt = torch.from_numpy(rs.normal(size=(B, 16)).astype(np.float32))
ll = torch.from_numpy(rs.normal(size=(B, 48)).astype(np.float32))
cots = [torch.from_numpy(rs.normal(size=s).astype(np.float32)) for s in ((B, R, N, 3, 2), (B, R, N, 2), (B, R, N, 1))]
ltd, lld = cu(lt).requires_grad_(), cu(ll).requires_grad_()
outd = g.nerf.forward(opt, cu(pts), ray_unit=cu(unit), latent_variable_trans=ltd, latent_variable_light=lld, mode="train")
sum((o * cu(c)).sum() for o, c in zip(outd, cots)).backward()
out = {k: p.grad.detach().cpu() for k, p in g.nerf.named_parameters() if p.grad is not None}
out["lt"] = ltd.grad.cpu(); out["ll"] = lld.grad.cpu()
torch.save(out, sys.argv[4])
if len(sys.argv) > 5:
    ref = torch.load(sys.argv[5])
    for k in out:
        d = (out[k] - ref[k]).abs()
        bad = d > 1e-3 * ref[k].abs().max()
        print(k, tuple(out[k].shape), "bad", int(bad.sum()), "maxdiff", float(d.max()))
        if bad.any() and out[k].dim() == 2:
            rows = bad.any(1).nonzero().flatten().tolist(); cols = bad.any(0).nonzero().flatten().tolist()
            print("   rows", rows[:40], "n", len(rows), " cols", cols[:40], "n", len(cols))
        elif bad.any():
            print("   idx", bad.nonzero().flatten().tolist()[:40])
