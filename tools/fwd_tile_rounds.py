"""GPU box: the f16x3 MLP forward (plain and recording) by the number of 128-sample tile rounds per CU -- per-round rate, start-up share,
what the activation record costs (docs/lab-notebook-r6.md section 2).  python tools/fwd_tile_rounds.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from texpose_amd import ops
from texpose_amd.graph import Graph
from texpose_amd.options import default_options
dev = torch.device("cuda:0")
opt = default_options(H=128, W=128, device="cuda:0")
g = Graph(opt).to(dev)
nerf = g.nerf
R, N = 256, 64
torch.manual_seed(0)
def timeit(fn, reps=40):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
packed = nerf.packed_weights("f16x3")
for B in (1, 2, 3, 4, 6, 8, 16, 32):
    center = torch.randn(B, R, 3, device=dev) * 0.1
    ray = torch.nn.functional.normalize(torch.randn(B, R, 3, device=dev), dim=-1)
    depth = torch.rand(B, R, N, device=dev).sort(-1).values + 0.5
    lt, ll = torch.randn(B, 16, device=dev), torch.randn(B, 48, device=dev)
    t = timeit(lambda: ops.mlp_forward(packed, lt, ll, center=center, ray=ray, depth=depth, save=False, precision="f16x3"))
    res = ops.mlp_forward(packed, lt, ll, center=center, ray=ray, depth=depth, save=True, precision="f16x3")
    sv = res[3]
    tr = timeit(lambda: ops.mlp_forward(packed, lt, ll, center=center, ray=ray, depth=depth, save=True, precision="f16x3", saved_out=sv))
    # one launch alone between events (no back-to-back overlap of launch overhead)
    def single():
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.mlp_forward(packed, lt, ll, center=center, ray=ray, depth=depth, save=True, precision="f16x3", saved_out=sv); e1.record()
        torch.cuda.synchronize(); return e0.elapsed_time(e1) * 1e3
    s1 = sorted(single() for _ in range(9))[4]
    print("B=%2d tiles/CU=%5.2f  plain %7.1f us  recording %7.1f us  (single launch between events: %7.1f)" % (B, B * R * N / 128 / 256, t, tr, s1))
