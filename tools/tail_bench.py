#!/usr/bin/env python3
"""GPU box: microseconds per launch of the PatchGAN tail kernels (K17) next to the K15 + K14 launches they replace, each replayed
20x from a hipGraph between two HIP events (bench._event_ms).  Usage: tools/tail_bench.py [B]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                                                    # noqa: E402
import bench                                                                    # noqa: E402
from texpose_amd import ops                                                     # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
d = torch.device("cuda:0")
torch.manual_seed(0)
K, N, H, L = 8192, 64, 64, 4
a, W0 = torch.randn(B, K, device=d), torch.randn(N, K, device=d) / K ** 0.5
scale = torch.rand(B, device=d) * 0.7 + 0.3
W1, W2, W3 = torch.randn(H, N + 2 * L + 1, device=d) / 8, torch.randn(H, H, device=d) / 8, torch.randn(1, H, device=d) / 8
out, t0, t1, t2 = ops.disc_tail_fwd(a, W0, scale, W1, W2, W3, L, 0.2)
g, ones = torch.randn(B, device=d), torch.ones(B, device=d)
gy2, a2 = torch.randn(B, N, device=d), torch.randn(B, K, device=d)
r = ops.disc_tail_bwd(ones, t0, t1, t2, W0, W1, W2, W3, L, 0.2, want_gW0=False, head_weight_grads=False, want_e=True)
e1, e2 = r["e1"], r["e2"]
acc = (torch.zeros_like(W1), torch.zeros_like(W2), torch.zeros_like(W3))
z = ops.skinny_linear_fwd(a, W0)
cases = {
    "tail_fwd": lambda: ops.disc_tail_fwd(a, W0, scale, W1, W2, W3, L, 0.2),
    "  skinny_fwd": lambda: ops.skinny_linear_fwd(a, W0),
    "  head_fwd": lambda: ops.disc_head_fwd(z, scale, W1, W2, W3, L, 0.2),
    "tail_bwd data-only (+e)": lambda: ops.disc_tail_bwd(ones, t0, t1, t2, W0, W1, W2, W3, L, 0.2, want_gW0=False, head_weight_grads=False, want_e=True),
    "tail_bwd all gradients": lambda: ops.disc_tail_bwd(g, t0, t1, t2, W0, W1, W2, W3, L, 0.2, a=a),
    "tail_bwd all + R1 rows + accumulate": lambda: ops.disc_tail_bwd(g, t0, t1, t2, W0, W1, W2, W3, L, 0.2, a=a, gy2=gy2, a2=a2, accumulate_into=acc),
    "  head_bwd data-only": lambda: ops.disc_head_bwd(ones, t0, t1, t2, W1, W2, W3, N, L, 0.2, weight_grads=False),
    "  head_bwd all": lambda: ops.disc_head_bwd(g, t0, t1, t2, W1, W2, W3, N, L, 0.2),
    "  skinny_wgrad": lambda: ops.skinny_linear_wgrad(z, a),
    "  skinny_dgrad (mm)": lambda: ops.skinny_linear_dgrad(z, W0),
    "tail_bwd_bwd": lambda: ops.disc_tail_bwd_bwd(a2, ones, t0, t1, t2, e1, e2, W0, W1, W2, W3, L, 0.2),
    "  head_bwd_bwd": lambda: ops.disc_head_bwd_bwd(z, ones, t0, t1, t2, e1, e2, W1, W2, W3, L, 0.2),
}
for name, fn in cases.items():
    print("%-40s %7.2f us" % (name, bench._event_ms(fn, 20) * 1e3))
if "--timing" in sys.argv:                 # TEXPOSE_AMD_LIB=.../libtexpose_amd_tail_timing.so: 100 MHz section stamps of one workgroup
    import ctypes
    from texpose_amd import _lib
    lib = _lib.load()
    buf = (ctypes.c_ulonglong * 16)()
    for name in ("tail_fwd", "tail_bwd data-only (+e)", "tail_bwd all + R1 rows + accumulate", "tail_bwd_bwd"):
        for rep in range(3):
            torch.cuda.synchronize()
            cases[name]()
            torch.cuda.synchronize()
            lib.tp_disc_tail_stamps(buf)
            v = list(buf)
            print("%-40s stamps (us since the first): %s" % (name, ["%.2f" % ((x - v[0]) / 100.0) for x in v[:10]]))
