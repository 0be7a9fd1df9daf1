"""GPU box: time of the K11 convolution kernels (csrc/patch_conv.hip) next to torch / MIOpen on the PatchGAN's shapes.
tools/conv_bench.py [images]   (default 4: the two ladder stages of a 16x16 patch, 9 -> 256 -> 512 channels)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from texpose_amd import ops


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


N = int(sys.argv[1]) if len(sys.argv) > 1 else 4
torch.backends.cudnn.benchmark = True
for C_in, H, Co in ((9, 16, 256), (256, 8, 512)):
    x = torch.randn(N, C_in, H, H, device="cuda")
    w = torch.randn(Co, C_in, 4, 4, device="cuda")
    gy = torch.randn(N, Co, H // 2, H // 2, device="cuda")
    xr, wr = x.clone().requires_grad_(), w.clone().requires_grad_()
    flop = 2 * N * (H // 2) ** 2 * Co * C_in * 16
    print("N=%d  %d -> %d channels, %dx%d map, %.1f MFLOP per op" % (N, C_in, Co, H, H, flop / 1e6))
    print("  fwd    %7.1f us   (torch conv2d %7.1f us)" % (timed(lambda: ops.conv4s2_fwd(x, w)), timed(lambda: F.conv2d(x, w, None, 2, 1))))
    print("  dgrad  %7.1f us   (torch %7.1f us)" % (timed(lambda: ops.conv4s2_dgrad(gy, w)),
                                                    timed(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, (2, 2), (1, 1), (1, 1), False, (0, 0), 1, (True, False, False)))))
    print("  wgrad  %7.1f us   (torch %7.1f us)" % (timed(lambda: ops.conv4s2_wgrad(gy, x)),
                                                    timed(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, (2, 2), (1, 1), (1, 1), False, (0, 0), 1, (False, True, False)))))
