"""GPU box: what ONE dependent launch costs inside a replayed hipGraph on this stack -- a chain of n one-thread tp_stamp launches
(device-clock writes) captured into a graph: (last stamp - first stamp) / (n - 1) from the device clock, and the event-bracketed
replay time / n.  The floor under every launch of the captured training step.   boundary_probe.py [n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from texpose_amd import ops

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda:0")
slots = torch.zeros(n, dtype=torch.int64, device=dev)
big = torch.zeros(64 << 20, dtype=torch.float32, device=dev)          # 256 MB: dirty lines for the "after a writer" case
st = torch.cuda.Stream()


def capture(body):
    g = torch.cuda.CUDAGraph()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        body()
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=st):
        body()
    return g


def run(g, reps=20):
    with torch.cuda.stream(st):
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(reps):
            g.replay()
        e1.record(st)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


g = capture(lambda: [ops.stamp(slots, i) for i in range(n)])
us = run(g)
s = slots.cpu()
print("chain of %d stamp launches in one graph: %.2f us per launch by events, %.2f us per launch by the device clock"
      % (n, us / n, float(s[-1] - s[0]) / 100.0 / (n - 1)))
# the same with a 4 MB fill between stamps (a predecessor that leaves dirty lines)
small = big[: 1 << 20]
g2 = capture(lambda: [(ops.stamp(slots, i), small.add_(1.0)) for i in range(n)])
us2 = run(g2)
g3 = capture(lambda: [small.add_(1.0) for i in range(n)])
us3 = run(g3)
print("stamp + 4 MB in-place add per link: %.2f us per link; the adds alone %.2f us each" % (us2 / n, us3 / n))
