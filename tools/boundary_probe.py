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

# graph -> graph: the last stamp of one replayed graph to the first stamp of the next one, (a) on the same stream, (b) the same with an
# event recorded in between, (c) the next graph on ANOTHER stream behind that event -- the boundaries of the six-graph training iteration
m = 4
slots2 = torch.zeros(2 * m, dtype=torch.int64, device=dev)
ga = capture(lambda: [ops.stamp(slots2, i) for i in range(m)])
gb = capture(lambda: [ops.stamp(slots2, m + i) for i in range(m)])
st2 = torch.cuda.Stream()
ev = torch.cuda.Event()


def pair(mode, reps=50):
    gaps = []
    for _ in range(reps):
        with torch.cuda.stream(st):
            small.add_(1.0)                              # (something in front, so that the host is ahead of the device)
            small.add_(1.0)
            ga.replay()
            if mode != "same":
                ev.record(st)
        if mode == "other":
            with torch.cuda.stream(st2):
                st2.wait_event(ev)
                gb.replay()
        else:
            with torch.cuda.stream(st):
                gb.replay()
        torch.cuda.synchronize()
        v = slots2.cpu()
        gaps.append(float(v[m] - v[m - 1]) / 100.0)
    gaps.sort()
    return gaps[len(gaps) // 2], gaps[0]


for mode, what in (("same", "same stream"), ("event", "same stream, event record between"), ("other", "other stream behind an event")):
    med, lo = pair(mode)
    print("graph -> graph, %-36s: median %.1f us, min %.1f us (last stamp of A to first stamp of B)" % (what, med, lo))
# and a plain launch behind a graph / a graph behind a plain launch
def mixed(first_graph, reps=50):
    gaps = []
    for _ in range(reps):
        with torch.cuda.stream(st):
            small.add_(1.0); small.add_(1.0)
            if first_graph:
                ga.replay(); ops.stamp(slots2, m)
            else:
                ops.stamp(slots2, m - 1); gb.replay()
        torch.cuda.synchronize()
        v = slots2.cpu()
        gaps.append(float(v[m] - v[m - 1]) / 100.0)
    gaps.sort()
    return gaps[len(gaps) // 2]
print("graph -> plain launch: %.1f us;  plain launch -> graph: %.1f us (medians)" % (mixed(True), mixed(False)))

# do dependent-launch chains on DIFFERENT hardware queues cost each other anything?  k graphs of n stamps, one per stream, replayed together
from texpose_amd.trainer import distinct_queue_streams
streams = distinct_queue_streams(dev, 3)
many = [torch.zeros(n, dtype=torch.int64, device=dev) for _ in streams]
graphs = []
for s_, sl in zip(streams, many):
    gk = torch.cuda.CUDAGraph()
    s_.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s_):
        for i in range(n):
            ops.stamp(sl, i)
    torch.cuda.synchronize()
    with torch.cuda.graph(gk, stream=s_):
        for i in range(n):
            ops.stamp(sl, i)
    graphs.append(gk)
for k in (1, 2, 3):
    reps = 20
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for s_ in streams[:k]:
        s_.wait_event(e0)
    for _ in range(reps):
        for s_, gk in zip(streams[:k], graphs[:k]):
            with torch.cuda.stream(s_):
                gk.replay()
    for s_ in streams[:k]:
        torch.cuda.current_stream().wait_stream(s_)
    e1.record()
    torch.cuda.synchronize()
    per = [float(sl[-1] - sl[0]) / 100.0 / (n - 1) for sl in many[:k]]
    print("%d chains of %d stamp launches side by side on %d queues: %.2f us per launch of a chain by events; by the device clock %s"
          % (k, n, k, e0.elapsed_time(e1) * 1e3 / reps / n, ", ".join("%.2f" % v for v in per)))
