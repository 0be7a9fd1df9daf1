"""GPU box: which streams of this process share a hardware queue (a spin kernel on A delays a fill on B iff they do).
queue_probe.py [streams made before] [candidates]"""
import sys, torch
dev = "cuda:0"
pre = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
torch.zeros(1, device=dev)
pre_streams = [torch.cuda.Stream(device=dev) for _ in range(pre)]
for st in pre_streams:
    with torch.cuda.stream(st):
        torch.zeros(1, device=dev)
streams = [torch.cuda.current_stream()] + [torch.cuda.Stream(device=dev) for _ in range(n)]
probe = torch.zeros(1, device=dev)
cycles = 1_000_000
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
e0.record(); torch.cuda._sleep(cycles); e1.record(); e1.synchronize()
print("1e6 cycles = %.3f ms" % e0.elapsed_time(e1))
cycles = int(cycles * 1.5 / e0.elapsed_time(e1))


def shares(a, b):
    torch.cuda.synchronize()
    da, db = torch.cuda.Event(), torch.cuda.Event()
    with torch.cuda.stream(a):
        torch.cuda._sleep(cycles); da.record(a)
    with torch.cuda.stream(b):
        probe.fill_(1.0); db.record(b)
    db.synchronize()
    r = da.query()
    torch.cuda.synchronize()
    return r


print("pre", pre, "rows/cols: null, s0..s%d; X = the column's fill waited for the row's spin" % (n - 1))
for a in streams:
    print("".join("-" if a is b else ("X" if shares(a, b) else ".") for b in streams))
