"""Which streams of a process run CONCURRENTLY: the measurement the captured training step picks its three streams by
(texpose_amd.graphed_trainer; tools/queue_probe.py is the stand-alone version)."""
from __future__ import annotations

import warnings

import torch

from . import dist as tdist
from . import knobs

LAST_QUEUE_PROBE = {}        # outcome of the most recent probe (requested / concurrent / candidate indices / sharing matrix)


def distinct_queue_streams(dev, n, candidates=8, votes=3):
    """``n`` streams that run CONCURRENTLY with each other and with the current stream, found by measurement.

    The HIP runtime multiplexes all streams of a process onto a few hardware queues (4 by default), round-robin in creation order, and a
    hardware queue is in-order: two streams on one queue serialise, barrier packets included.  Which queue a new stream gets depends on
    how many streams the process made before -- the captured training step ran 850 it/s with its three streams on three queues of their
    own and 640-735 it/s when one of them shared the calling stream's queue (tools/train_bench.py with TP_PRE_STREAMS=1,2; profiles/r4).
    There is no API to ask for a stream's queue, but sharing is observable: a spin kernel on A delays a one-element fill on B if and only
    if they share a queue.  A few candidate streams are made and tested that way (each test ~1 ms, once per capture); streams that do
    not pass are simply not used.  If fewer than ``n`` concurrent ones exist (GPU_MAX_HW_QUEUES < 4) the rest are taken in creation
    order: the step is then correct as ever and slower."""
    cur = torch.cuda.current_stream(dev)
    pool = [torch.cuda.Stream(device=dev) for _ in range(candidates)]
    probe = torch.zeros(1, device=dev)
    # a spin long enough to tell: calibrate cycles -> ~1.5 ms
    cycles = 1_000_000
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(dev)
    e0.record(cur); torch.cuda._sleep(cycles); e1.record(cur); e1.synchronize()
    cycles = int(cycles * 1.5 / max(e0.elapsed_time(e1), 1e-3))

    def shares_queue(a, b):
        """True when a one-element fill on ``b`` waits for a spin kernel issued on ``a`` just before."""
        torch.cuda.synchronize(dev)
        done_a, done_b = torch.cuda.Event(), torch.cuda.Event()
        with torch.cuda.stream(a):
            torch.cuda._sleep(cycles)
            done_a.record(a)
        with torch.cuda.stream(b):
            probe.fill_(1.0)
            done_b.record(b)
        done_b.synchronize()
        shared = done_a.query()
        torch.cuda.synchronize(dev)
        return shared

    # every pair, twice: a stream gets its queue on first use and the first answers about it are not the final ones (measured: a pair
    # reported as concurrent in the pass that first touched it serialised ever after); the second pass is the one that counts
    every = [cur] + pool
    for st in pool:
        with torch.cuda.stream(st):
            probe.fill_(0.0)
    shared = [[a is not b and shares_queue(a, b) for b in every] for a in every]          # (first pass: discarded)
    # A single probe can lie in ONE direction only: a host stall of more than the spin between the two enqueues lets the spin finish
    # first and reads as "shared".  So a pair counts as sharing a queue only if the majority of `votes` probes say so.
    counts = [[0] * len(every) for _ in every]
    for _ in range(votes):
        for i, a in enumerate(every):
            for j, b in enumerate(every):
                counts[i][j] += int(a is not b and shares_queue(a, b))
    shared = [[2 * c > votes for c in row] for row in counts]
    together = lambda i, j: shared[i][j] or shared[j][i]
    picked = []
    # (Stream priorities are no lever here: with the discriminator stream, or it and the third one, at priority -1 the B=4 iteration ran
    # 420-520 it/s instead of 910 on the same box -- docs/lab-notebook-r5.md.)
    for i in range(1, len(every)):
        if len(picked) < n and not together(0, i) and not any(together(j, i) for j in picked):
            picked.append(i)
    concurrent = len(picked)
    for i in range(1, len(every)):                   # (not enough concurrent ones: fill up in creation order)
        if len(picked) < n and i not in picked:
            picked.append(i)
    chosen = [every[i] for i in picked]
    idx = [0] + picked
    report = dict(requested=n, concurrent=concurrent, candidates=[i - 1 for i in picked], votes=votes,
                  sharing=[[int(together(a, b)) if a != b else 0 for b in idx] for a in idx])
    LAST_QUEUE_PROBE.clear()
    LAST_QUEUE_PROBE.update(report)
    if concurrent < n:
        warnings.warn("texpose_amd: only %d of the %d streams of the captured training step run concurrently with each other and "
                      "with the calling stream (GPU_MAX_HW_QUEUES too small, or the process holds many streams): the step is "
                      "correct and slower (measured 640-735 instead of 850 it/s at B=4)" % (concurrent, n))
    if knobs.K.queue_probe_verbose:
        print("distinct_queue_streams:", report, "pairwise sharing after the choice:",
              [[int(shares_queue(a, b)) for b in [cur] + chosen] for a in [cur] + chosen], flush=True)
    return chosen


def streams_after_collectives(dev, n, group=None, probe=None):
    """`distinct_queue_streams` ordered BEHIND the job's first collective: in a multi-rank process RCCL makes streams (and hardware
    queues) of its own on its first collective, and a stream -> queue assignment probed before that is not the one the step runs
    with.  `dist.ranks_seen` is that collective (a no-op once it has run, or without a process group)."""
    tdist.ranks_seen(group)
    assert tdist.warm_collective_done(group)
    return (probe or distinct_queue_streams)(dev, n)
