"""Data parallelism for the hot path: one process per GPU, images sharded by batch, and ONE flat
all-reduce per optimiser step over RCCL/xGMI (torch.distributed backend "nccl" on ROCm).

The reference is single-GPU by assertion (options.py:112), so this layer is an addition, designed for
the payloads of this model: the nerf step moves 433 k floats (1.7 MB: mlp_rgb + mlp_trans +
the two embedding tables, row-sparse per rank), the discriminator step 2.67 M floats (10.7 MB).  Both
are latency-bound on xGMI, so gradients are packed into one contiguous fp32 buffer and reduced with a
single collective after ALL backward calls of the step (the GAN step has three, including the R1
double backward) -- no per-backward hooks, no bucketing, nothing to overlap.  Rendering itself needs no
collective: rays are independent.
"""
from __future__ import annotations

import os
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


def init_distributed(backend: Optional[str] = None) -> tuple:
    """Initialise from the torchrun environment; returns (rank, world_size, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local)
            kw["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend, **kw)
    return rank, world, local


class _GroupCache:
    """value per process group, keyed by the group OBJECT (an id() can be handed to a new group once the old one is destroyed);
    the default group (None) has a slot of its own, dropped when the process group is destroyed and made again."""
    def __init__(self):
        import weakref
        self._by_group, self._default = weakref.WeakKeyDictionary(), None

    def _default_alive(self):
        return self._default is not None and dist.is_initialized() and self._default[0] is dist.group.WORLD

    def get(self, group):
        if group is None:
            return self._default[1] if self._default_alive() else None
        try:
            return self._by_group.get(group)
        except TypeError:                      # (an object that cannot be weakly referenced: not cached)
            return None

    def put(self, group, value):
        if group is None:
            self._default = (dist.group.WORLD, value)
            return
        try:
            self._by_group[group] = value
        except TypeError:
            pass


_RANKS_SEEN = _GroupCache()


def ranks_seen(group=None, device=None) -> int:
    """SUM all-reduce of a one per rank: how many ranks the communicator (RCCL on GPUs) actually joins -- the figure every
    multi-GPU line of bench.py / tools carries, so that the record itself says whether the collective library saw N ranks.
    Also the WARM collective of a job: RCCL creates its channels, proxy threads and internal HIP streams on the first collective,
    and whatever picks streams by measurement afterwards (trainer.distinct_queue_streams) must run behind it
    (`warm_collective_done`).  1 without a process group.  The value is cached per group object."""
    if not (dist.is_available() and dist.is_initialized()):
        return 1
    seen = _RANKS_SEEN.get(group)
    if seen is not None:
        return seen
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    one = torch.ones(1, dtype=torch.float32, device=device)
    dist.all_reduce(one, op=dist.ReduceOp.SUM, group=group)
    seen = int(round(float(one)))
    _RANKS_SEEN.put(group, seen)
    return seen


def warm_collective_done(group=None) -> bool:
    """True once `ranks_seen` has run its all-reduce on this group (or when there is no process group to warm up)."""
    return not (dist.is_available() and dist.is_initialized()) or _RANKS_SEEN.get(group) is not None


def shard_batch(n_items: int, rank: int, world: int) -> range:
    """Contiguous, balanced shard of n_items (images or rays) for this rank."""
    base, rem = divmod(n_items, world)
    start = rank * base + min(rank, rem)
    return range(start, start + base + (1 if rank < rem else 0))


class FlatGradAllReducer:
    """Averages the gradients of ``params`` across ranks with a single all-reduce on a persistent flat buffer.

    Parameters whose .grad is None on this rank (e.g. nothing touched them) contribute zeros; after
    ``reduce()`` every rank holds identical .grad tensors (allocated if they were None on a rank where
    another rank had a gradient is NOT needed: a parameter either takes part on all ranks or on none,
    which is checked once).
    """

    FLAG_WORDS = 4                      # tail of the flat buffer: step-gate words that must be decided globally

    def __init__(self, params: Iterable[torch.nn.Parameter], group=None, single_rank_collective: bool = False):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        self.group = group
        # tests only: issue the collective even in a 1-rank group (exercises RCCL under hipGraph capture on one GPU)
        self.single_rank_collective = single_rank_collective
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device if self.params else torch.device("cpu")
        self.flat = torch.zeros(n + self.FLAG_WORDS, dtype=torch.float32, device=dev)
        self.flag_tail = self.flat[n:]
        self.views = []
        off = 0
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()

    @property
    def nbytes(self) -> int:
        return self.flat.numel() * 4

    @property
    def world_size(self) -> int:
        return dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1

    # ------------------------------------------------------------------ the step in three stream-ordered pieces
    # pack | all_reduce | adopt: what a captured training step needs -- `pack` ends the graph that formed the gradients, the collective
    # is an ordinary stream-ordered call (or a node of that graph), `adopt` opens the graph that holds the optimiser launch, which
    # reads the averaged gradients straight from the flat buffer and the tail as its gate (trainer.GraphedGanTrainer, linear form).
    @property
    def gate_words(self) -> torch.Tensor:
        """The tail as int32 words (a view): non-zero <=> some rank raised that gate word in this or an earlier step (0.0f is the
        only float whose bits are zero here: the tail holds sums of 0 / 1).  What the fused optimiser launches take as ``gate``."""
        return self.flag_tail.view(torch.int32)

    @property
    def active(self) -> bool:
        """A collective is part of the step: several ranks, or a 1-rank group with the collective forced on."""
        return dist.is_available() and dist.is_initialized() and (dist.get_world_size(self.group) > 1 or self.single_rank_collective)

    def average(self) -> None:
        """pack | all_reduce | adopt in one call (the eager trainer's step): afterwards every ``.grad`` is the job's average."""
        if not self.active:
            return
        self.pack()
        self.all_reduce()
        self.adopt()

    def clear_gate(self) -> None:
        self.flag_tail.zero_()

    def pack(self, flags: Optional[torch.Tensor] = None) -> None:
        """``flat`` <- the gradients scaled by 1 / world (a parameter without one: zeros), tail <- sticky OR of ``flags`` (int32
        words) as 0 / 1 floats.  GPU tensors: ONE launch (K13 tp_grad_pack); CPU tensors (gloo tests): the same values with torch ops."""
        scale = 1.0 / self.world_size
        # (a parameter takes part on all ranks or on none -- e.g. the discriminator's `progress` never has a gradient: `adopt` leaves
        # such a parameter without one, so the optimiser makes no state for it, as in the reference)
        self._packed = [p.grad is not None for p in self.params]
        if self.flat.is_cuda:
            from . import ops
            ops.grad_pack([p.grad if p.grad is not None else (None, p.numel()) for p in self.params], self.flat, scale,
                          words=flags, tail=self.flag_tail)
            return
        with torch.no_grad():
            n = self.flag_tail.numel() if flags is None else flags.numel()
            raised = self.flag_tail[:n] != 0
            if flags is not None:
                raised = raised | (flags != 0)
            self.flag_tail[:n].copy_(raised)
            for p, v in zip(self.params, self.views):
                if p.grad is None:
                    v.zero_()
                else:
                    torch.mul(p.grad, scale, out=v)

    def all_reduce(self) -> None:
        """SUM over the ranks of the packed buffer, on the current stream (nothing without a process group, or in a 1-rank group
        unless the collective is forced)."""
        if not (dist.is_available() and dist.is_initialized()):
            return
        if dist.get_world_size(self.group) == 1 and not self.single_rank_collective:
            return
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)

    def adopt(self) -> None:
        """``.grad`` of every parameter that had a gradient at the last `pack` becomes its view of the flat buffer (the averaged
        gradient; no copy back)."""
        for p, v, had in zip(self.params, self.views, getattr(self, "_packed", [True] * len(self.params))):
            if had:
                p.grad = v

    def reduce(self, average: bool = True, flags: Optional[torch.Tensor] = None) -> None:
        """``flags`` (int32, at most FLAG_WORDS words; e.g. the sticky words of the captured step gate) travel in the tail of
        the same buffer: after the call every word is non-zero on ALL ranks if it was non-zero on ANY rank -- the gate
        decision becomes global without a second collective.  (A non-finite gradient entry cannot leak into the tail: the
        reduction is element-wise.)"""
        if not (dist.is_available() and dist.is_initialized()):
            return
        if dist.get_world_size(self.group) == 1 and not self.single_rank_collective:
            return
        world = dist.get_world_size(self.group)
        if flags is not None:
            self.flag_tail[:flags.numel()].copy_(flags != 0)
        else:
            self.flag_tail.zero_()
        have = [(p, v) for p, v in zip(self.params, self.views) if p.grad is not None]
        for p, v in zip(self.params, self.views):
            if p.grad is None:
                v.zero_()
        if have:                                           # one multi-tensor launch each way instead of one copy per tensor
            torch._foreach_copy_([v for _, v in have], [p.grad for p, _ in have])
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
        if flags is not None:
            flags.copy_(self.flag_tail[:flags.numel()] != 0)
        if average:
            self.flat.mul_(1.0 / world)
        if have:
            torch._foreach_copy_([p.grad for p, _ in have], [v for _, v in have])
        for p, v in zip(self.params, self.views):
            if p.grad is None:
                p.grad = v.clone()


def broadcast_module_state(module: torch.nn.Module, src: int = 0, group=None) -> None:
    """Make parameters and buffers (spectral-norm u/v, progress counters) identical on all ranks."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src=src, group=group)


def all_reduce_scalars(*values: torch.Tensor, group=None) -> List[torch.Tensor]:
    """SUM a handful of scalars (e.g. the masked-loss numerator / denominator) in one collective."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return list(values)
    buf = torch.stack([v.detach().reshape(()) for v in values])
    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    return list(buf.unbind(0))


def all_reduce_flags(word: torch.Tensor, group=None) -> torch.Tensor:
    """MAX over the ranks of a small int32 flag vector (range flag, non-finite loss): what makes a step-gate decision
    global.  Returns ``word`` itself when there is one rank."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return word
    word = word.contiguous()
    dist.all_reduce(word, op=dist.ReduceOp.MAX, group=group)
    return word


def setup_data_parallel(graph: torch.nn.Module, seed: int = 0, group=None) -> tuple:
    """What a data-parallel training process does once, before it builds its trainer (texpose_amd.trainer):
      1. every rank starts from rank 0's state: parameters AND buffers (spectral-norm u / v, ``progress``) are broadcast;
      2. per-rank random streams: patch scale / shift draws (torch.rand) and the in-kernel Philox jitter (seeded from
         torch.initial_seed(), texpose_amd/graph.py) must differ between ranks, so rank r seeds with ``seed + r``.
    Returns (rank, world).  Normaliser note: the photometric term divides by THIS rank's mask
    count (reference model/nerf_adapt_st_gan.py:750); averaging the ranks' gradients therefore weights every image by
    1 / (world * sum_rank(mask)) instead of 1 / sum_all(mask) -- identical when the ranks' mask counts agree, and at
    most the relative spread of the counts otherwise.  ``all_reduce_scalars`` is there for callers who want the exact
    global normaliser."""
    rank = dist.get_rank(group) if dist.is_available() and dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    broadcast_module_state(graph, src=0, group=group)
    ranks_seen(group)                                     # (the job's warm collective; cached for the lines that report it)
    torch.manual_seed(seed + rank)
    return rank, world


# per-sample entries of a collated batch (reference data/lm.py:112-159, data/lmsyn2real.py)
PER_SAMPLE_KEYS = frozenset(("idx", "image", "image_syn", "nocs_pred", "normal_pred", "obj_mask", "mask_syn", "intr", "pose",
                             "pose_init", "pose_gt", "z_near", "z_far", "frame_index", "depth_gt", "mask_visib", "mask_full",
                             "erode_mask"))                # (data/lm.py:148-150, with opt.data.erode_mask_loss)


_SHARD_WARNED = set()


def shard_training_batch(var, rank: int, world: int, per_sample_keys=PER_SAMPLE_KEYS):
    """This rank's images of a collated batch.  Only the entries named in ``per_sample_keys`` are sliced (a tensor that merely
    HAPPENS to have the batch size as its leading dimension -- a 3x3 matrix at B=3, an anchor-pose table -- is passed through);
    a per-sample key whose leading dimension is not the batch size raises."""
    B = len(var["idx"])
    sl = shard_batch(B, rank, world)
    out = type(var)()
    for k, v in var.items():
        if k in per_sample_keys and torch.is_tensor(v):
            if v.dim() == 0 or v.shape[0] != B:
                raise ValueError("shard_training_batch: %r is a per-sample entry but has shape %s at batch size %d" % (k, tuple(v.shape), B))
            out[k] = v[sl.start:sl.stop]
        else:
            # an entry this list does not know that LOOKS per-sample (image-like: >= 4 dimensions, leading one = B) would stay at
            # the full batch size on every rank; a dataset that adds such a key must name it (per_sample_keys=...)
            if torch.is_tensor(v) and world > 1 and v.dim() >= 4 and v.shape[0] == B and k not in _SHARD_WARNED:
                _SHARD_WARNED.add(k)
                import warnings
                warnings.warn("shard_training_batch: %r has the batch size as its leading dimension but is not a known per-sample "
                              "entry; it is passed to every rank UNSLICED (add it to per_sample_keys if it is per-sample)" % (k,))
            out[k] = v
    return out
