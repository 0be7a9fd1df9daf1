"""Checkpoint wire format of the reference (reference util.py:172-263): a torch-pickled dict
``{epoch, iter, graph=<graph.state_dict()>, optim_*/sched_* state}``.  State-dict keys and shapes of the
mirror are the reference's, so files are interchangeable in both directions; these helpers
restate the three restore policies on a Graph (they are host-side dictionary plumbing, no kernels).
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch


def _child_state(state: Dict[str, torch.Tensor], name: str) -> Dict[str, torch.Tensor]:
    pre = name + "."
    return {k[len(pre):]: v for k, v in state.items() if k.startswith(pre)}


def make_checkpoint(graph: torch.nn.Module, epoch: Optional[int], it: Optional[int], **optim_and_sched) -> dict:
    """``optim_and_sched``: objects whose names start with ``optim`` / ``sched`` (as on the reference Model)."""
    ck = dict(epoch=epoch, iter=it, graph=graph.state_dict())
    for key, obj in optim_and_sched.items():
        if key.split("_")[0] in ("optim", "sched"):
            ck[key] = obj.state_dict()
    return ck


def save_checkpoint(path: str, graph: torch.nn.Module, epoch=None, it=None, **optim_and_sched) -> None:
    torch.save(make_checkpoint(graph, epoch, it, **optim_and_sched), path)


def restore_checkpoint(graph: torch.nn.Module, checkpoint: dict, resume: bool = True,
                       **optim_and_sched) -> Tuple[Optional[int], Optional[int]]:
    """Per-child (possibly partial) load of ``checkpoint['graph']`` + optimiser / scheduler state (util.py:172-199)."""
    for name, child in graph.named_children():
        sd = _child_state(checkpoint["graph"], name)
        if sd:
            child.load_state_dict(sd)
    if resume:
        for key, obj in optim_and_sched.items():
            if key.split("_")[0] in ("optim", "sched") and key in checkpoint:
                obj.load_state_dict(checkpoint[key])
        return checkpoint.get("epoch"), checkpoint.get("iter")
    return None, None


def restore_pretrained_trunk(graph: torch.nn.Module, checkpoint: dict) -> int:
    """Only the frozen geometry trunk (keys containing 'mlp_feat') is taken from a pre-training checkpoint
    (util.py:202-222); returns the number of tensors loaded."""
    n = 0
    for name, child in graph.named_children():
        own = child.state_dict()
        take = {k: v for k, v in _child_state(checkpoint["graph"], name).items() if "mlp_feat" in k and k in own}
        if take:
            own.update(take)
            child.load_state_dict(own)
            n += len(take)
    return n
