"""Checkpoint wire format of the reference (reference util.py:172-263): a torch-pickled dict
``{epoch, iter, graph=<graph.state_dict()>, optim_*/sched_* state}``.  State-dict keys and shapes of the
mirror are the reference's, so files are interchangeable in both directions; these helpers
restate the save variants (whole graph / ``children`` prefixes, ``model/<it>.ckpt`` copy) and the three restore policies
(resume, trunk only, nerf only) on a Graph (they are host-side dictionary plumbing, no kernels).
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch


def _child_state(state: Dict[str, torch.Tensor], name: str) -> Dict[str, torch.Tensor]:
    pre = name + "."
    return {k[len(pre):]: v for k, v in state.items() if k.startswith(pre)}


def make_checkpoint(graph: torch.nn.Module, epoch: Optional[int], it: Optional[int], children=None, **optim_and_sched) -> dict:
    """``optim_and_sched``: objects whose names start with ``optim`` / ``sched`` (as on the reference Model).  ``children``: a
    prefix or tuple of prefixes -- only the graph entries whose KEY STARTS WITH one of them are kept (util.py:246-249: a plain
    ``str.startswith``, so "latent_vars" keeps both latent tables and "nerf.mlp_rgb" one head)."""
    state = graph.state_dict()
    if children is not None:
        prefixes = (children,) if isinstance(children, str) else tuple(children)
        state = {k: v for k, v in state.items() if k.startswith(prefixes)}
    ck = dict(epoch=epoch, iter=it, graph=state)
    for key, obj in optim_and_sched.items():
        if key.split("_")[0] in ("optim", "sched"):
            ck[key] = obj.state_dict()
    return ck


def save_checkpoint(path: str, graph: torch.nn.Module, epoch=None, it=None, children=None, **optim_and_sched) -> None:
    torch.save(make_checkpoint(graph, epoch, it, children=children, **optim_and_sched), path)


def save_checkpoint_dir(output_path: str, graph: torch.nn.Module, epoch=None, it=None, latest: bool = False, children=None,
                        **optim_and_sched) -> str:
    """The reference's ``util.save_checkpoint(opt, model, ep, it, latest, children)`` on a directory (util.py:244-263):
    ``<output_path>/model.ckpt`` is (over)written and, unless ``latest``, copied to ``<output_path>/model/<it>.ckpt`` (the
    reference tracks the ITERATION in the copy's name; the ``model`` directory is made either way).  Returns the main file's path."""
    import os
    import shutil
    os.makedirs(os.path.join(output_path, "model"), exist_ok=True)
    main = os.path.join(output_path, "model.ckpt")
    save_checkpoint(main, graph, epoch=epoch, it=it, children=children, **optim_and_sched)
    if not latest:
        shutil.copy(main, os.path.join(output_path, "model", "{}.ckpt".format(it)))
    return main


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


def restore_pretrain_nerf(graph: torch.nn.Module, checkpoint: dict) -> Tuple[None, None]:
    """Only the ``nerf`` child is taken -- strictly, every key of it -- from the checkpoint of the real-data pre-training stage
    (``pretrain_model_real.ckpt``; util.py:225-242); the latent tables, the discriminator and the optimisers keep what they hold.
    Returns (None, None) like the reference (no epoch / iteration is resumed)."""
    for name, child in graph.named_children():
        if name != "nerf":
            continue
        sd = _child_state(checkpoint["graph"], name)
        if sd:
            child.load_state_dict(sd)
            if hasattr(child, "mark_heads_dirty"):
                child.mark_heads_dirty()
    return None, None
