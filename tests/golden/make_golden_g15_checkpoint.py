#!/usr/bin/env python3
"""G15: the checkpoint WIRE FORMAT as the reference itself writes and reads it.

Run in the build container only (imports /root/reference):   python tests/golden/make_golden_g15_checkpoint.py

A reference ``Model`` (graph = reference Graph with 6 latent rows, reference optimisers and scheduler exactly as
``setup_optimizer`` builds them, model/nerf_adapt_st_gan.py:62-84) is filled with the key-seeded recipe
``oracle.texpose_oracle.seeded_state`` (every tensor of the state dict is a function of its key), takes one optimiser
step per optimiser on key-seeded gradients, and is written with the reference's ``util.save_checkpoint``.  The blob is
read back with ``torch.load`` and summarised into a MANIFEST (top-level keys, per-tensor shape / dtype / sum / abs-sum of
the graph, structure and per-tensor summaries of the optimiser / scheduler state).  Two more reference graphs with
other contents are restored from the file with ``util.restore_checkpoint`` (resume) and
``util.restore_pretrain_partial_checkpoint`` (trunk only); the summaries of what they hold afterwards are recorded too.
Only the manifest (JSON, a few kB) is committed: tests rebuild the same contents from the recipe with texpose_amd, write
a checkpoint with texpose_amd.checkpoint and compare manifests, and restore from a blob they assemble in the reference's
layout.  The VGG19 stub has torchvision's public layer configuration (random weights): only keys and shapes matter here.
"""
import json
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG                                            # noqa: E402

VGG19_CFG = [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512, 512, 512, 512, "M"]


def vgg19_features():
    layers, c = [], 3
    for v in VGG19_CFG:
        if v == "M":
            layers.append(torch.nn.MaxPool2d(2, 2))
        else:
            layers += [torch.nn.Conv2d(c, v, 3, padding=1), torch.nn.ReLU(inplace=True)]
            c = v
    return torch.nn.Sequential(*layers)


def main():
    opt, camera, M, NeRF, RaySampler, FlexPatchSampler = MG._load_reference()
    sys.modules["torchvision.models"].vgg19 = lambda pretrained=True: types.SimpleNamespace(features=vgg19_features())
    sys.modules["torchvision"].models = sys.modules["torchvision.models"]
    from oracle import texpose_oracle as O
    import util
    opt.patch_size = 16
    n_train = 6

    def model(seed):
        torch.manual_seed(seed)
        m = types.SimpleNamespace()
        m.graph = M.Graph(opt)
        m.graph.latent_vars_trans = torch.nn.Embedding(n_train, opt.nerf.N_latent_trans)
        m.graph.latent_vars_light = torch.nn.Embedding(n_train, opt.nerf.N_latent_light)
        m.train_data = list(range(n_train))
        M.Model.setup_optimizer(m, opt)                       # optim_nerf, sched_nerf, optim_disc as the reference builds them
        return m

    a = model(1)
    a.graph.load_state_dict(O.seeded_state(a.graph.state_dict(), salt=11))
    for optim in (a.optim_nerf, a.optim_disc):
        O.seeded_grads(optim, salt=5)
        optim.step()
    a.sched_nerf.step()
    out = tempfile.mkdtemp()
    opt.output_path = out
    util.save_checkpoint(opt, a, ep=3, it=1234, latest=True)
    blob = torch.load(os.path.join(out, "model.ckpt"), map_location="cpu", weights_only=False)
    manifest = O.checkpoint_manifest(blob)
    manifest["gamma"] = float(opt.optim.sched.gamma)

    b = model(2)
    b.graph.load_state_dict(O.seeded_state(b.graph.state_dict(), salt=22))
    ep, it = util.restore_checkpoint(opt, b, resume=True)
    manifest["restored_resume"] = dict(epoch=ep, iter=it, graph=O.state_summary(b.graph.state_dict()),
                                       optim_nerf=O.optim_summary(b.optim_nerf.state_dict()),
                                       optim_disc=O.optim_summary(b.optim_disc.state_dict()))
    c = model(3)
    c.graph.load_state_dict(O.seeded_state(c.graph.state_dict(), salt=33))
    opt.output_root, opt.group = out, "grp"
    os.makedirs(os.path.join(out, "grp"), exist_ok=True)
    os.replace(os.path.join(out, "model.ckpt"), os.path.join(out, "grp", "pretrain_model.ckpt"))
    util.restore_pretrain_partial_checkpoint(opt, c, resume=True)
    manifest["restored_trunk_only"] = dict(graph=O.state_summary(c.graph.state_dict()))
    path = os.path.join(HERE, "g15_checkpoint_manifest.json")
    json.dump(manifest, open(path, "w"), indent=1, sort_keys=True)
    print("wrote", path, os.path.getsize(path), "bytes;", len(manifest["graph"]), "graph tensors;", sorted(manifest["top_level"]))

    # ---- G15b: the two remaining variants of util.py:225-263
    # (1) save_checkpoint(children=...) with latest=False: only the graph entries whose key starts with one of `children`, and a copy
    #     of model.ckpt under model/<it>.ckpt; (2) restore_pretrain_nerf: only the `nerf` child is taken from pretrain_model_real.ckpt.
    variants = {}
    out2 = tempfile.mkdtemp()
    opt.output_path = out2
    children = ("nerf", "latent_vars_light")
    util.save_checkpoint(opt, a, ep=None, it=77, latest=False, children=children)
    main_file, copy_file = os.path.join(out2, "model.ckpt"), os.path.join(out2, "model", "77.ckpt")
    assert os.path.exists(copy_file) and open(main_file, "rb").read() == open(copy_file, "rb").read()
    blob2 = torch.load(main_file, map_location="cpu", weights_only=False)
    variants["children"] = list(children)
    variants["saved_children"] = O.checkpoint_manifest(blob2)
    variants["copy_relpath"] = "model/77.ckpt"
    # a full checkpoint as the real-data pre-training stage leaves it
    util.save_checkpoint(opt, a, ep=3, it=1234, latest=True)
    os.replace(os.path.join(out2, "model.ckpt"), os.path.join(out2, "pretrain_model_real.ckpt"))
    d = model(4)
    d.graph.load_state_dict(O.seeded_state(d.graph.state_dict(), salt=44))
    ep, it = util.restore_pretrain_nerf(opt, d, resume=True)
    variants["restored_nerf_only"] = dict(epoch=ep, iter=it, graph=O.state_summary(d.graph.state_dict()))
    path = os.path.join(HERE, "g15b_checkpoint_variants.json")
    json.dump(variants, open(path, "w"), indent=1, sort_keys=True)
    print("wrote", path, os.path.getsize(path), "bytes;", len(variants["saved_children"]["graph"]), "graph tensors in the children file")


if __name__ == "__main__":
    main()
