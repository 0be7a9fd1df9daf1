"""Every environment switch the package reads, in ONE place.

The product path has no behaviour that depends on a mistyped or forgotten variable: all ``TP_*`` switches are parsed once, when
this module is imported, into one frozen object (``knobs.K``); nothing else in ``texpose_amd`` reads ``os.environ`` for them.  A
``TP_*`` variable that is set but not known here draws a warning at import (and at ``reload()``), so a typo cannot silently select
another schedule.  ``reload()`` parses the environment again -- for tests and tools that flip a switch between two trainers in one
process; ``override(...)`` is the same as a context manager without touching the environment.

All of them are ABLATION / DIAGNOSTIC switches: the defaults are the product, every non-default value selects a slower or
differently-structured way of computing the same numbers (the bit-identity tests in tests/test_gpu_parity.py flip them one at a time).
INTEGRATION.md lists them (``python -m texpose_amd.knobs`` prints that table).  The reference has no counterpart.
"""
from __future__ import annotations

import contextlib
import dataclasses
import os
import warnings
from typing import Dict, Tuple

_TRUE, _FALSE = ("1", "true", "yes", "on"), ("0", "false", "no", "off", "")

# (field, environment variable, default, what a NON-default value does)
_SPEC: Tuple[Tuple[str, str, object, str], ...] = (
    # ---- form of the captured training step (trainer.GraphedGanTrainer)
    ("linear_graphs", "TP_LINEAR_GRAPHS", True, "0: the iteration as ONE captured graph (two with several ranks) instead of the linear graphs on three streams"),
    ("no_linear_dp", "TP_NO_LINEAR_DP", False, "several ranks take the generic form [one graph: gradients] | eager all-reduces | [one graph: optimisers]"),
    ("split_graph", "TP_SPLIT_GRAPH", False, "generic form: gradients and optimiser steps as two graphs even without a collective (tests)"),
    ("no_branch_overlap", "TP_NO_BRANCH_OVERLAP", False, "generic form: the discriminator step on the capturing stream instead of a second one"),
    ("no_feat_branch", "TP_NO_FEAT_BRANCH", False, "generic form: the feature chain on the capturing stream instead of a third one (also switches the linear graphs off)"),
    ("no_sn_prefetch", "TP_NO_SN_PREFETCH", False, "spectral normalisations in front of each discriminator pass instead of three sets up front (also switches the linear graphs off)"),
    ("no_sn_split", "TP_NO_SN_SPLIT", False, "linear graphs: the three spectral normalisations as one graph instead of [first set] | [the other two]"),
    ("no_fused_prologue", "TP_NO_FUSED_PROLOGUE", False, "linear graphs: patch coordinates and latent rows as launches of their own instead of parts of the ray-generation launch"),
    ("no_disc_split", "TP_NO_DISC_SPLIT", False, "linear graphs: the discriminator step as one graph instead of two (no `pipeline_disc_tail`)"),
    ("no_queue_probe", "TP_NO_QUEUE_PROBE", False, "the step's three streams in creation order instead of by the measured stream -> hardware-queue probe"),
    ("pipeline_disc", "TP_PIPELINE_DISC", False, "default of GraphedGanTrainer.pipeline_disc_tail"),
    ("defer_results", "TP_DEFER_RESULTS", False, "default of GraphedGanTrainer.defer_results"),
    ("torch_rng", "TP_TORCH_RNG", False, "patch / jitter draws from torch's generator (launches of their own) instead of in-kernel Philox streams"),
    ("no_fused_adam", "TP_NO_FUSED_ADAM", False, "torch.optim.Adam(capturable) instead of the one-launch K13 tp_adam_step"),
    ("no_fused_rmsprop", "TP_NO_FUSED_RMSPROP", False, "torch.optim.RMSprop(capturable) instead of the one-launch K10 tp_rmsprop_step"),
    # ---- which launches form the discriminator / generator passes
    ("disc_autograd", "TP_DISC_AUTOGRAD", False, "discriminator step through autograd over the K7 / K9 / K11 / K14 / K15 Functions instead of the explicit schedule K16"),
    ("no_gen_schedule", "TP_NO_GEN_SCHEDULE", False, "the generator's pass through the frozen discriminator through autograd instead of disc_step.generator_pass"),
    ("no_disc_pairs", "TP_NO_DISC_PAIRS", False, "real and fake pass as separate launches instead of tp_*_pair launches"),
    ("no_disc_tail", "TP_NO_DISC_TAIL", False, "full-map convolution and head as K15 + K14 launches instead of the fused K17 tail"),
    ("no_disc_step_tail", "TP_NO_DISC_STEP_TAIL", False, "loss total, gate and RMSprop as launches of their own instead of inside tp_sn_bwd_step"),
    ("no_conv_inorm", "TP_NO_CONV_INORM", False, "InstanceNorm + LeakyReLU as a launch behind the stride-2 convolution instead of in its epilogue"),
    ("no_dgrad_inorm", "TP_NO_DGRAD_INORM", False, "InstanceNorm backward as a launch in front of the 8x8 data gradient instead of inside it"),
    ("no_conv_copy", "TP_NO_CONV_COPY", False, "the discriminator step's private patch copies by a tp_step_inputs launch instead of by the first convolution pair"),
    ("no_sn_sets", "TP_NO_SN_SETS", False, "three tp_sn_fwd calls instead of one tp_sn_fwd_sets"),
    ("skinny_dgrad_mm", "TP_SKINNY_DGRAD_MM", False, "torch.mm (rocBLAS) instead of K15's own data-gradient kernel for the skinny linear layer reached through autograd"),
    ("no_total_in_bwd", "TP_NO_TOTAL_IN_BWD", False, "the generator's loss total + gate as a launch of its own instead of a side job of tp_nerf_losses_bwd_total"),
    ("no_gather_disc", "TP_NO_GATHER_DISC", False, "the PatchGAN's stacks by tp_disc_inputs instead of by the patch gather's launch"),
    ("no_feat_chain", "TP_NO_FEAT_CHAIN", False, "feature loss through per-layer launches + autograd instead of the one-call K18 chain"),
    # ---- MLP kernels
    ("no_ray_bias", "TP_NO_RAY_BIAS", False, "evaluation renders with N % 128 == 0 take the plain f16x3 kernel instead of the ray-bias form"),
    ("wgrad_all_cus", "TP_WGRAD_ALL_CUS", False, "linear graphs: the render's weight gradient fills every CU instead of 7/8 (the discriminator step then sits it out)"),
    ("no_pack_merge", "TP_NO_PACK_MERGE", False, "f16x3 training: head stream and transposed head image packed by two launches instead of one"),
    # ---- diagnostics
    ("stamps", "TP_STAMPS", False, "one-thread launches writing the device clock at the boundaries of the captured graphs (tools/linear_timeline.py)"),
    ("extra_launches", "TP_EXTRA_LAUNCHES", "", "'G2a=10,F=10': that many one-thread launches appended to the named graphs (critical-path slope probe)"),
    ("queue_probe_verbose", "TP_QUEUE_PROBE_VERBOSE", False, "print the stream -> hardware-queue probe's sharing matrix"),
)

# read by libtexpose_amd.so itself (getenv in csrc/*): kernel-variant A/B switches, listed so that they are KNOWN names
LIBRARY_SWITCHES: Dict[str, str] = {
    "TP_FP32_CXX": "1: the compiled exact-fp32 MLP kernels instead of the generated-assembly ones (bit-identical; read per call)",
    "TP_DGRAD_CXX": "1: the compiled f16x3 data-gradient kernel instead of the generated-assembly one",
    "TP_SN_GRAD_ELEMS": "elements per workgroup of the spectral-norm backward's second launch",
    "TP_ADAM_BLOCKS": "upper bound on tp_adam_step's workgroups",
    "TP_CONV_TARGET_WGS": "split-K target of the stride-2 convolution kernels",
}
# read by bench.py / tools/* / tests only (never by the package)
TOOL_SWITCHES = ("TP_BENCH_TRAIN_EAGER", "TP_MIOPEN_FIND", "TP_NO_DEFER", "TP_NO_PIPELINE_DISC", "TP_PRE_STREAMS", "TP_SOAK_STRICT",
                 "TP_TIMELINE_ORDER", "TP_RANGE_CHECK_OFF")

Knobs = dataclasses.make_dataclass("Knobs", [(f, type(d), dataclasses.field(default=d)) for f, _e, d, _doc in _SPEC], frozen=True)
Knobs.__doc__ = "Frozen values of the package's environment switches (see the module docstring)."


def _parse(environ) -> "Knobs":
    values = {}
    for field, env, default, _doc in _SPEC:
        raw = environ.get(env)
        if raw is None:
            continue
        if isinstance(default, bool):
            low = raw.strip().lower()
            if low in _TRUE:
                values[field] = True
            elif low in _FALSE:
                values[field] = False
            else:
                warnings.warn("texpose_amd: %s=%r is neither 0 nor 1; keeping the default (%d)" % (env, raw, default))
        else:
            values[field] = type(default)(raw)
    known = {env for _f, env, _d, _doc in _SPEC} | set(LIBRARY_SWITCHES) | set(TOOL_SWITCHES)
    for name in sorted(environ):
        if name.startswith("TP_") and name not in known:
            warnings.warn("texpose_amd: environment variable %s is not a switch this package knows (python -m texpose_amd.knobs lists "
                          "them); it is ignored" % name)
    return Knobs(**values)


K = _parse(os.environ)


def reload() -> "Knobs":
    """Parse the environment again (tests / tools that flip a switch inside one process)."""
    global K
    K = _parse(os.environ)
    return K


@contextlib.contextmanager
def override(**values):
    """``with knobs.override(no_disc_pairs=True): ...`` -- the given switches for the duration of the block."""
    global K
    old = K
    K = dataclasses.replace(K, **values)
    try:
        yield K
    finally:
        K = old


def table() -> str:
    """The switches as a markdown table."""
    rows = ["| variable | default | a non-default value selects |", "|---|---|---|"]
    for _f, env, default, doc in _SPEC:
        rows.append("| `%s` | %s | %s |" % (env, int(default) if isinstance(default, bool) else repr(default), doc))
    for env, doc in LIBRARY_SWITCHES.items():
        rows.append("| `%s` (library) | unset | %s |" % (env, doc))
    return "\n".join(rows)


if __name__ == "__main__":
    print(table())
