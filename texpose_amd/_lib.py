"""ctypes binding of libtexpose_amd.so (the C ABI declared in include/texpose_amd.h).

There is deliberately NO fallback: if the HIP library is missing or a launch fails, the
product path raises.  (The CPU oracle under oracle/ is test infrastructure only.)
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# TEXPOSE_AMD_LIB selects another build of the SAME library (e.g. the `make trace` diagnostic build); never a fallback
LIB_PATH = os.environ.get("TEXPOSE_AMD_LIB") or os.path.join(_HERE, "libtexpose_amd.so")

ABI_VERSION = 13

# Every symbol include/texpose_amd.h declares (checked by tests/test_capi_cpu.py).
SYMBOLS = (
    "tp_abi_version", "tp_last_error",
    "tp_raygen", "tp_raygen_train", "tp_aabb", "tp_sample_depth",
    "tp_mlp_packed_bytes", "tp_mlp_pack", "tp_mlp_pack_heads_f16x3", "tp_mlp_pack_host", "tp_mlp_workspace_bytes", "tp_mlp_fwd", "tp_posenc",
    "tp_mlp_saved_bytes", "tp_mlp_ray_bias_bytes", "tp_mlp_packed_t_bytes", "tp_mlp_bwd_workspace_bytes", "tp_mlp_bwd",
    "tp_composite_fwd", "tp_composite_bwd",
    "tp_patch_gather",
    "tp_eval_metrics_workspace_bytes", "tp_eval_metrics",
    "tp_sn_work_floats", "tp_sn_fwd", "tp_sn_fwd_sets", "tp_sn_bwd",
    "tp_nerf_losses_fwd", "tp_nerf_losses_bwd", "tp_nerf_losses_bwd_total",
    "tp_render_eval_workspace_bytes", "tp_render_eval",
    "tp_inorm_lrelu_fwd", "tp_inorm_lrelu_bwd", "tp_inorm_lrelu_bwd_bwd", "tp_inorm_lrelu_bwd_pair",
    "tp_rmsprop_step",
    "tp_conv4s2_workspace", "tp_conv4s2_fwd", "tp_conv4s2_dgrad", "tp_conv4s2_wgrad", "tp_conv4s2_fwd_inorm_workspace", "tp_conv4s2_fwd_inorm",
    "tp_conv4s2_fwd_inorm_pair", "tp_conv4s2_dgrad_pair", "tp_conv4s2_wgrad_pair",
    "tp_conv3s1_workspace", "tp_conv3s1_fwd", "tp_conv3s1_dgrad",
    "tp_patch_coords", "tp_bce_logits_fwd", "tp_bce_logits_bwd", "tp_feat_inputs_fwd", "tp_feat_inputs_bwd", "tp_disc_inputs", "tp_step_flags", "tp_adam_step", "tp_step_inputs", "tp_grad_pack", "tp_capture_node_count", "tp_stamp", "tp_clock_probe",
    "tp_fake_patch_bwd", "tp_feat_pair_loss_fwd", "tp_feat_pair_loss_bwd", "tp_sumsq_mean_fwd", "tp_sumsq_mean_bwd", "tp_sumsq_mean_fwd_bwd", "tp_gan_disc_losses", "tp_maxpool2_fwd", "tp_maxpool2_bwd", "tp_latent_rows_fwd",
    "tp_latent_rows_bwd", "tp_weighted_sum", "tp_weighted_sum_flags", "tp_sn_bwd_step",
    "tp_disc_head_fwd", "tp_disc_head_bwd", "tp_disc_head_bwd_bwd",
    "tp_skinny_linear_fwd", "tp_skinny_linear_wgrad", "tp_skinny_linear_dgrad",
    "tp_disc_tail_workspace_bytes", "tp_disc_tail_fwd", "tp_disc_tail_bwd", "tp_disc_tail_bwd_bwd", "tp_disc_tail_fwd_pair", "tp_disc_tail_bwd_pair",
    "tp_feat_chain_workspace", "tp_feat_chain_packed_floats", "tp_feat_chain_pack", "tp_feat_chain",
)

vp = C.c_void_p


class RaygenArgs(C.Structure):
    _fields_ = [("intr", vp), ("pose", vp), ("coords", vp), ("ray_idx", vp), ("z_near", vp),
                ("z_far", vp), ("rand", vp), ("aabb_min", C.c_float * 3), ("aabb_max", C.c_float * 3),
                ("bg_near", C.c_float), ("bg_far", C.c_float), ("valid_rect", vp), ("seed", C.c_uint64), ("offset", C.c_uint64), ("offset_dev", vp),
                ("B", C.c_int), ("R", C.c_int), ("H", C.c_int), ("W", C.c_int), ("N", C.c_int),
                ("pixel_mode", C.c_int), ("bounds_mode", C.c_int), ("jitter_mode", C.c_int), ("ndc", C.c_int), ("depth_param", C.c_int),
                ("center", vp), ("ray", vp), ("near", vp), ("far", vp), ("depth", vp)]


class PatchSamplerJob(C.Structure):
    _fields_ = [("u", vp), ("p", C.c_int), ("lattice", vp), ("lo_dev", vp), ("lo_host", C.c_float), ("span_host", C.c_float), ("hi", C.c_float),
                ("random_scale", C.c_int), ("random_shift", C.c_int), ("seed", C.c_uint64), ("counter", vp), ("coords", vp), ("scales", vp)]


class LatentRowsJob(C.Structure):
    _fields_ = [("w_trans", vp), ("w_light", vp), ("idx", vp), ("B", C.c_int), ("C_trans", C.c_int), ("C_light", C.c_int),
                ("out_trans", vp), ("out_light", vp), ("idx_copy", vp)]


class MlpWeights(C.Structure):
    _fields_ = [("feat_w", vp * 8), ("feat_b", vp * 8), ("rgb_w", vp * 4), ("rgb_b", vp * 4),
                ("trans_w", vp * 4), ("trans_b", vp * 4)]


class MlpFwdArgs(C.Structure):
    _fields_ = [("packed", vp), ("center", vp), ("ray", vp), ("depth", vp), ("points", vp),
                ("ray_unit", vp), ("lat_trans", vp), ("lat_light", vp),
                ("B", C.c_int), ("R", C.c_int), ("N", C.c_int),
                ("rgb", vp), ("density", vp), ("uncert", vp), ("saved", vp), ("workspace", vp),
                ("precision", C.c_int), ("status", vp), ("act_max", vp), ("ray_bias", vp), ("density_noise", vp)]


class MlpBwdArgs(C.Structure):
    _fields_ = [("weights", MlpWeights), ("packed_t", vp), ("repack", C.c_int), ("saved", vp), ("rgb", vp),
                ("density", vp), ("uncert", vp), ("g_rgb", vp), ("g_density", vp), ("g_uncert", vp),
                ("lat_trans", vp), ("lat_light", vp), ("B", C.c_int), ("R", C.c_int), ("N", C.c_int),
                ("g_rgb_w", vp * 4), ("g_rgb_b", vp * 4), ("g_trans_w", vp * 4), ("g_trans_b", vp * 4),
                ("g_lat_trans", vp), ("g_lat_light", vp), ("workspace", vp), ("wgrad_precision", C.c_int),
                ("dz_max_is_clear", C.c_int), ("wgrad_cus", C.c_int)]


class CompositeArgs(C.Structure):
    _fields_ = [("ray", vp), ("rgb", vp), ("density", vp), ("depth", vp), ("uncert", vp),
                ("n", C.c_int64), ("N", C.c_int), ("min_uncert", C.c_float),
                ("out_ray", vp), ("alpha_static", vp), ("alpha_transient", vp), ("prob", vp),
                ("rgb_ray", vp), ("uncert_ray", vp)]


class CompositeBwdArgs(C.Structure):
    _fields_ = [("fwd", CompositeArgs), ("g_out_ray", vp), ("g_alpha_static", vp),
                ("g_alpha_transient", vp), ("g_prob", vp), ("g_rgb", vp), ("g_density", vp),
                ("g_uncert", vp), ("g_rgb_ray", vp), ("g_uncert_ray", vp), ("g_rgb_ray2", vp), ("g_rgb_ray3", vp),
                ("g_density_add", vp)]


class PatchGatherArgs(C.Structure):
    _fields_ = [("coords", vp), ("image", vp), ("image_syn", vp), ("nocs", vp), ("normal", vp),
                ("obj_mask", vp), ("mask_syn", vp), ("B", C.c_int), ("P", C.c_int), ("H", C.c_int),
                ("W", C.c_int), ("out", vp), ("disc_rgb", vp), ("disc_real", vp), ("disc_fake", vp), ("disc_geo", C.c_int), ("pad_", C.c_int)]


class EvalMetricsArgs(C.Structure):
    _fields_ = [("rgb_static", vp), ("image", vp), ("obj_mask", vp), ("B", C.c_int), ("h", C.c_int), ("w", C.c_int),
                ("out_h", C.c_int), ("out_w", C.c_int), ("workspace", vp), ("out", vp)]


class SnWeight(C.Structure):
    _fields_ = [("weight", vp), ("u", vp), ("v", vp), ("weight_sn", vp), ("sigma", vp), ("grad_sn", vp), ("grad", vp),
                ("work", vp), ("rows", C.c_int), ("cols", C.c_int), ("u_out", vp), ("v_out", vp), ("accumulate", C.c_int32),
                ("grad_sn2", vp), ("weight_sn2", vp), ("u2", vp), ("v2", vp), ("sigma2", vp)]


SN_MAX_WEIGHTS = 8


class SnStepTail(C.Structure):
    _fields_ = [("terms", vp * 4), ("weights", C.c_float * 4), ("n_terms", C.c_int), ("word_finite", C.c_int), ("total", vp),
                ("bad", vp), ("snapshot", vp), ("n_bad", C.c_int), ("pad_", C.c_int),
                ("param", vp * SN_MAX_WEIGHTS), ("square_avg", vp * SN_MAX_WEIGHTS), ("step", vp * SN_MAX_WEIGHTS),
                ("lr_dev", vp), ("lr_host", C.c_float), ("alpha", C.c_float), ("one_minus_alpha", C.c_float), ("eps", C.c_float)]


class RmspropTensor(C.Structure):
    _fields_ = [("param", vp), ("grad", vp), ("square_avg", vp), ("numel", C.c_int64), ("step", vp)]


RMSPROP_MAX_TENSORS = 16


class AdamTensor(C.Structure):
    _fields_ = [("param", vp), ("grad", vp), ("exp_avg", vp), ("exp_avg_sq", vp), ("step", vp), ("numel", C.c_int64)]


ADAM_MAX_TENSORS = 32
GRAD_PACK_MAX_TENSORS = 32
CONV_FWD, CONV_DGRAD, CONV_WGRAD = 0, 1, 2


class DiscHeadArgs(C.Structure):
    _fields_ = [(k, vp) for k in ("z", "scale", "W1", "W2", "W3", "g_out", "c_gz", "t0", "t1", "t2", "e1", "e2", "out", "gW1", "gW2",
                                  "gW3")] + [("B", C.c_int32), ("C", C.c_int32), ("L", C.c_int32), ("H", C.c_int32), ("slope", C.c_float),
                                                 ("accumulate_gw", C.c_int32)]


DISC_TAIL_MAX_ROWS = 16


class DiscTailArgs(C.Structure):
    _fields_ = [(k, vp) for k in ("a", "W0", "scale", "W1", "W2", "W3", "g_out", "t0", "t1", "t2", "e1", "e2", "out", "gz", "c_a", "gW0",
                                  "gy2", "a2", "gW1", "gW2", "gW3", "in_xhat", "in_rstd", "in_addend", "c_z")] + [("in_P", C.c_int32)] + \
               [(k, vp) for k in ("workspace", "ticket")] + \
               [(k, C.c_int32) for k in ("M", "M2", "K", "N", "L", "H")] + [("slope", C.c_float), ("accumulate_gw", C.c_int32)]


class FeatInputsArgs(C.Structure):
    _fields_ = [("rgb", vp), ("gathered", vp), ("B", C.c_int32), ("P", C.c_int32), ("n_channels", C.c_int32),
                ("c_image", C.c_int32), ("c_image_syn", C.c_int32), ("c_mask", C.c_int32), ("c_mask_syn", C.c_int32),
                ("mean", C.c_float * 3), ("std", C.c_float * 3)]


class InormBwdArgs(C.Structure):
    _fields_ = [("xhat", vp), ("rstd", vp), ("gy", vp), ("n_inst", C.c_int64), ("hw", C.c_int32), ("slope", C.c_float), ("addend", vp), ("gx", vp)]


class FeatChainArgs(C.Structure):
    _fields_ = [("rgb", vp), ("gathered", vp), ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("n_channels", C.c_int32),
                ("c_image", C.c_int32), ("c_image_syn", C.c_int32), ("c_mask", C.c_int32), ("c_mask_syn", C.c_int32),
                ("mean", C.c_float * 3), ("std", C.c_float * 3), ("packed", vp), ("bias", vp * 7), ("w2", C.c_float), ("scale", C.c_float),
                ("loss", vp), ("g_rgb", vp), ("workspace", vp), ("workspace_floats", C.c_int64), ("counters", vp), ("n_counters", C.c_int64)]


class Conv3s1Args(C.Structure):
    _fields_ = [("inp", vp), ("w", vp), ("bias", vp), ("mask", vp), ("out", vp), ("workspace", vp), ("counters", vp),
                ("N", C.c_int32), ("C", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("Co", C.c_int32), ("relu", C.c_int32)]


class Conv4s2Args(C.Structure):
    _fields_ = [("x", vp), ("w", vp), ("gy", vp), ("out", vp), ("workspace", vp), ("counters", vp),
                ("N", C.c_int32), ("C", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("Co", C.c_int32), ("skip_out", C.c_int32),
                ("in_xhat", vp), ("in_rstd", vp), ("in_addend", vp), ("in_gx", vp), ("in_slope", C.c_float), ("pad_", C.c_int32), ("x_copy", vp)]


class NerfLossesArgs(C.Structure):
    _fields_ = [("rgb", vp), ("uncert", vp), ("density", vp), ("gathered", vp), ("B", C.c_int), ("P", C.c_int),
                ("N", C.c_int), ("workspace", vp), ("sums", vp), ("losses", vp), ("ticket", vp)]


NERF_LOSSES_MAX_BLOCKS = 1024
STEP_INPUTS_MAX_COPIES, STEP_INPUTS_MAX_SCALARS = 24, 8


class StepCopy(C.Structure):
    _fields_ = [("dst", vp), ("src", vp), ("bytes", C.c_int64)]


class RenderEvalArgs(C.Structure):
    _fields_ = [("raygen", RaygenArgs), ("packed", vp), ("lat_trans", vp), ("lat_light", vp), ("precision", C.c_int),
                ("status", vp), ("min_uncert", C.c_float), ("workspace", vp), ("out_ray", vp), ("alpha_static", vp),
                ("alpha_transient", vp), ("packed_ray_bias", C.c_int)]


class TexposeLibraryError(RuntimeError):
    pass


_lib = None


def load() -> C.CDLL:
    """Load the HIP library (after torch, so that both share one libamdhip64 runtime)."""
    global _lib
    if _lib is not None:
        return _lib
    import torch  # noqa: F401  (loads torch's bundled libamdhip64.so.7 first; same SONAME => one runtime)
    if not os.path.exists(LIB_PATH):
        raise TexposeLibraryError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  texpose_amd has no CPU or eager fallback.")
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    lib.tp_abi_version.restype = C.c_int
    lib.tp_last_error.restype = C.c_char_p
    if lib.tp_abi_version() != ABI_VERSION:
        raise TexposeLibraryError(f"ABI mismatch: library {lib.tp_abi_version()} != binding {ABI_VERSION}")

    def sig(name, argtypes, restype=C.c_int):
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = restype

    sig("tp_raygen", [C.POINTER(RaygenArgs), vp])
    sig("tp_raygen_train", [C.POINTER(RaygenArgs), C.POINTER(PatchSamplerJob), C.POINTER(LatentRowsJob), vp])
    sig("tp_aabb", [C.POINTER(C.c_float), C.POINTER(C.c_float), vp, vp, C.c_int64, vp, vp, vp, vp])
    sig("tp_sample_depth", [vp, vp, vp, C.c_int, C.c_uint64, C.c_uint64, C.c_int64, C.c_int, C.c_int, vp, vp])
    sig("tp_composite_fwd", [C.POINTER(CompositeArgs), vp])
    sig("tp_composite_bwd", [C.POINTER(CompositeBwdArgs), vp])
    sig("tp_mlp_packed_bytes", [], C.c_size_t)
    sig("tp_mlp_workspace_bytes", [C.c_int64], C.c_size_t)
    sig("tp_mlp_saved_bytes", [C.c_int64], C.c_size_t)
    sig("tp_mlp_ray_bias_bytes", [C.c_int, C.c_int], C.c_size_t)
    sig("tp_mlp_packed_t_bytes", [], C.c_size_t)
    sig("tp_mlp_bwd_workspace_bytes", [C.c_int64], C.c_size_t)
    sig("tp_mlp_bwd", [C.POINTER(MlpBwdArgs), vp])
    sig("tp_mlp_pack", [C.POINTER(MlpWeights), C.c_int, vp, vp])
    sig("tp_mlp_pack_heads_f16x3", [C.POINTER(MlpWeights), vp, vp, vp])
    sig("tp_mlp_pack_host", [C.POINTER(MlpWeights), vp])
    sig("tp_mlp_fwd", [C.POINTER(MlpFwdArgs), vp])
    sig("tp_posenc", [vp, C.c_int64, C.c_int, C.c_int, vp, vp])
    sig("tp_patch_gather", [C.POINTER(PatchGatherArgs), vp])
    sig("tp_eval_metrics_workspace_bytes", [C.c_int, C.c_int, C.c_int], C.c_int64)
    sig("tp_eval_metrics", [C.POINTER(EvalMetricsArgs), vp])
    sig("tp_sn_work_floats", [C.c_int, C.c_int], C.c_int64)
    sig("tp_sn_fwd", [C.POINTER(SnWeight), C.c_int, C.c_int, vp])
    sig("tp_sn_fwd_sets", [C.POINTER(SnWeight), C.c_int, C.c_int, vp])
    sig("tp_sn_bwd", [C.POINTER(SnWeight), C.c_int, vp])
    sig("tp_sn_bwd_step", [C.POINTER(SnWeight), C.c_int, C.POINTER(SnStepTail), vp])
    sig("tp_nerf_losses_fwd", [C.POINTER(NerfLossesArgs), vp])
    sig("tp_nerf_losses_bwd", [C.POINTER(NerfLossesArgs), vp, vp, vp, vp, vp, vp, vp])
    sig("tp_nerf_losses_bwd_total", [C.POINTER(NerfLossesArgs), vp, vp, vp, vp, vp, vp, C.POINTER(vp), C.POINTER(C.c_float), C.c_int, vp, vp, vp,
                                     C.c_int, C.c_int, C.c_int, vp, vp, vp])
    sig("tp_render_eval_workspace_bytes", [C.c_int, C.c_int, C.c_int], C.c_size_t)
    sig("tp_render_eval", [C.POINTER(RenderEvalArgs), vp])
    sig("tp_inorm_lrelu_fwd", [vp, C.c_int64, C.c_int, C.c_float, C.c_float, vp, vp, vp, vp])
    sig("tp_inorm_lrelu_bwd", [vp, vp, vp, C.c_int64, C.c_int, C.c_float, vp, vp, vp])
    sig("tp_inorm_lrelu_bwd_bwd", [vp, vp, vp, vp, C.c_int64, C.c_int, C.c_float, vp, vp, vp])
    sig("tp_rmsprop_step", [C.POINTER(RmspropTensor), C.c_int, vp, C.c_double, C.c_double, C.c_double, vp, C.c_int, vp])
    sig("tp_step_flags", [vp, vp, vp, C.c_int, C.c_int, C.c_int, vp, vp])
    sig("tp_stamp", [vp, vp])
    sig("tp_clock_probe", [vp, C.c_int, C.c_int64, vp])
    sig("tp_step_inputs", [C.POINTER(StepCopy), C.c_int, C.POINTER(vp), C.POINTER(C.c_float), C.c_int, vp, vp, C.c_int, vp])
    sig("tp_grad_pack", [C.POINTER(vp), C.POINTER(C.c_int64), C.c_int, vp, C.c_float, vp, C.c_int, vp, vp])
    sig("tp_capture_node_count", [vp], C.c_int64)
    sig("tp_adam_step", [C.POINTER(AdamTensor), C.c_int, vp, C.c_double, C.c_double, C.c_double, C.c_double, vp, C.c_int, vp, vp])
    sig("tp_conv4s2_workspace", [C.POINTER(Conv4s2Args), C.c_int, C.POINTER(C.c_int64)], C.c_int64)
    sig("tp_conv4s2_fwd_inorm_workspace", [C.POINTER(Conv4s2Args), C.POINTER(C.c_int64)], C.c_int64)
    sig("tp_conv4s2_fwd_inorm", [C.POINTER(Conv4s2Args), C.c_float, C.c_float, vp, vp, vp])
    for name in ("tp_conv4s2_fwd", "tp_conv4s2_dgrad", "tp_conv4s2_wgrad"):
        sig(name, [C.POINTER(Conv4s2Args), vp])
    sig("tp_conv4s2_fwd_inorm_pair", [C.POINTER(Conv4s2Args), vp, vp, C.POINTER(Conv4s2Args), vp, vp, C.c_float, C.c_float, vp])
    for name in ("tp_conv4s2_dgrad_pair", "tp_conv4s2_wgrad_pair"):
        sig(name, [C.POINTER(Conv4s2Args), C.POINTER(Conv4s2Args), vp])
    sig("tp_inorm_lrelu_bwd_pair", [C.POINTER(InormBwdArgs), C.POINTER(InormBwdArgs), vp])
    sig("tp_conv3s1_workspace", [C.POINTER(Conv3s1Args), C.c_int, C.POINTER(C.c_int64)], C.c_int64)
    for name in ("tp_conv3s1_fwd", "tp_conv3s1_dgrad"):
        sig(name, [C.POINTER(Conv3s1Args), vp])
    sig("tp_patch_coords", [vp, C.c_int, C.c_int, vp, vp, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int, C.c_uint64, vp, vp, vp, vp])
    sig("tp_bce_logits_fwd", [vp, C.c_int, C.c_float, vp, vp])
    sig("tp_bce_logits_bwd", [vp, C.c_int, C.c_float, vp, vp, vp])
    sig("tp_feat_chain_workspace", [C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int64)], C.c_int64)
    sig("tp_feat_chain_packed_floats", [], C.c_int64)
    sig("tp_feat_chain_pack", [C.POINTER(vp), vp, vp])
    sig("tp_feat_chain", [C.POINTER(FeatChainArgs), vp])
    sig("tp_feat_inputs_fwd", [C.POINTER(FeatInputsArgs), vp, vp])
    sig("tp_feat_inputs_bwd", [C.POINTER(FeatInputsArgs), vp, vp, vp])
    sig("tp_disc_inputs", [vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp])
    sig("tp_fake_patch_bwd", [vp, C.c_int, C.c_int, C.c_int, vp, vp])
    sig("tp_feat_pair_loss_fwd", [vp, C.c_int64, C.c_float, vp, vp])
    sig("tp_feat_pair_loss_bwd", [vp, C.c_int64, C.c_float, vp, vp, vp])
    sig("tp_sumsq_mean_fwd", [vp, C.c_int64, C.c_int, vp, vp])
    sig("tp_sumsq_mean_bwd", [vp, C.c_int64, C.c_int, vp, vp, vp])
    sig("tp_sumsq_mean_fwd_bwd", [vp, C.c_int64, C.c_int, C.c_float, vp, vp, vp])
    sig("tp_maxpool2_fwd", [vp, C.c_int64, C.c_int, C.c_int, vp, vp, vp])
    sig("tp_maxpool2_bwd", [vp, vp, C.c_int64, C.c_int, C.c_int, vp, vp])
    sig("tp_gan_disc_losses", [vp, vp, C.c_int, C.c_float, C.c_float, vp, vp, vp, vp])
    sig("tp_weighted_sum", [C.POINTER(vp), C.POINTER(C.c_float), C.c_int, vp, vp])
    sig("tp_weighted_sum_flags", [C.POINTER(vp), C.POINTER(C.c_float), C.c_int, vp, vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp])
    sig("tp_latent_rows_fwd", [vp, vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp])
    sig("tp_latent_rows_bwd", [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp])
    for name in ("tp_disc_head_fwd", "tp_disc_head_bwd", "tp_disc_head_bwd_bwd"):
        sig(name, [C.POINTER(DiscHeadArgs), vp])
    for name in ("tp_skinny_linear_fwd", "tp_skinny_linear_wgrad", "tp_skinny_linear_dgrad"):
        sig(name, [vp, vp, vp, C.c_int, C.c_int, C.c_int, vp])
    sig("tp_disc_tail_workspace_bytes", [C.c_int], C.c_size_t)
    for name in ("tp_disc_tail_fwd_pair", "tp_disc_tail_bwd_pair"):
        sig(name, [C.POINTER(DiscTailArgs), C.POINTER(DiscTailArgs), vp])
    for name in ("tp_disc_tail_fwd", "tp_disc_tail_bwd", "tp_disc_tail_bwd_bwd"):
        sig(name, [C.POINTER(DiscTailArgs), vp])
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise TexposeLibraryError(f"{what} failed (rc={rc}): {load().tp_last_error().decode()}")
