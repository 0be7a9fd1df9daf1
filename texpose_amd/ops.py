"""Thin torch-facing wrappers over the C ABI (include/texpose_amd.h).

PyTorch is plumbing here: it owns device memory and the HIP stream; every computation on the hot
path is a hand-written gfx950 kernel in texpose_amd/csrc reached through ctypes.  All functions
require CUDA (ROCm) float32 tensors and raise if the library is unavailable -- there is no CPU or
eager fallback.
"""
from __future__ import annotations

import ctypes as C
import functools
from typing import Dict, Optional, Tuple


import torch

from . import _lib, knobs
from ._lib import (CompositeArgs, CompositeBwdArgs, MlpBwdArgs, MlpFwdArgs, MlpWeights, PatchGatherArgs, RaygenArgs,
                   check)

Tensor = torch.Tensor

PIX_COORDS, PIX_INDEX = 0, 1
BOUNDS_MAP, BOUNDS_AABB, BOUNDS_NONE = 0, 1, 2
JITTER_MID, JITTER_GIVEN, JITTER_PHILOX = 0, 1, 2
DEPTH_PARAMS = {"metric": 0, "inverse": 1}          # options nerf.depth.param -> TP_DEPTH_*
PACK_TRUNK, PACK_HEADS, PACK_ALL, PACK_F16X3, PACK_RAYBIAS = 1, 2, 3, 4, 8
MLP_FP32, MLP_F16X3 = 0, 1
PRECISIONS = {"fp32": MLP_FP32, "f16x3": MLP_F16X3}

COMPOSITE_RAY_FIELDS = (("rgb", 0, 3), ("rgb_static", 3, 6), ("rgb_transient", 6, 9), ("depth", 9, 10),
                        ("opacity", 10, 11), ("opacity_static", 11, 12), ("opacity_transient", 12, 13),
                        ("uncert", 13, 14))


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _tensors(objs):
    for o in objs:
        if isinstance(o, torch.Tensor):
            yield o
        elif isinstance(o, (list, tuple)):
            yield from _tensors(o)
        elif isinstance(o, dict):
            yield from _tensors(o.values())


def _on_tensor_device(fn):
    """The C entry points launch on the CURRENT HIP device and stream (and keep per-device kernel attributes): make the
    tensors' own device current for the call, and refuse arguments that live on different devices."""
    @functools.wraps(fn)
    def wrapped(*args, **kwargs):
        dev = None
        for t in _tensors(args + tuple(kwargs.values())):
            if t.is_cuda:
                if dev is None:
                    dev = t.device
                elif t.device != dev:
                    raise _lib.TexposeLibraryError(f"{fn.__name__}: tensors on different devices ({dev} and {t.device})")
        if dev is None or dev.index == torch.cuda.current_device():
            return fn(*args, **kwargs)
        with torch.cuda.device(dev):
            return fn(*args, **kwargs)
    return wrapped


def _f32(t: Tensor, name: str) -> Tensor:
    if not t.is_cuda:
        raise _lib.TexposeLibraryError(f"{name} must live on the GPU (texpose_amd has no CPU path)")
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def _out_like(out: Optional[Tensor], like: Tensor, shape=None) -> Tensor:
    """``out`` (checked: float32, contiguous, on ``like``'s device, of the wanted number of elements) or a fresh tensor."""
    shape = tuple(like.shape) if shape is None else tuple(shape)
    if out is None:
        return torch.empty(shape, device=like.device, dtype=torch.float32)
    n = 1
    for d in shape:
        n *= d
    if out.dtype != torch.float32 or not out.is_contiguous() or out.device != like.device or out.numel() != n:
        raise _lib.TexposeLibraryError("out= must be a contiguous float32 tensor of %s elements on %s" % (n, like.device))
    return out


def _ptr(t: Optional[Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


# ------------------------------------------------------------------------------------------ K1
@_on_tensor_device
def raygen(intr: Tensor, pose: Tensor, *, H: int, W: int, n_samples: int = 0, coords: Optional[Tensor] = None,
           ray_idx: Optional[Tensor] = None, z_near: Optional[Tensor] = None, z_far: Optional[Tensor] = None,
           aabb: Optional[Tuple[Tuple[float, float, float], Tuple[float, float, float]]] = None,
           bg_range: Tuple[float, float] = (0.0, 30.0), rand: Optional[Tensor] = None,
           jitter: int = JITTER_MID, seed: int = 0, offset: int = 0, valid_rect: Optional[Tensor] = None,
           offset_dev: Optional[Tensor] = None, ndc: bool = False, depth_param: str = "metric",
           sampler: Optional[dict] = None, rows: Optional[dict] = None):
    """Fused ray-gen + bounds + stratified depths.  Returns (center, ray, near, far, depth);
    near/far/depth are None when no bounds source is given, depth is [B,R,N].  ``offset_dev`` (int64 [1] on the device): added
    to the Philox ``offset`` inside the kernel (the step counter of a captured training step).  ``ndc``: centre / ray in normalised
    device coordinates (camera.py:325-342; the bounds still come from the metric rays, as in the reference); ``depth_param``
    'inverse': depth = 1 / (sample + 1e-8) (model/nerf_adapt_st_gan.py:699).
    Training step (tp_raygen_train): ``sampler`` (a `patch_coords(..., defer=True)` job; give its `coords` tensor as ``coords``): the launch
    draws the patch coordinates itself and fills the job's coords / scales; ``rows`` (a `latent_rows_fwd(..., defer=True)` job): extra
    workgroups of the launch gather the latent rows.  Same values as the separate launches."""
    lib = _lib.load()
    intr, pose = _f32(intr, "intr"), _f32(pose, "pose")
    B = pose.shape[0]
    a = RaygenArgs()
    if coords is not None:
        coords = _f32(coords, "coords")
        R = coords.numel() // (2 * B)
        a.pixel_mode, a.coords = PIX_COORDS, coords.data_ptr()
    else:
        ray_idx = ray_idx.to(torch.int64).contiguous()
        R = ray_idx.numel() // B
        a.pixel_mode, a.ray_idx = PIX_INDEX, ray_idx.data_ptr()
    dev = pose.device
    center = torch.empty(B, R, 3, device=dev)
    ray = torch.empty(B, R, 3, device=dev)
    near = far = depth = None
    if aabb is not None:
        a.bounds_mode = BOUNDS_AABB
        a.aabb_min = (C.c_float * 3)(*[float(v) for v in aabb[0]])
        a.aabb_max = (C.c_float * 3)(*[float(v) for v in aabb[1]])
        a.bg_near, a.bg_far = float(bg_range[0]), float(bg_range[1])
        if valid_rect is not None:                      # [B,4] (x0,y0,x1,y1): pixels outside get the fallback range
            valid_rect = _f32(valid_rect, "valid_rect")
            assert valid_rect.shape == (B, 4)
            a.valid_rect = valid_rect.data_ptr()
    elif z_near is not None:
        z_near, z_far = _f32(z_near, "z_near"), _f32(z_far, "z_far")
        assert z_near.numel() == B * H * W and z_far.numel() == B * H * W
        a.bounds_mode, a.z_near, a.z_far = BOUNDS_MAP, z_near.data_ptr(), z_far.data_ptr()
    else:
        a.bounds_mode = BOUNDS_NONE
    if a.bounds_mode != BOUNDS_NONE:
        near = torch.empty(B, R, device=dev)
        far = torch.empty(B, R, device=dev)
        a.near, a.far = near.data_ptr(), far.data_ptr()
        if n_samples > 0:
            depth = torch.empty(B, R, n_samples, device=dev)
            a.depth = depth.data_ptr()
    if rand is not None:
        rand = _f32(rand, "rand")
        assert rand.numel() == B * R * n_samples
        jitter = JITTER_GIVEN
        a.rand = rand.data_ptr()
    a.jitter_mode, a.seed, a.offset = jitter, seed, offset
    a.ndc, a.depth_param = int(bool(ndc)), DEPTH_PARAMS[depth_param]
    if offset_dev is not None:
        if offset_dev.dtype != torch.int64 or offset_dev.numel() != 1 or offset_dev.device != dev:
            raise ValueError("raygen: offset_dev must be one int64 word on the device of the inputs")
        a.offset_dev = offset_dev.data_ptr()
    a.intr, a.pose = intr.data_ptr(), pose.data_ptr()
    a.B, a.R, a.H, a.W, a.N = B, R, H, W, n_samples
    a.center, a.ray = center.data_ptr(), ray.data_ptr()
    if sampler is None and rows is None:
        check(lib.tp_raygen(C.byref(a), _stream()), "tp_raygen")
        return center, ray, near, far, depth
    sj = rj = None
    if sampler is not None:
        if coords is None or coords.data_ptr() != sampler["coords"].data_ptr() or sampler["B"] != B:
            raise ValueError("raygen: a sampler job fills ITS coords tensor -- pass that tensor as coords")
        sj = _lib.PatchSamplerJob()
        lo = sampler["lo"]
        sj.u, sj.p, sj.lattice = _ptr(sampler["u"]), sampler["p"], sampler["lattice"].data_ptr()
        sj.lo_dev = lo.data_ptr() if torch.is_tensor(lo) else None
        sj.lo_host = 0.0 if torch.is_tensor(lo) else float(lo)
        sj.span_host, sj.hi = float(sampler["hi"]) - sj.lo_host, float(sampler["hi"])
        sj.random_scale, sj.random_shift = int(bool(sampler["random_scale"])), int(bool(sampler["random_shift"]))
        sj.seed, sj.counter = int(sampler["seed"]) & (2 ** 64 - 1), _ptr(sampler["counter"])
        sj.coords, sj.scales = sampler["coords"].data_ptr(), sampler["scales"].data_ptr()
    if rows is not None:
        rj = _lib.LatentRowsJob()
        rj.w_trans, rj.w_light, rj.idx = rows["w_trans"].data_ptr(), rows["w_light"].data_ptr(), rows["idx"].data_ptr()
        rj.B, rj.C_trans, rj.C_light = rows["idx"].numel(), rows["w_trans"].shape[1], rows["w_light"].shape[1]
        rj.out_trans, rj.out_light, rj.idx_copy = rows["out_trans"].data_ptr(), rows["out_light"].data_ptr(), _ptr(rows["idx_copy"])
    check(lib.tp_raygen_train(C.byref(a), None if sj is None else C.byref(sj), None if rj is None else C.byref(rj), _stream()), "tp_raygen_train")
    return center, ray, near, far, depth


@_on_tensor_device
def aabb_intersect(aabb_min, aabb_max, o: Tensor, d: Tensor):
    lib = _lib.load()
    o, d = _f32(o, "ray_o"), _f32(d, "ray_d")
    n = o.numel() // 3
    tn = torch.empty(o.shape[:-1], device=o.device)
    tf = torch.empty_like(tn)
    ok = torch.empty(o.shape[:-1], device=o.device, dtype=torch.uint8)
    lo = (C.c_float * 3)(*[float(v) for v in torch.as_tensor(aabb_min).flatten().tolist()])
    hi = (C.c_float * 3)(*[float(v) for v in torch.as_tensor(aabb_max).flatten().tolist()])
    check(lib.tp_aabb(lo, hi, o.data_ptr(), d.data_ptr(), n, tn.data_ptr(), tf.data_ptr(), ok.data_ptr(), _stream()),
          "tp_aabb")
    return tn, tf, ok.bool()


@_on_tensor_device
def sample_depth(near: Tensor, far: Tensor, n_samples: int, rand: Optional[Tensor] = None, jitter: int = JITTER_MID,
                 seed: int = 0, offset: int = 0, depth_param: str = "metric") -> Tensor:
    lib = _lib.load()
    near, far = _f32(near, "near"), _f32(far, "far")
    if rand is not None:
        rand, jitter = _f32(rand, "rand"), JITTER_GIVEN
    depth = torch.empty(*near.shape, n_samples, device=near.device)
    check(lib.tp_sample_depth(near.data_ptr(), far.data_ptr(), _ptr(rand), jitter, seed, offset, near.numel(),
                              n_samples, DEPTH_PARAMS[depth_param], depth.data_ptr(), _stream()), "tp_sample_depth")
    return depth


# ------------------------------------------------------------------------------------------ K2
def packed_bytes() -> int:
    return int(_lib.load().tp_mlp_packed_bytes())


@_on_tensor_device
def pack_weights(state: Dict[str, Tensor], packed: Optional[Tensor] = None, parts: int = PACK_ALL,
                 prefix: str = "", precision: str = "fp32", ray_bias: bool = False) -> Tensor:
    """state: reference state-dict style mapping (``mlp_feat.0.weight`` ...) of CUDA tensors.
    precision 'f16x3' builds the split-fp16 stream for the fast forward (same size); ``ray_bias``: its variant for
    mlp_forward(..., ray_bias=True) (tp_mlp_fwd_args.ray_bias)."""
    lib = _lib.load()
    w = MlpWeights()
    keep = []

    def put(arr_w, arr_b, name, n):
        for i in range(n):
            wt, bt = _f32(state[f"{prefix}{name}.{i}.weight"], name), _f32(state[f"{prefix}{name}.{i}.bias"], name)
            keep.extend((wt, bt))
            arr_w[i], arr_b[i] = wt.data_ptr(), bt.data_ptr()

    if parts & PACK_TRUNK:
        put(w.feat_w, w.feat_b, "mlp_feat", 8)
    if parts & PACK_HEADS:
        put(w.rgb_w, w.rgb_b, "mlp_rgb", 4)
        put(w.trans_w, w.trans_b, "mlp_trans", 4)
    dev = keep[0].device
    if packed is None:
        packed = torch.empty(packed_bytes() // 4, device=dev)
    flags = parts | (PACK_F16X3 if PRECISIONS[precision] == MLP_F16X3 else 0) | (PACK_RAYBIAS if ray_bias else 0)
    check(lib.tp_mlp_pack(C.byref(w), flags, packed.data_ptr(), _stream()), "tp_mlp_pack")
    return packed


def packed_t_bytes() -> int:
    return int(_lib.load().tp_mlp_packed_t_bytes())


@_on_tensor_device
def pack_heads_train(state: Dict[str, Tensor], packed: Tensor, packed_t: Optional[Tensor], prefix: str = "") -> None:
    """Training with the f16x3 kernels: the head part of the forward stream ``packed`` (chunks + biases) and the transposed image
    ``packed_t`` of the data-gradient kernel from the head weights in ONE launch (tp_mlp_pack_heads_f16x3) -- what
    pack_weights(PACK_HEADS, 'f16x3') and the repack of mlp_backward do in three."""
    lib = _lib.load()
    w = MlpWeights()
    keep = []
    for name, arr_w, arr_b in (("mlp_rgb", w.rgb_w, w.rgb_b), ("mlp_trans", w.trans_w, w.trans_b)):
        for i in range(4):
            wt, bt = _f32(state[f"{prefix}{name}.{i}.weight"], name), _f32(state[f"{prefix}{name}.{i}.bias"], name)
            keep.extend((wt, bt))
            arr_w[i], arr_b[i] = wt.data_ptr(), bt.data_ptr()
    check(lib.tp_mlp_pack_heads_f16x3(C.byref(w), packed.data_ptr(), _ptr(packed_t), _stream()), "tp_mlp_pack_heads_f16x3")


_workspaces: Dict[Tuple[int, int], Tensor] = {}


def _workspace(n_samples: int, dev: torch.device) -> Tensor:
    need = int(_lib.load().tp_mlp_workspace_bytes(n_samples)) // 4
    key = (dev.index or 0, torch.cuda.current_stream().cuda_stream)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, device=dev)
        _workspaces[key] = ws
    return ws


_status_words: Dict[int, Tensor] = {}
RANGE_MESSAGE = ("f16x3 MLP: an activation exceeded the fp16 range (6e4); render with precision='fp32' "
                 "(Graph.render_by_slices and the trainers do that by themselves)")


def mlp_status(device) -> Tensor:
    """int32 device word; bit 0 is raised by the f16x3 forward if an activation left the fp16 range."""
    key = torch.device(device).index or 0
    if key not in _status_words:
        _status_words[key] = torch.zeros(1, dtype=torch.int32, device=device)
    return _status_words[key]


_act_max_words: Dict[int, Tensor] = {}
_track_act_max = False


def track_activation_max(enable: bool = True) -> None:
    """Diagnostics: let every following f16x3 forward fold its largest hidden activation into a device word
    (tp_mlp_fwd_args.act_max; one atomic per wave and tile).  Off by default."""
    global _track_act_max
    _track_act_max = bool(enable)


def take_activation_max(device) -> float:
    """Largest hidden activation seen by the f16x3 forwards since the last take (blocking read, then cleared).  The range
    guard fires at 6e4."""
    key = torch.device(device).index or 0
    word = _act_max_words.get(key)
    if word is None:
        return 0.0
    value = float(word.view(torch.float32).item())
    word.zero_()
    return value


def _act_max_ptr(dev):
    if not _track_act_max:
        return None
    key = dev.index or 0
    if key not in _act_max_words:
        _act_max_words[key] = torch.zeros(1, dtype=torch.int32, device=dev)
    return _act_max_words[key].data_ptr()


def take_mlp_status(device) -> int:
    """Blocking read-and-clear of the status word: what was raised since the last take.  One host sync."""
    word = mlp_status(device)
    value = int(word.item())
    if value:
        word.zero_()
    return value


def check_mlp_status(device) -> None:
    """Host-synchronising check of the f16x3 range flag; a reported violation is cleared (later renders start clean)."""
    if take_mlp_status(device) & 1:
        raise _lib.TexposeLibraryError(RANGE_MESSAGE)


_status_polls: Dict[int, tuple] = {}


def poll_mlp_status(device, raise_on_flag: bool = True) -> bool:
    """Non-blocking variant: looks at the copy requested by the PREVIOUS poll if it has completed and queues a new
    asynchronous copy of the flag.  A seen violation is cleared on the device (queued on the current stream) and
    either raised or returned as True."""
    key = torch.device(device).index or 0
    prev = _status_polls.get(key)
    seen = False
    if prev is not None and prev[1].query():
        seen = bool(int(prev[0][0]) & 1)
        prev = None
        _status_polls.pop(key, None)
        if seen:
            mlp_status(device).zero_()
            if raise_on_flag:
                raise _lib.TexposeLibraryError(RANGE_MESSAGE)
    if prev is None:
        host = torch.empty(1, dtype=torch.int32, pin_memory=True)
        host.copy_(mlp_status(device), non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        _status_polls[key] = (host, ev)
    return seen


@_on_tensor_device
def mlp_forward(packed: Tensor, lat_trans: Tensor, lat_light: Tensor, *, center: Optional[Tensor] = None,
                ray: Optional[Tensor] = None, depth: Optional[Tensor] = None, points: Optional[Tensor] = None,
                ray_unit: Optional[Tensor] = None, save: bool = False, precision: str = "fp32", ray_bias: bool = False,
                saved_out: Optional[Tensor] = None, density_noise: Optional[Tensor] = None):
    """Returns rgb [B,R,N,3,2], density [B,R,N,2], uncert [B,R,N,1] (+ saved activations if save).
    ``density_noise`` [B,R,N]: added to the static density's pre-activation (reference nerf.density_noise_reg, train mode).
    ``packed`` must have been built with the same ``precision`` (and the same ``ray_bias``, see `ray_bias_applies`).
    ``saved_out``: caller-provided record buffer (tp_mlp_saved_bytes floats; its last tile's part zeroed when B*R*N % 128)."""
    lib = _lib.load()
    a = MlpFwdArgs()
    if center is not None:
        center, ray, depth = _f32(center, "center"), _f32(ray, "ray"), _f32(depth, "depth")
        B, R = center.shape[0], center.shape[1]
        N = depth.numel() // (B * R)
        a.center, a.ray, a.depth = center.data_ptr(), ray.data_ptr(), depth.data_ptr()
    else:
        points, ray_unit = _f32(points, "points"), _f32(ray_unit, "ray_unit")
        B, R, N = points.shape[0], points.shape[1], points.shape[2]
        a.points, a.ray_unit = points.data_ptr(), ray_unit.data_ptr()
    lat_trans, lat_light = _f32(lat_trans, "lat_trans"), _f32(lat_light, "lat_light")
    assert lat_trans.shape == (B, 16) and lat_light.shape == (B, 48), (lat_trans.shape, lat_light.shape)
    dev = lat_trans.device
    S = B * R * N
    rgb = torch.empty(B, R, N, 3, 2, device=dev)
    density = torch.empty(B, R, N, 2, device=dev)
    uncert = torch.empty(B, R, N, 1, device=dev)
    saved = None
    if save and saved_out is not None:
        assert saved_out.numel() == int(lib.tp_mlp_saved_bytes(S)) // 4 and saved_out.is_contiguous() and saved_out.dtype == torch.float32
        saved = saved_out
    elif save:
        saved = torch.empty(int(lib.tp_mlp_saved_bytes(S)) // 4, device=dev)
        if S % 128:        # the weight-gradient GEMM contracts whole 32-sample groups: padding must be zero
            saved[-(int(lib.tp_mlp_saved_bytes(128)) // 4):].zero_()
    ws = _workspace(S, dev)
    a.packed, a.lat_trans, a.lat_light = packed.data_ptr(), lat_trans.data_ptr(), lat_light.data_ptr()
    a.B, a.R, a.N = B, R, N
    a.rgb, a.density, a.uncert = rgb.data_ptr(), density.data_ptr(), uncert.data_ptr()
    a.saved, a.workspace = _ptr(saved), ws.data_ptr()
    a.precision = PRECISIONS[precision]
    if a.precision == MLP_F16X3:
        a.status = mlp_status(dev).data_ptr()
        a.act_max = _act_max_ptr(dev)
    rb = None
    if ray_bias:
        assert ray_bias_applies(precision, N, save, center is not None), "mlp_forward: ray_bias outside the configuration it covers"
        rb = torch.empty(int(lib.tp_mlp_ray_bias_bytes(B, R)) // 4, device=dev)
        a.ray_bias = rb.data_ptr()
    if density_noise is not None:
        density_noise = _f32(density_noise, "density_noise")
        if density_noise.numel() != B * R * N or density_noise.device != dev:
            raise _lib.TexposeLibraryError("mlp_forward: density_noise must hold one value per sample, on the render's device")
        a.density_noise = density_noise.data_ptr()
    check(lib.tp_mlp_fwd(C.byref(a), _stream()), "tp_mlp_fwd")
    return (rgb, density, uncert, saved) if save else (rgb, density, uncert)


def ray_bias_applies(precision: str, n_samples_per_ray: int, save: bool, center_form: bool) -> bool:
    """The f16x3 forward can take the ray-constant inputs of mlp_rgb.0 / mlp_trans.0 as a per-ray bias (tp_mlp_fwd_args.ray_bias)
    when no activation record is written, rays come as (center, ray, depth) and every 128-sample tile lies inside one ray.
    TP_NO_RAY_BIAS=1 switches it off (same-box A/B)."""
    return (precision == "f16x3" and not save and center_form and n_samples_per_ray % 128 == 0
            and not knobs.K.no_ray_bias)


_bwd_scratch: Dict[Tuple[int, int], Dict[str, Tensor]] = {}


@_on_tensor_device
def mlp_backward(nerf, lat_trans: Tensor, lat_light: Tensor, saved: Tensor, rgb: Tensor, density: Tensor,
                 uncert: Tensor, g_rgb: Optional[Tensor], g_density: Optional[Tensor], g_uncert: Optional[Tensor],
                 wgrad_precision: str = "fp32"):
    """Gradients of the two heads and the latent rows.  Returns dict(params=[...] in the order of
    ``nerf.head_parameters()``, lat_trans=[B,16], lat_light=[B,48]).  ``wgrad_precision='f16x3'`` runs the
    weight-gradient GEMM as split-fp16 products (fp32-grade); only for records of a range-checked f16x3 forward."""
    lib = _lib.load()
    B, R, N = rgb.shape[0], rgb.shape[1], rgb.shape[2]
    if B > 32:
        raise _lib.TexposeLibraryError("tp_mlp_bwd handles at most 32 images per call")
    dev = rgb.device
    S = B * R * N
    z = lambda like: torch.zeros_like(like)
    g_rgb = z(rgb) if g_rgb is None else _f32(g_rgb, "g_rgb")
    g_density = z(density) if g_density is None else _f32(g_density, "g_density")
    g_uncert = z(uncert) if g_uncert is None else _f32(g_uncert, "g_uncert")
    names, params = zip(*nerf.head_parameters())
    grads = [torch.empty_like(p) for p in params]
    by_name = dict(zip(names, zip(params, grads)))
    a = MlpBwdArgs()
    keep = []
    for i in range(4):
        for head, wf, gwf, gbf in (("mlp_rgb", a.weights.rgb_w, a.g_rgb_w, a.g_rgb_b),
                                   ("mlp_trans", a.weights.trans_w, a.g_trans_w, a.g_trans_b)):
            w, gw = by_name[f"{head}.{i}.weight"]
            _, gb = by_name[f"{head}.{i}.bias"]
            w = _f32(w.detach(), "weight")
            keep.append(w)
            wf[i], gwf[i], gbf[i] = w.data_ptr(), gw.data_ptr(), gb.data_ptr()
    key = (dev.index or 0, torch.cuda.current_stream().cuda_stream)
    sc = _bwd_scratch.setdefault(key, {})
    need = int(lib.tp_mlp_bwd_workspace_bytes(S)) // 4
    if "ws" not in sc or sc["ws"].numel() < need:
        sc["ws"] = torch.empty(need, device=dev)
        sc.pop("clear_for", None)
    lat_trans, lat_light = _f32(lat_trans.detach(), "lat_trans"), _f32(lat_light.detach(), "lat_light")
    g_lt, g_ll = torch.empty(B, 16, device=dev), torch.empty(B, 48, device=dev)
    # the transposed weight image: the one the forward's pack launch of this step wrote (NeRF.packed_weights, f16x3 training), else
    # rebuilt by this call
    pre = nerf.packed_t_current() if (wgrad_precision == "f16x3" and hasattr(nerf, "packed_t_current")) else None
    if pre is not None:
        a.packed_t, a.repack = pre.data_ptr(), 0
    else:
        if "packed_t" not in sc:
            sc["packed_t"] = torch.empty(int(lib.tp_mlp_packed_t_bytes()) // 4, device=dev)
        a.packed_t, a.repack = sc["packed_t"].data_ptr(), 1
    # the scale word inside the workspace is cleared by the last kernel of every call: no memset when the previous call used this
    # workspace with this size
    # The mark says "the last EXECUTED call on this workspace ran to its end".  An eager call drops it before launching and sets it
    # only after tp_mlp_bwd has returned without error (a failure after the dgrad kernel leaves the word dirty: the next call
    # memsets).  A call that is being captured into a hipGraph executes nothing now: it may rely on the mark (at replay time the
    # word is clean -- the eager call or the replay before it cleared it) but must not touch it, so that a capture which is
    # discarded without a replay changes nothing for the next eager call.
    capturing = bool(torch.cuda.is_current_stream_capturing())
    clear_key = (sc["ws"].data_ptr(), S, wgrad_precision)
    a.dz_max_is_clear = 1 if sc.get("clear_for") == clear_key else 0
    if not capturing:
        sc.pop("clear_for", None)
    a.saved, a.rgb, a.density, a.uncert = saved.data_ptr(), rgb.data_ptr(), density.data_ptr(), uncert.data_ptr()
    a.g_rgb, a.g_density, a.g_uncert = g_rgb.data_ptr(), g_density.data_ptr(), g_uncert.data_ptr()
    a.lat_trans, a.lat_light = lat_trans.data_ptr(), lat_light.data_ptr()
    a.B, a.R, a.N = B, R, N
    a.g_lat_trans, a.g_lat_light, a.workspace = g_lt.data_ptr(), g_ll.data_ptr(), sc["ws"].data_ptr()
    a.wgrad_precision = PRECISIONS[wgrad_precision]
    # (a caller that runs other streams beside the backward -- the captured GAN iteration -- leaves them a share of the device for the
    # length of the weight gradient: NeRF.wgrad_cus, 0 = all)
    a.wgrad_cus = int(getattr(nerf, "wgrad_cus", 0) or 0)
    check(lib.tp_mlp_bwd(C.byref(a), _stream()), "tp_mlp_bwd")
    if not capturing:
        sc["clear_for"] = clear_key
    return dict(params=grads, lat_trans=g_lt, lat_light=g_ll)


@_on_tensor_device
def posenc(x: Tensor, L: int) -> Tensor:
    lib = _lib.load()
    x = _f32(x, "x")
    Cn = x.shape[-1]
    out = torch.empty(*x.shape[:-1], 2 * Cn * L, device=x.device)
    check(lib.tp_posenc(x.data_ptr(), x.numel() // Cn, Cn, L, out.data_ptr(), _stream()), "tp_posenc")
    return out


# ------------------------------------------------------------------------------------------ K4
def _composite_args(ray, rgb, density, depth, uncert, min_uncert) -> Tuple[CompositeArgs, tuple]:
    ray, rgb, density = _f32(ray, "ray"), _f32(rgb, "rgb_samples"), _f32(density, "density_samples")
    depth, uncert = _f32(depth, "depth_samples"), _f32(uncert, "uncert_samples")
    n = ray.numel() // 3
    N = depth.numel() // n
    a = CompositeArgs()
    a.ray, a.rgb, a.density, a.depth, a.uncert = (ray.data_ptr(), rgb.data_ptr(), density.data_ptr(),
                                                  depth.data_ptr(), uncert.data_ptr())
    a.n, a.N, a.min_uncert = n, N, float(min_uncert)
    return a, (ray, rgb, density, depth, uncert)


@_on_tensor_device
def composite_fwd(ray, rgb, density, depth, uncert, min_uncert: float = 0.05, per_sample: bool = True,
                  want_prob: bool = True, compact: bool = False):
    """-> out_ray [..,14], alpha_static, alpha_transient, prob ([..,N] or None); with ``compact`` also (rgb_ray [..,3],
    uncert_ray [..,1]): contiguous copies of out_ray[..., 0:3] / [..., 13:14] written by the same launch (the losses, the
    feature network and the PatchGAN consume these two; slices of out_ray would need a copy each and a slice-backward)."""
    lib = _lib.load()
    a, keep = _composite_args(ray, rgb, density, depth, uncert, min_uncert)
    lead = keep[0].shape[:-1]
    dev = keep[0].device
    out = torch.empty(*lead, 14, device=dev)
    rgb_ray = unc_ray = None
    if compact:
        rgb_ray, unc_ray = torch.empty(*lead, 3, device=dev), torch.empty(*lead, 1, device=dev)
        a.rgb_ray, a.uncert_ray = rgb_ray.data_ptr(), unc_ray.data_ptr()
    a_s = torch.empty(*lead, a.N, device=dev) if per_sample else None
    a_t = torch.empty(*lead, a.N, device=dev) if per_sample else None
    prob = torch.empty(*lead, a.N, device=dev) if want_prob else None
    a.out_ray, a.alpha_static, a.alpha_transient, a.prob = out.data_ptr(), _ptr(a_s), _ptr(a_t), _ptr(prob)
    check(lib.tp_composite_fwd(C.byref(a), _stream()), "tp_composite_fwd")
    return (out, a_s, a_t, prob, rgb_ray, unc_ray) if compact else (out, a_s, a_t, prob)


@_on_tensor_device
def composite_bwd(ray, rgb, density, depth, uncert, g_out: Optional[Tensor], g_alpha_s: Optional[Tensor] = None,
                  g_alpha_t: Optional[Tensor] = None, g_prob: Optional[Tensor] = None, min_uncert: float = 0.05,
                  g_rgb_ray: Optional[Tensor] = None, g_uncert_ray: Optional[Tensor] = None, g_rgb_ray2: Optional[Tensor] = None,
                  g_rgb_ray3: Optional[Tensor] = None, g_density_add: Optional[Tensor] = None):
    """``g_out`` [..,14] may be None when the cotangent arrives through ``g_rgb_ray`` [..,3] / ``g_uncert_ray`` [..,1] (they
    are ADDED to columns 0..2 / 13 of g_out inside the kernel; ``g_rgb_ray2`` / ``g_rgb_ray3`` likewise).  ``g_density_add``
    [..,N,2] is added to the returned density gradient."""
    lib = _lib.load()
    b = CompositeBwdArgs()
    fa, keep = _composite_args(ray, rgb, density, depth, uncert, min_uncert)
    b.fwd = fa
    # (every cotangent is optional: the kernel reads a missing one as zero -- no zero g_out is made up here)
    g_out = None if g_out is None else _f32(g_out, "g_out")
    g_rgb_ray = None if g_rgb_ray is None else _f32(g_rgb_ray, "g_rgb_ray")
    g_uncert_ray = None if g_uncert_ray is None else _f32(g_uncert_ray, "g_uncert_ray")
    opt = [None if g is None else _f32(g, "g") for g in (g_alpha_s, g_alpha_t, g_prob)]
    g_rgb, g_den, g_unc = torch.empty_like(keep[1]), torch.empty_like(keep[2]), torch.empty_like(keep[4])
    b.g_out_ray, b.g_rgb_ray, b.g_uncert_ray = _ptr(g_out), _ptr(g_rgb_ray), _ptr(g_uncert_ray)
    extra = [None if g is None else _f32(g, "g") for g in (g_rgb_ray2, g_rgb_ray3, g_density_add)]
    if extra[2] is not None and extra[2].numel() != keep[2].numel():
        raise ValueError("composite_bwd: g_density_add must have the shape of density")
    b.g_rgb_ray2, b.g_rgb_ray3, b.g_density_add = _ptr(extra[0]), _ptr(extra[1]), _ptr(extra[2])
    b.g_alpha_static, b.g_alpha_transient, b.g_prob = _ptr(opt[0]), _ptr(opt[1]), _ptr(opt[2])
    b.g_rgb, b.g_density, b.g_uncert = g_rgb.data_ptr(), g_den.data_ptr(), g_unc.data_ptr()
    check(lib.tp_composite_bwd(C.byref(b), _stream()), "tp_composite_bwd")
    return g_rgb, g_den, g_unc


# ------------------------------------------------------------------------------------------ K5
@_on_tensor_device
def patch_gather(coords: Tensor, image: Tensor, image_syn: Tensor, nocs: Tensor, normal: Tensor, obj_mask: Tensor,
                 mask_syn: Tensor, disc_rgb: Optional[Tensor] = None, disc_geo: bool = False):
    """-> [B,14,p,p]: image3, image_syn3, nocs3*mask_syn, normal3*mask_syn, mask, mask_syn.
    ``disc_rgb`` [B,P,3] (the rendered colours): -> (that, real stack [2B,nc,p,p], fake [B,nc,p,p]) -- the PatchGAN's input stacks of
    the same pixels from the same launch (`disc_inputs(disc_rgb, gathered, ..., stacked=True)`, bit for bit)."""
    lib = _lib.load()
    coords = _f32(coords, "coords")
    B, ph, pw, _ = coords.shape
    ts = [_f32(t, "image") for t in (image, image_syn, nocs, normal, obj_mask, mask_syn)]
    H, W = ts[0].shape[-2:]
    out = torch.empty(B, 14, ph, pw, device=coords.device)
    a = PatchGatherArgs()
    a.coords = coords.data_ptr()
    a.image, a.image_syn, a.nocs, a.normal, a.obj_mask, a.mask_syn = [t.data_ptr() for t in ts]
    a.B, a.P, a.H, a.W, a.out = B, ph * pw, H, W, out.data_ptr()
    if disc_rgb is None:
        check(lib.tp_patch_gather(C.byref(a), _stream()), "tp_patch_gather")
        return out
    rgb = _f32(disc_rgb.detach(), "disc_rgb")
    if rgb.numel() != B * ph * pw * 3:
        raise ValueError("patch_gather: disc_rgb [B,P,3] expected")
    nc = 9 if disc_geo else 3
    real, fake = torch.empty(2 * B, nc, ph, pw, device=coords.device), torch.empty(B, nc, ph, pw, device=coords.device)
    a.disc_rgb, a.disc_real, a.disc_fake, a.disc_geo = rgb.data_ptr(), real.data_ptr(), fake.data_ptr(), int(bool(disc_geo))
    check(lib.tp_patch_gather(C.byref(a), _stream()), "tp_patch_gather")
    return out, real, fake


@_on_tensor_device
def eval_metrics(rgb_static: Tensor, image: Tensor, obj_mask: Tensor, H: int, W: int, out_hw=None):
    """PSNR / SSIM of the static render against the masked image (reference evaluate_full, :340-362).
    rgb_static [B,H*W,3], image [B,3,H,W], obj_mask [B,H,W]; ``out_hw`` = (480, 640) reproduces the resize the
    reference applies to non-crop data.  Returns (psnr, ssim, mse) as 0-dim fp64 device tensors (no host sync)."""
    lib = _lib.load()
    rgb_static, image, obj_mask = _f32(rgb_static, "rgb_static"), _f32(image, "image"), _f32(obj_mask, "obj_mask")
    B = image.shape[0]
    if rgb_static.numel() != B * H * W * 3 or image.shape[1:] != (3, H, W) or obj_mask.numel() != B * H * W:
        raise ValueError("eval_metrics: rgb_static [B,H*W,3], image [B,3,H,W], obj_mask [B,H,W] expected")
    oh, ow = (H, W) if out_hw is None else (int(out_hw[0]), int(out_hw[1]))
    ws = torch.empty(max(1, lib.tp_eval_metrics_workspace_bytes(B, oh, ow) // 4), device=image.device)
    out = torch.empty(B, 2, dtype=torch.float64, device=image.device)
    a = _lib.EvalMetricsArgs()
    a.rgb_static, a.image, a.obj_mask = rgb_static.data_ptr(), image.data_ptr(), obj_mask.data_ptr()
    a.B, a.h, a.w, a.out_h, a.out_w = B, H, W, oh, ow
    a.workspace, a.out = ws.data_ptr(), out.data_ptr()
    check(lib.tp_eval_metrics(C.byref(a), _stream()), "tp_eval_metrics")
    n = float(B * 3 * oh * ow)
    mse = out[:, 0].sum() / n
    return -10.0 * torch.log10(mse), out[:, 1].sum() / n, mse


_ticket_words = {}           # (device index, stream, entry point) -> one zero-filled int32 word (the kernel leaves it zero)


# ---- pairs: two calls of ONE pairable op issued as one launch (tp_*_pair: the real and the fake pass of the discriminator step)
_pair_state = {"active": False, "pending": None}
PAIRABLE = ("tp_conv4s2_fwd_inorm", "tp_conv4s2_dgrad", "tp_conv4s2_wgrad", "tp_disc_tail_fwd", "tp_disc_tail_bwd", "tp_inorm_lrelu_bwd")


class paired:
    """``with ops.paired():`` -- inside, calls of the pairable ops (PAIRABLE) must come in twos of the same op; the first of a pair only
    prepares its arguments and outputs, the second launches both problems in ONE launch.  Outputs are returned by each call as usual
    and are valid behind the pair's launch.  The second problem of a pair gets its own tile counters / workspace / ticket (slot 1)."""

    def __enter__(self):
        if _pair_state["active"]:
            raise RuntimeError("ops.paired() does not nest")
        _pair_state["active"], _pair_state["pending"] = True, None
        return self

    def __exit__(self, exc_type, exc, tb):
        pending, _pair_state["active"], _pair_state["pending"] = _pair_state["pending"], False, None
        if exc_type is None and pending is not None:
            raise RuntimeError("ops.paired(): %s was issued without a partner" % pending[0])
        return False


def _pair_slot() -> int:
    """0 for the first problem of a pair (and outside ops.paired()), 1 for the second: which scratch set an op takes."""
    return 1 if _pair_state["active"] and _pair_state["pending"] is not None else 0


def _launch(name: str, args, extra=()):
    """Launch ``lib.<name>(byref(args), *extra, stream)`` -- or, inside ops.paired(), hold it back / launch it with its partner through
    ``lib.<name>_pair``.  ``extra``: per-problem trailing arguments (pointers), interleaved per problem in the pair entry points as the
    header declares them; scalars shared by both problems are taken from the second call."""
    lib = _lib.load()
    if not _pair_state["active"]:
        if name == "tp_inorm_lrelu_bwd":
            check(lib.tp_inorm_lrelu_bwd(args.xhat, args.rstd, args.gy, args.n_inst, args.hw, args.slope, args.addend, args.gx, _stream()), name)
        elif name == "tp_conv4s2_fwd_inorm":
            check(lib.tp_conv4s2_fwd_inorm(C.byref(args), extra[2], extra[3], extra[0], extra[1], _stream()), name)
        else:
            check(getattr(lib, name)(C.byref(args), _stream()), name)
        return
    pending = _pair_state["pending"]
    if pending is None:
        _pair_state["pending"] = (name, args, extra, torch.cuda.current_stream().cuda_stream)
        return
    p_name, p_args, p_extra, p_stream = pending
    _pair_state["pending"] = None
    if p_name != name or p_stream != torch.cuda.current_stream().cuda_stream:
        raise RuntimeError("ops.paired(): %s cannot be paired with %s (same op, same stream)" % (name, p_name))
    if name == "tp_conv4s2_fwd_inorm":
        check(lib.tp_conv4s2_fwd_inorm_pair(C.byref(p_args), p_extra[0], p_extra[1], C.byref(args), extra[0], extra[1], extra[2], extra[3],
                                            _stream()), name + "_pair")
    else:
        check(getattr(lib, name + "_pair")(C.byref(p_args), C.byref(args), _stream()), name + "_pair")


def _ticket(dev, name: str) -> int:
    """The arrival counter of a last-block hand-over (tp_nerf_losses_fwd, tp_adam_step) for the CURRENT stream: launches of one
    entry point that overlap on different streams of a device must not count each other's arrivals, so the word is owned by
    (device, stream, entry point).  Made on first use; inside a hipGraph capture on a stream that has none yet the fill that
    creates it is simply part of the captured step (correct, one launch more: warm up on the capturing stream to avoid it)."""
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream, name)
    t = _ticket_words.get(key)
    if t is None:
        t = _ticket_words[key] = torch.zeros(1, dtype=torch.int32, device=dev)
    return t.data_ptr()


@_on_tensor_device
def spectral_norm_fwd(weights, us, vs, training: bool, keep_uv: bool = False, out=None):
    """weights[i] [out, ...] (contiguous), us[i] [out], vs[i] [K]: one power iteration per weight when ``training``
    (u, v updated IN PLACE, like torch.nn.utils.spectral_norm), then W_sn = W / sigma.  Returns (W_sn list, sigma
    list of 1-element tensors).  All weights of a module in three launches (two in eval mode).  ``keep_uv``: also returns copies of u / v as
    they stand after this call (written by the last launch) as a third / fourth list.
    ``out`` = (W_sn list, sigma list, u-copy list, v-copy list, work list) of pre-allocated tensors (spectral_norm_buffers): nothing is
    allocated -- the captured training step writes the NEXT iteration's normalised weights into static buffers."""
    lib = _lib.load()
    n = len(weights)
    arr = (_lib.SnWeight * n)()
    outs, sigmas, keep = [], [], []
    u_copies = v_copies = None
    if out is not None:
        keep_uv = True
        u_copies, v_copies = list(out[2]), list(out[3])
    elif keep_uv:
        flat = torch.empty(sum(u.numel() + v.numel() for u, v in zip(us, vs)), device=us[0].device)
        parts = flat.split([t.numel() for t in list(us) + list(vs)])
        u_copies, v_copies = list(parts[:n]), list(parts[n:])
    for i, (w, u, v) in enumerate(zip(weights, us, vs)):
        w = _f32(w, "weight")
        rows, cols = w.shape[0], w.numel() // w.shape[0]
        if out is not None:
            o, sg, wk = _out_like(out[0][i], w), out[1][i], out[4][i]
        else:
            o, sg = torch.empty_like(w), torch.empty(1, device=w.device)
            wk = torch.empty(lib.tp_sn_work_floats(rows, cols), device=w.device)
        a = arr[i]
        a.weight, a.u, a.v, a.weight_sn, a.sigma, a.work = w.data_ptr(), u.data_ptr(), v.data_ptr(), o.data_ptr(), sg.data_ptr(), wk.data_ptr()
        a.rows, a.cols = rows, cols
        if keep_uv:
            a.u_out, a.v_out = u_copies[i].data_ptr(), v_copies[i].data_ptr()
        outs.append(o); sigmas.append(sg); keep += [w, wk]
    check(lib.tp_sn_fwd(arr, n, int(bool(training)), _stream()), "tp_sn_fwd")
    return (outs, sigmas, u_copies, v_copies) if keep_uv else (outs, sigmas)


SN_MAX_SETS = 3


@_on_tensor_device
def spectral_norm_fwd_sets(weights, us, vs, n_sets: int):
    """``n_sets`` training-mode `spectral_norm_fwd(..., keep_uv=True)` calls in a row -- each advances u / v once -- as 2 n_sets + 1
    launches instead of 3 n_sets (tp_sn_fwd_sets: one normalisation launch for all sets, W read once by it).  Returns a list of n_sets
    tuples (W_sn list, sigma list, u copies, v copies); bit-identical to the separate calls."""
    lib = _lib.load()
    n = len(weights)
    if not 1 <= n_sets <= SN_MAX_SETS:
        raise ValueError("spectral_norm_fwd_sets: 1..%d sets" % SN_MAX_SETS)
    arr = (_lib.SnWeight * (n * n_sets))()
    ws = [_f32(w, "weight") for w in weights]
    works = [torch.empty(lib.tp_sn_work_floats(w.shape[0], w.numel() // w.shape[0]), device=w.device) for w in ws]
    sets = []
    for k in range(n_sets):
        flat = torch.empty(sum(u.numel() + v.numel() for u, v in zip(us, vs)), device=us[0].device)
        parts = flat.split([t.numel() for t in list(us) + list(vs)])
        u_copies, v_copies = list(parts[:n]), list(parts[n:])
        outs, sigmas = [], []
        for i, (w, u, v) in enumerate(zip(ws, us, vs)):
            o, sg = torch.empty_like(w), torch.empty(1, device=w.device)
            a = arr[k * n + i]
            a.weight, a.u, a.v, a.weight_sn, a.sigma, a.work = w.data_ptr(), u.data_ptr(), v.data_ptr(), o.data_ptr(), sg.data_ptr(), works[i].data_ptr()
            a.rows, a.cols = w.shape[0], w.numel() // w.shape[0]
            a.u_out, a.v_out = u_copies[i].data_ptr(), v_copies[i].data_ptr()
            outs.append(o); sigmas.append(sg)
        sets.append((outs, sigmas, u_copies, v_copies))
    check(lib.tp_sn_fwd_sets(arr, n, n_sets, _stream()), "tp_sn_fwd_sets")
    return sets


def spectral_norm_buffers(weights, us, vs):
    """Pre-allocated outputs of one `spectral_norm_fwd(..., out=)` call: (W_sn, sigma, u copies, v copies, work)."""
    lib = _lib.load()
    dev = weights[0].device
    return ([torch.empty_like(w, memory_format=torch.contiguous_format) for w in weights], [torch.empty(1, device=dev) for _ in weights],
            [torch.empty_like(u) for u in us], [torch.empty_like(v) for v in vs],
            [torch.empty(lib.tp_sn_work_floats(w.shape[0], w.numel() // w.shape[0]), device=dev) for w in weights])


@_on_tensor_device
def spectral_norm_bwd(grads_sn, weights_sn, us, vs, sigmas, accumulate_into=None, second=None, step=None):
    """dL/dW from dL/dW_sn with u, v treated as constants (torch's convention): (G - <G, W_sn> u v^T) / sigma.
    ``accumulate_into``: a list of tensors the results are ADDED to; they are what is returned.
    ``second`` = (grads_sn2, weights_sn2, us2, vs2, sigmas2): a second normalised instance of the same weights in one optimiser
    step (the discriminator step's fake pass) -- its term is added inside the same two launches.
    ``step`` = dict(terms, weights, flags=dict(bad, word_finite, snapshot), params, square_avgs, steps, lr, alpha, eps): the END of a
    discriminator step in the same two launches (tp_sn_bwd_step) -- the loss total and the step gate (what `weighted_sum(flags=)`
    does) in the first, the RMSprop update of ``params`` (what `rmsprop_step(gate=snapshot)` does) in the second; the total goes to
    ``step["total"]`` (a 0-dim tensor made here)."""
    lib = _lib.load()
    n = len(grads_sn)
    arr = (_lib.SnWeight * n)()
    outs, keep = [], []
    for i, (g, ws, u, v, sg) in enumerate(zip(grads_sn, weights_sn, us, vs, sigmas)):
        g = _f32(g, "grad")
        rows, cols = ws.shape[0], ws.numel() // ws.shape[0]
        o = torch.empty_like(ws) if accumulate_into is None else _out_like(accumulate_into[i], ws)
        arr[i].accumulate = 0 if accumulate_into is None else 1
        wk = torch.empty(lib.tp_sn_work_floats(rows, cols), device=ws.device)
        a = arr[i]
        a.u, a.v, a.weight_sn, a.sigma, a.grad_sn, a.grad, a.work = (u.data_ptr(), v.data_ptr(), ws.data_ptr(), sg.data_ptr(),
                                                                       g.data_ptr(), o.data_ptr(), wk.data_ptr())
        a.rows, a.cols = rows, cols
        if second is not None:
            g2 = _f32(second[0][i], "grad")
            if g2.shape != g.shape or second[1][i].shape != ws.shape:
                raise ValueError("spectral_norm_bwd: the second instance must have the shapes of the first")
            a.grad_sn2, a.weight_sn2, a.u2, a.v2, a.sigma2 = (g2.data_ptr(), second[1][i].data_ptr(), second[2][i].data_ptr(),
                                                              second[3][i].data_ptr(), second[4][i].data_ptr())
            keep.append(g2)
        outs.append(o); keep += [g, wk]
    if step is None:
        check(lib.tp_sn_bwd(arr, n, _stream()), "tp_sn_bwd")
        return outs
    if accumulate_into is not None:
        raise ValueError("spectral_norm_bwd(step=): the gradients are the step's own (no accumulate_into)")
    t = _lib.SnStepTail()
    terms = [_f32(x.detach(), "term") for x in step["terms"]]
    if not 1 <= len(terms) <= 4 or len(step["weights"]) != len(terms) or len(step["params"]) != n:
        raise ValueError("spectral_norm_bwd(step=): 1..4 terms with their weights, one parameter per weight")
    for k, (x, w) in enumerate(zip(terms, step["weights"])):
        t.terms[k], t.weights[k] = x.data_ptr(), float(w)
    flags = step["flags"]
    step["total"] = total = torch.empty((), device=terms[0].device)
    t.n_terms, t.word_finite, t.total = len(terms), int(flags["word_finite"]), total.data_ptr()
    t.bad, t.snapshot, t.n_bad = flags["bad"].data_ptr(), flags["snapshot"].data_ptr(), flags["bad"].numel()
    if flags["snapshot"].numel() != flags["bad"].numel() or flags["bad"].dtype != torch.int32 or flags["snapshot"].dtype != torch.int32:
        raise ValueError("spectral_norm_bwd(step=): int32 gate words and a snapshot of the same length")
    steps = step.get("steps") or [None] * n
    for i, (p, sq, st, o) in enumerate(zip(step["params"], step["square_avgs"], steps, outs)):
        if not (p.is_contiguous() and sq.is_contiguous() and p.dtype == sq.dtype == torch.float32 and p.shape == o.shape == sq.shape):
            raise _lib.TexposeLibraryError("spectral_norm_bwd(step=): contiguous float32 parameters shaped like their gradients")
        t.param[i], t.square_avg[i] = p.data_ptr(), sq.data_ptr()
        if st is not None:
            if not (st.is_cuda and st.dtype == torch.float32):
                raise _lib.TexposeLibraryError("spectral_norm_bwd(step=): step counters must be float32 device tensors")
            t.step[i] = st.data_ptr()
    lr = step["lr"]
    t.lr_dev = lr.data_ptr() if isinstance(lr, torch.Tensor) else None
    t.lr_host = 0.0 if isinstance(lr, torch.Tensor) else float(lr)
    t.alpha, t.one_minus_alpha, t.eps = float(step["alpha"]), float(1.0 - float(step["alpha"])), float(step["eps"])
    check(lib.tp_sn_bwd_step(arr, n, C.byref(t), _stream()), "tp_sn_bwd_step")
    keep.append(terms)
    return outs


def _nerf_losses_args(rgb, uncert, density, gathered):
    lib = _lib.load()
    rgb, uncert, density, gathered = (_f32(rgb, "rgb"), _f32(uncert, "uncert"), _f32(density, "density"),
                                      _f32(gathered, "gathered"))
    B, P = rgb.shape[0], rgb.shape[1]
    N = density.shape[2]
    if rgb.shape != (B, P, 3) or uncert.numel() != B * P or density.shape != (B, P, N, 2) or gathered.numel() != B * 14 * P:
        raise ValueError("nerf_losses: rgb [B,P,3], uncert [B,P,1], density [B,P,N,2], gathered [B,14,p,p] expected")
    a = _lib.NerfLossesArgs()
    a.rgb, a.uncert, a.density, a.gathered = rgb.data_ptr(), uncert.data_ptr(), density.data_ptr(), gathered.data_ptr()
    a.B, a.P, a.N = B, P, N
    return lib, a, (rgb, uncert, density, gathered)


@_on_tensor_device
def nerf_losses_fwd(rgb: Tensor, uncert: Tensor, density: Tensor, gathered: Tensor, want_losses: bool = False):
    """Four fp64 sums [sum m*se/u^2, sum m, sum log u^2, sum sigma_t] (device tensor) for the render / uncert /
    trans_reg terms of the generator step (reference compute_loss :747-760); with ``want_losses`` also the three fp32 loss
    values [render, uncert, trans_reg] formed by the same launch."""
    lib, a, keep = _nerf_losses_args(rgb, uncert, density, gathered)
    ws = torch.empty(4 * _lib.NERF_LOSSES_MAX_BLOCKS, device=rgb.device)
    sums = torch.empty(4, dtype=torch.float64, device=rgb.device)
    losses = torch.empty(3, device=rgb.device) if want_losses else None
    a.workspace, a.sums, a.losses = ws.data_ptr(), sums.data_ptr(), _ptr(losses)
    a.ticket = _ticket(rgb.device, "nerf_losses")
    check(lib.tp_nerf_losses_fwd(C.byref(a), _stream()), "tp_nerf_losses_fwd")
    return (sums, losses) if want_losses else sums


@_on_tensor_device
def nerf_losses_bwd(rgb: Tensor, uncert: Tensor, density: Tensor, gathered: Tensor, sums: Tensor, g_losses):
    """Gradients wrt rgb, uncert, density for the upstream gradients of (render, uncert, trans_reg): ``g_losses`` is a [3] tensor or
    a triple of 0-dim tensors / None (None = zero; no stacking launch)."""
    lib, a, keep = _nerf_losses_args(rgb, uncert, density, gathered)
    ws = torch.empty(4, device=rgb.device)
    a.workspace, a.sums = ws.data_ptr(), sums.data_ptr()
    if torch.is_tensor(g_losses):
        g_losses = _f32(g_losses, "g_losses")
        gs = [g_losses[k] for k in range(3)]
    else:
        gs = [None if g is None else _f32(g, "g_loss") for g in g_losses]
    ptrs = [None if g is None else g.data_ptr() for g in gs]
    g_rgb, g_unc, g_den = torch.empty_like(keep[0]), torch.empty_like(keep[1]), torch.empty_like(keep[2])
    job = _pending_total.pop("job", None)
    if job is not None and job["stream"] == _stream():
        # (the generator step's loss total + gate, handed over by weighted_sum(defer=True): a side job of this launch)
        tp, wsf, n, out, flags = job["ptrs"], job["ws"], job["n"], job["out"], job["flags"]
        bad = flags["bad"]
        check(lib.tp_nerf_losses_bwd_total(C.byref(a), ptrs[0], ptrs[1], ptrs[2], g_rgb.data_ptr(), g_unc.data_ptr(), g_den.data_ptr(),
                                           tp, wsf, n, out.data_ptr(), _ptr(flags.get("status")), bad.data_ptr(), bad.numel(),
                                           int(flags.get("word_status", 0)), int(flags["word_finite"]), flags["snapshot"].data_ptr(),
                                           _ptr(flags.get("step_counter")), _stream()), "tp_nerf_losses_bwd_total")
        return g_rgb, g_unc, g_den
    if job is not None:
        _pending_total["job"] = job
    check(lib.tp_nerf_losses_bwd(C.byref(a), ptrs[0], ptrs[1], ptrs[2], g_rgb.data_ptr(), g_unc.data_ptr(), g_den.data_ptr(),
                                 _stream()), "tp_nerf_losses_bwd")
    return g_rgb, g_unc, g_den


# ------------------------------------------------------------------------------------------ K9
@_on_tensor_device
def inorm_lrelu_fwd(x: Tensor, eps: float, slope: float, y_out: Optional[Tensor] = None):
    """x [B,C,H,W] -> (y, xhat, rstd [B*C]) = LeakyReLU(InstanceNorm2d(x)) and what its derivatives need.  ``y_out``: where y is
    written (a contiguous view of x's shape, e.g. one half of a stacked buffer)."""
    lib = _lib.load()
    x = _f32(x, "x")
    n_inst, hw = x.shape[0] * x.shape[1], x.shape[2] * x.shape[3]
    y, xhat = _out_like(y_out, x), torch.empty_like(x)
    rstd = torch.empty(n_inst, device=x.device)
    check(lib.tp_inorm_lrelu_fwd(x.data_ptr(), n_inst, hw, float(eps), float(slope), y.data_ptr(), xhat.data_ptr(),
                                 rstd.data_ptr(), _stream()), "tp_inorm_lrelu_fwd")
    return y, xhat, rstd


@_on_tensor_device
def inorm_lrelu_bwd(xhat: Tensor, rstd: Tensor, gy: Tensor, slope: float, addend: Optional[Tensor] = None,
                    out: Optional[Tensor] = None) -> Tensor:
    """gx; ``addend`` (same shape) is added to it in the same launch (a second cotangent of x)."""
    lib = _lib.load()
    gy = _f32(gy, "gy")
    gx = _out_like(out, xhat)
    if addend is not None:
        addend = _f32(addend, "addend")
        if addend.numel() != xhat.numel():
            raise ValueError("inorm_lrelu_bwd: addend must have the shape of x")
    a = _lib.InormBwdArgs()
    a.xhat, a.rstd, a.gy, a.n_inst, a.hw, a.slope = xhat.data_ptr(), rstd.data_ptr(), gy.data_ptr(), rstd.numel(), xhat.numel() // rstd.numel(), float(slope)
    a.addend, a.gx = _ptr(addend), gx.data_ptr()
    a._keep = (xhat, rstd, gy, addend, gx)
    _launch("tp_inorm_lrelu_bwd", a)
    return gx


@_on_tensor_device
def inorm_lrelu_bwd_bwd(xhat: Tensor, rstd: Tensor, gy: Tensor, ggx: Tensor, slope: float, out_gy: Optional[Tensor] = None):
    """cotangent ggx of the backward's output gx -> (grad wrt gy, grad wrt x)."""
    lib = _lib.load()
    gy, ggx = _f32(gy, "gy"), _f32(ggx, "ggx")
    g_gy, g_x = _out_like(out_gy, xhat), torch.empty_like(xhat)
    check(lib.tp_inorm_lrelu_bwd_bwd(xhat.data_ptr(), rstd.data_ptr(), gy.data_ptr(), ggx.data_ptr(), rstd.numel(),
                                     xhat.numel() // rstd.numel(), float(slope), g_gy.data_ptr(), g_x.data_ptr(), _stream()),
          "tp_inorm_lrelu_bwd_bwd")
    return g_gy, g_x


# ------------------------------------------------------------------------------------------ K10
@_on_tensor_device
def rmsprop_step(params, grads, square_avgs, lr, alpha: float = 0.99, eps: float = 1e-8, gate: Optional[Tensor] = None, steps=None) -> None:
    """One launch: sq = alpha sq + (1 - alpha) g^2;  p -= lr g / (sqrt(sq) + eps) for up to 16 tensors per call.
    ``lr``: python float, or a 0-dim CUDA tensor read on the device (captured training step).  ``gate``: int32 device words;
    if any is non-zero nothing is changed (the step counters included)."""
    lib = _lib.load()
    lr_dev = lr.data_ptr() if isinstance(lr, torch.Tensor) else None
    lr_host = 0.0 if isinstance(lr, torch.Tensor) else float(lr)
    for i0 in range(0, len(params), _lib.RMSPROP_MAX_TENSORS):
        chunk = list(zip(params, grads, square_avgs, steps if steps is not None else [None] * len(params)))[i0:i0 + _lib.RMSPROP_MAX_TENSORS]
        arr = (_lib.RmspropTensor * len(chunk))()
        for a, (p, g, sq, st) in zip(arr, chunk):
            if not (p.is_contiguous() and g.is_contiguous() and sq.is_contiguous() and p.dtype == g.dtype == sq.dtype == torch.float32):
                raise _lib.TexposeLibraryError("rmsprop_step needs contiguous float32 tensors")
            a.param, a.grad, a.square_avg, a.numel = p.data_ptr(), g.data_ptr(), sq.data_ptr(), p.numel()
            if st is not None:                                 # (``steps``: 0-dim float32 DEVICE tensors, += 1 by the same launch)
                if not (st.is_cuda and st.dtype == torch.float32):
                    raise _lib.TexposeLibraryError("rmsprop_step: step counters must be float32 device tensors")
                a.step = st.data_ptr()
        check(lib.tp_rmsprop_step(arr, len(chunk), lr_dev, lr_host, float(alpha), float(eps), _ptr(gate), gate.numel() if gate is not None else 0,
                                  _stream()), "tp_rmsprop_step")


@_on_tensor_device
def render_eval(packed: Tensor, intr: Tensor, pose: Tensor, ray_idx: Tensor, z_near: Tensor, z_far: Tensor, lat_trans: Tensor,
                lat_light: Tensor, *, H: int, W: int, n_samples: int, precision: str = "f16x3", min_uncert: float = 0.05,
                rand: Optional[Tensor] = None, with_alphas: bool = False, ray_bias: bool = False, ndc: bool = False,
                depth_param: str = "metric"):
    """The C ABI's one-call evaluation render (tp_render_eval: ray-gen + MLP + composite, intermediates in one workspace).
    ``ray_bias``: ``packed`` is the ray-bias stream (pack_weights(..., ray_bias=True); f16x3, n_samples % 128 == 0).
    Returns out_ray [B,R,14] (COMPOSITE_RAY_FIELDS) and, if asked, (alpha_static, alpha_transient) [B,R,N].  The Python
    mirror (Graph.render) launches the same three kernels itself; this entry point exists for non-Python hosts."""
    lib = _lib.load()
    intr, pose = _f32(intr, "intr"), _f32(pose, "pose")
    z_near, z_far = _f32(z_near, "z_near"), _f32(z_far, "z_far")
    lat_trans, lat_light = _f32(lat_trans, "lat_trans"), _f32(lat_light, "lat_light")
    ray_idx = ray_idx.to(torch.int64).contiguous()
    B, R = ray_idx.shape
    dev = pose.device
    a = _lib.RenderEvalArgs()
    rg = a.raygen
    rg.intr, rg.pose, rg.ray_idx, rg.z_near, rg.z_far = intr.data_ptr(), pose.data_ptr(), ray_idx.data_ptr(), z_near.data_ptr(), z_far.data_ptr()
    rg.B, rg.R, rg.H, rg.W, rg.N = B, R, H, W, n_samples
    rg.pixel_mode, rg.bounds_mode = PIX_INDEX, BOUNDS_MAP
    rg.ndc, rg.depth_param = int(bool(ndc)), DEPTH_PARAMS[depth_param]
    if rand is not None:
        rand = _f32(rand, "rand")
        rg.rand, rg.jitter_mode = rand.data_ptr(), JITTER_GIVEN
    else:
        rg.jitter_mode = JITTER_MID
    a.packed, a.lat_trans, a.lat_light = packed.data_ptr(), lat_trans.data_ptr(), lat_light.data_ptr()
    a.precision, a.min_uncert = PRECISIONS[precision], float(min_uncert)
    a.packed_ray_bias = 1 if ray_bias else 0
    a.status = mlp_status(dev).data_ptr() if precision == "f16x3" else None
    ws = torch.empty(int(lib.tp_render_eval_workspace_bytes(B, R, n_samples)) // 4 + 64, device=dev)
    out = torch.empty(B, R, 14, device=dev)
    a.workspace, a.out_ray = ws.data_ptr(), out.data_ptr()
    alphas = None
    if with_alphas:
        alphas = (torch.empty(B, R, n_samples, device=dev), torch.empty(B, R, n_samples, device=dev))
        a.alpha_static, a.alpha_transient = alphas[0].data_ptr(), alphas[1].data_ptr()
    check(lib.tp_render_eval(C.byref(a), _stream()), "tp_render_eval")
    return (out, alphas) if with_alphas else out


# ------------------------------------------------------------------------------------------ K11
_conv_counters = {}          # (device index, stream) -> zero-filled int32 tensor (the kernels leave it zero)
_conv_counters_retired = []  # outgrown counter tensors: a captured hipGraph may still hold their address -- never freed


def _conv_scratch(lib, ws_fn, a, op: int, dev):
    """(workspace tensor or None, counters tensor) for a K11 / K12 launch described by the argument struct ``a``."""
    n_cnt = C.c_int64(0)
    ws_floats = ws_fn(C.byref(a), op, C.byref(n_cnt))
    if ws_floats < 0:
        check(-1, "tp_conv_workspace")
    # one counter array per (device, stream): launches on different streams may run concurrently (the two branches of the
    # captured training step) and must not see each other's tile arrivals
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream, _pair_slot())
    cnt = _conv_counters.get(key)
    if cnt is None or cnt.numel() < n_cnt.value:
        if torch.cuda.is_current_stream_capturing():
            raise _lib.TexposeLibraryError("tp_conv: the tile counters must exist before a hipGraph capture (run one "
                                           "eager step first)")
        if cnt is not None:
            _conv_counters_retired.append(cnt)          # a graph captured earlier keeps incrementing / zeroing this one
        cnt = torch.zeros(max(int(n_cnt.value), 1 << 14), dtype=torch.int32, device=dev)
        _conv_counters[key] = cnt
    return (torch.empty(int(ws_floats), device=dev) if ws_floats else None), cnt


def _conv4s2(op: int, name: str, x, w, gy, out, N, C_in, H, W, Co, inorm=None):
    lib = _lib.load()
    a = _lib.Conv4s2Args()
    a.N, a.C, a.H, a.W, a.Co = int(N), int(C_in), int(H), int(W), int(Co)
    ws, cnt = _conv_scratch(lib, lib.tp_conv4s2_workspace, a, op, out.device)
    a.x, a.w, a.gy = _ptr(x), _ptr(w), _ptr(gy)
    a.out, a.counters, a.workspace = out.data_ptr(), cnt.data_ptr(), _ptr(ws)
    a._keep = (x, w, gy, out, ws, cnt)                # (a held-back first problem of a pair keeps its tensors alive)
    if inorm is not None:                             # (xhat, rstd, addend or None, gx, slope, skip_out)
        a.in_xhat, a.in_rstd, a.in_addend, a.in_gx = inorm[0].data_ptr(), inorm[1].data_ptr(), _ptr(inorm[2]), inorm[3].data_ptr()
        a.in_slope, a.skip_out = float(inorm[4]), int(bool(inorm[5]))
        a._keep = a._keep + tuple(inorm[:4])
    if name in PAIRABLE:
        _launch(name, a)
    else:
        check(getattr(lib, name)(C.byref(a), _stream()), name)
    return out


@_on_tensor_device
def conv4s2_fwd(x: Tensor, w: Tensor, out: Optional[Tensor] = None) -> Tensor:
    """conv2d(x [N,C,H,W], w [Co,C,4,4], stride 2, padding 1) -> [N,Co,H/2,W/2]."""
    x, w = _f32(x, "x"), _f32(w, "w")
    N, C_in, H, W = x.shape
    y = _out_like(out, x, (N, w.shape[0], H // 2, W // 2))
    return _conv4s2(_lib.CONV_FWD, "tp_conv4s2_fwd", x, w, None, y, N, C_in, H, W, w.shape[0])


def conv4s2_fwd_inorm_supported(x: Tensor) -> bool:
    """The fused convolution + InstanceNorm + LeakyReLU launch covers 4x4 and 8x8 output maps (whole instances per workgroup)."""
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and (x.shape[2] // 2) * (x.shape[3] // 2) in (16, 64)
            and x.shape[2] == x.shape[3] and not knobs.K.no_conv_inorm)


@_on_tensor_device
def conv4s2_fwd_inorm(x: Tensor, w: Tensor, eps: float, slope: float, y_out: Optional[Tensor] = None, copy_to: Optional[Tensor] = None):
    """-> (y, xhat, rstd) = inorm_lrelu_fwd(conv4s2_fwd(x, w), eps, slope) in ONE launch (the normalisation runs in the epilogue of
    the workgroup that holds an instance's split-K totals); ``y_out`` as in inorm_lrelu_fwd.  ``copy_to`` (shaped like x, contiguous):
    the launch also leaves a copy of x there."""
    lib = _lib.load()
    x, w = _f32(x, "x"), _f32(w, "w")
    N, C_in, H, W = x.shape
    Co = w.shape[0]
    y = _out_like(y_out, x, (N, Co, H // 2, W // 2))
    xhat, rstd = torch.empty(N, Co, H // 2, W // 2, device=x.device), torch.empty(N * Co, device=x.device)
    a = _lib.Conv4s2Args()
    a.N, a.C, a.H, a.W, a.Co = int(N), int(C_in), int(H), int(W), int(Co)
    ws, cnt = _conv_scratch(lib, lambda args, _op, n: lib.tp_conv4s2_fwd_inorm_workspace(args, n), a, 0, x.device)
    a.x, a.w = x.data_ptr(), w.data_ptr()
    a.out, a.counters, a.workspace = y.data_ptr(), cnt.data_ptr(), _ptr(ws)
    a._keep = (x, w, y, xhat, rstd, ws, cnt, copy_to)
    if copy_to is not None:
        if copy_to.shape != x.shape or copy_to.dtype != torch.float32 or not copy_to.is_contiguous() or copy_to.device != x.device:
            raise ValueError("conv4s2_fwd_inorm: copy_to must be a contiguous float32 tensor shaped like x")
        a.x_copy = copy_to.data_ptr()
    _launch("tp_conv4s2_fwd_inorm", a, (xhat.data_ptr(), rstd.data_ptr(), float(eps), float(slope)))
    return y, xhat, rstd


def conv4s2_dgrad_inorm_supported(gy: Tensor) -> bool:
    """The data gradient can carry the InstanceNorm + LeakyReLU backward of the stage in front of it: 8x8 input maps (whole instances
    per workgroup)."""
    return gy.is_cuda and gy.dim() == 4 and tuple(gy.shape[-2:]) == (4, 4) and not knobs.K.no_dgrad_inorm


@_on_tensor_device
def conv4s2_dgrad(gy: Tensor, w: Tensor, out: Optional[Tensor] = None, inorm=None):
    """gradient of conv4s2_fwd wrt x: gy [N,Co,H/2,W/2], w [Co,C,4,4] -> [N,C,H,W].
    ``inorm`` = dict(xhat [N,C,8,8], rstd [N*C], slope, addend=None, out=None, keep=True) (conv4s2_dgrad_inorm_supported): the
    InstanceNorm + LeakyReLU backward of the stage in front of the convolution in the same launch -- returns (data gradient or None if
    not ``keep``, inorm_lrelu_bwd(xhat, rstd, data gradient, slope, addend, out)), bit-identical to the two launches."""
    gy, w = _f32(gy, "gy"), _f32(w, "w")
    N, Co, OH, OW = gy.shape
    gx = _out_like(out, gy, (N, w.shape[1], 2 * OH, 2 * OW))
    if inorm is None:
        return _conv4s2(_lib.CONV_DGRAD, "tp_conv4s2_dgrad", None, w, gy, gx, N, w.shape[1], 2 * OH, 2 * OW, Co)
    xhat, rstd = _f32(inorm["xhat"], "xhat"), _f32(inorm["rstd"], "rstd")
    if tuple(xhat.shape) != (N, w.shape[1], 8, 8) or (OH, OW) != (4, 4) or rstd.numel() != N * w.shape[1]:
        raise ValueError("conv4s2_dgrad(inorm=): xhat [N,C,8,8] / rstd [N*C] of the stage in front of the convolution expected")
    addend = _f32(inorm["addend"], "addend") if inorm.get("addend") is not None else None
    cz = _out_like(inorm.get("out"), xhat)
    keep = bool(inorm.get("keep", True))
    _conv4s2(_lib.CONV_DGRAD, "tp_conv4s2_dgrad", None, w, gy, gx, N, w.shape[1], 2 * OH, 2 * OW, Co,
             inorm=(xhat, rstd, addend, cz, inorm["slope"], not keep))
    return (gx if keep else None), cz


@_on_tensor_device
def conv4s2_wgrad(gy: Tensor, x: Tensor, out: Optional[Tensor] = None) -> Tensor:
    """gradient of conv4s2_fwd wrt w: gy [N,Co,H/2,W/2], x [N,C,H,W] -> [Co,C,4,4] (a sum over the N samples: several
    (gy, x) pairs stacked along N give the sum of their weight gradients in one launch)."""
    gy, x = _f32(gy, "gy"), _f32(x, "x")
    N, C_in, H, W = x.shape
    gw = _out_like(out, x, (gy.shape[1], C_in, 4, 4))
    return _conv4s2(_lib.CONV_WGRAD, "tp_conv4s2_wgrad", x, None, gy, gw, N, C_in, H, W, gy.shape[1])


# ------------------------------------------------------------------------------------------ K12
def conv3s1_supported(x: Tensor) -> bool:
    H, W = x.shape[-2:]
    return x.is_cuda and x.dtype == torch.float32 and H >= 4 and W >= 4 and (H & (H - 1)) == 0 and (W & (W - 1)) == 0


@_on_tensor_device
def conv3s1_fwd(x: Tensor, w: Tensor, bias: Optional[Tensor], relu: bool) -> Tensor:
    """relu?(conv2d(x [N,C,H,W], w [Co,C,3,3], bias, stride 1, padding 1))."""
    lib = _lib.load()
    x, w = _f32(x, "x"), _f32(w, "w")
    bias = _f32(bias, "bias") if bias is not None else None
    N, C_in, H, W = x.shape
    y = torch.empty(N, w.shape[0], H, W, device=x.device)
    a = _lib.Conv3s1Args()
    a.N, a.C, a.H, a.W, a.Co, a.relu = N, C_in, H, W, w.shape[0], int(bool(relu))
    ws, cnt = _conv_scratch(lib, lib.tp_conv3s1_workspace, a, _lib.CONV_FWD, x.device)
    a.inp, a.w, a.bias, a.out, a.counters, a.workspace = x.data_ptr(), w.data_ptr(), _ptr(bias), y.data_ptr(), cnt.data_ptr(), _ptr(ws)
    check(lib.tp_conv3s1_fwd(C.byref(a), _stream()), "tp_conv3s1_fwd")
    return y


@_on_tensor_device
def conv3s1_dgrad(gy: Tensor, w: Tensor, mask: Optional[Tensor]) -> Tensor:
    """gradient of conv3s1_fwd wrt x; ``mask`` = the forward output of a ReLU layer (gy counts where it is > 0)."""
    lib = _lib.load()
    gy, w = _f32(gy, "gy"), _f32(w, "w")
    mask = _f32(mask, "mask") if mask is not None else None
    N, Co, H, W = gy.shape
    gx = torch.empty(N, w.shape[1], H, W, device=gy.device)
    a = _lib.Conv3s1Args()
    a.N, a.C, a.H, a.W, a.Co, a.relu = N, w.shape[1], H, W, Co, 0
    ws, cnt = _conv_scratch(lib, lib.tp_conv3s1_workspace, a, _lib.CONV_DGRAD, gy.device)
    a.inp, a.w, a.mask, a.out, a.counters, a.workspace = gy.data_ptr(), w.data_ptr(), _ptr(mask), gx.data_ptr(), cnt.data_ptr(), _ptr(ws)
    check(lib.tp_conv3s1_dgrad(C.byref(a), _stream()), "tp_conv3s1_dgrad")
    return gx


# ------------------------------------------------------------------------------------------ K13
_lattices = {}


@_on_tensor_device
def patch_coords(u: Optional[Tensor], patch_size: int, lo, hi: float, random_scale: bool = True, random_shift: bool = True, *,
                 nbatch: Optional[int] = None, seed: int = 0, counter: Optional[Tensor] = None, device=None, defer: bool = False):
    """FlexPatchSampler in one launch: u [3,B,...] uniforms -> (coords [B,p,p,2], scales [B,1,1,1]); ``lo`` is a float or a
    0-dim device tensor (the annealed bound of a captured step).  ``u`` None: ``nbatch`` images, the uniforms drawn inside the
    kernel from (``seed``, the device word ``counter``: int64 [1], the step counter of a captured training step).
    ``defer``: nothing is launched; the returned tensors are filled by the ray-generation launch that is given the job stored as
    ``coords._tp_sampler_job`` (`raygen(..., coords=coords, sampler=job)`)."""
    lib = _lib.load()
    if u is not None:
        u = _f32(u, "u")
        B, dev = u.numel() // 3, u.device
    else:
        B, dev = int(nbatch), (counter.device if counter is not None else torch.device(device))
        if counter is not None and (counter.dtype != torch.int64 or counter.numel() != 1):
            raise ValueError("patch_coords: counter must be one int64 device word")
    p = int(patch_size)
    key = (p, dev.index)
    if key not in _lattices:
        _lattices[key] = torch.linspace(-1, 1, p, device=dev)
    coords = torch.empty(B, p, p, 2, device=dev)
    scales = torch.empty(B, 1, 1, 1, device=dev)
    if defer:
        coords._tp_sampler_job = dict(u=u, B=B, p=p, lattice=_lattices[key], lo=lo, hi=hi, random_scale=random_scale, random_shift=random_shift,
                                      seed=seed, counter=counter, coords=coords, scales=scales)
        return coords, scales
    lo_dev = lo.data_ptr() if torch.is_tensor(lo) else None
    lo_host = 0.0 if torch.is_tensor(lo) else float(lo)
    check(lib.tp_patch_coords(_ptr(u), B, p, _lattices[key].data_ptr(), lo_dev, lo_host, float(hi) - lo_host, float(hi),
                              int(bool(random_scale)), int(bool(random_shift)), int(seed) & (2 ** 64 - 1), _ptr(counter),
                              coords.data_ptr(), scales.data_ptr(), _stream()),
          "tp_patch_coords")
    return coords, scales


@_on_tensor_device
def bce_logits_fwd(x: Tensor, target: float) -> Tensor:
    lib = _lib.load()
    x = _f32(x, "x")
    out = torch.empty((), device=x.device)
    check(lib.tp_bce_logits_fwd(x.data_ptr(), x.numel(), float(target), out.data_ptr(), _stream()), "tp_bce_logits_fwd")
    return out


@_on_tensor_device
def bce_logits_bwd(x: Tensor, target: float, g: Tensor) -> Tensor:
    lib = _lib.load()
    x, g = _f32(x, "x"), _f32(g, "g")
    gx = torch.empty_like(x)
    check(lib.tp_bce_logits_bwd(x.data_ptr(), x.numel(), float(target), g.data_ptr(), gx.data_ptr(), _stream()), "tp_bce_logits_bwd")
    return gx


def _feat_args(rgb, gathered, mean, std):
    a = _lib.FeatInputsArgs()
    B, P = rgb.shape[0], rgb.shape[1]
    if rgb.shape != (B, P, 3) or gathered.shape[0] != B or gathered.numel() != B * 14 * P:
        raise ValueError("feat_inputs: rgb [B,P,3] and gathered [B,14,p,p] expected")
    a.rgb, a.gathered, a.B, a.P, a.n_channels = rgb.data_ptr(), gathered.data_ptr(), B, P, 14
    a.c_image, a.c_image_syn, a.c_mask, a.c_mask_syn = 0, 3, 12, 13
    for c in range(3):
        a.mean[c], a.std[c] = float(mean[c]), float(std[c])
    return a


@_on_tensor_device
def feat_inputs_fwd(rgb: Tensor, gathered: Tensor, mean, std, hw) -> Tensor:
    """-> [4B,3,h,w]: the (fake, real) pairs of the feature loss, masked and ImageNet-normalised (K13)."""
    lib = _lib.load()
    rgb, gathered = _f32(rgb, "rgb"), _f32(gathered, "gathered")
    out = torch.empty(4 * rgb.shape[0], 3, hw[0], hw[1], device=rgb.device)
    check(lib.tp_feat_inputs_fwd(C.byref(_feat_args(rgb, gathered, mean, std)), out.data_ptr(), _stream()), "tp_feat_inputs_fwd")
    return out


@_on_tensor_device
def feat_inputs_bwd(rgb: Tensor, gathered: Tensor, mean, std, g_out: Tensor) -> Tensor:
    lib = _lib.load()
    g_out = _f32(g_out, "g_out")
    g_rgb = torch.empty_like(rgb)
    check(lib.tp_feat_inputs_bwd(C.byref(_feat_args(rgb, gathered, mean, std)), g_out.data_ptr(), g_rgb.data_ptr(), _stream()),
          "tp_feat_inputs_bwd")
    return g_rgb


_feat_chain_scratch: Dict[tuple, tuple] = {}


def feat_chain_supported(rgb: Tensor, gathered: Tensor, hw) -> bool:
    """K18 covers 16 x 16 patches of float32 CUDA tensors (tp_feat_chain)."""
    return (rgb.is_cuda and rgb.dtype == torch.float32 and gathered.dtype == torch.float32 and tuple(hw) == (16, 16)
            and rgb.dim() == 3 and rgb.shape[1] == 256 and not knobs.K.no_feat_chain)


def feat_chain_pack(weights, out: Optional[Tensor] = None) -> Tensor:
    """The seven frozen [Co,C,3,3] weights of VGG19 features[:15] in K18's operand order (tp_feat_chain_pack): one launch.  ``out``:
    an earlier result to overwrite (a captured step keeps its address).  The OWNER of the weights caches the result
    (gan_modules.PerceptualLoss): a cache keyed by addresses here would serve a new network the old one's weights."""
    lib = _lib.load()
    shapes = [(64, 3), (64, 64), (128, 64), (128, 128), (256, 128), (256, 256), (256, 256)]
    if len(weights) != 7:
        raise ValueError("feat_chain_pack: the seven convolutions of VGG19 features[:15] expected")
    for w, (co, ci) in zip(weights, shapes):
        if tuple(w.shape) != (co, ci, 3, 3) or not w.is_contiguous() or w.dtype != torch.float32 or not w.is_cuda:
            raise ValueError("feat_chain_pack: weight %s does not fit VGG19 features[:15]" % (tuple(w.shape),))
    dev = weights[0].device
    n = int(lib.tp_feat_chain_packed_floats())
    if out is None:
        if torch.cuda.is_current_stream_capturing():
            raise _lib.TexposeLibraryError("tp_feat_chain_pack: pack before the hipGraph capture (run one eager step first)")
        out = torch.empty(n, device=dev)
    ptrs = (C.c_void_p * 7)(*[w.data_ptr() for w in weights])
    with torch.cuda.device(dev):
        check(lib.tp_feat_chain_pack(ptrs, out.data_ptr(), _stream()), "tp_feat_chain_pack")
    return out


@_on_tensor_device
def feat_chain(rgb: Tensor, gathered: Tensor, packed: Tensor, biases, mean, std, hw, w2: float = 5.0, scale: float = 1.0):
    """K18 (tp_feat_chain): the feature loss of the generator step and its gradient wrt the rendered colours in ONE call --
    (loss3 [3] = {l1 + w2 l2, l1, l2}, g_rgb [B,P,3] = scale * d loss3[0] / d rgb).  ``packed``: `feat_chain_pack` of the seven
    convolution weights of VGG19 features[:15]; ``biases``: theirs.  Workspace and tile counters are per (device, stream, batch size)
    and persistent: calls on different streams may overlap, and a captured call keeps its buffers."""
    lib = _lib.load()
    rgb, gathered = _f32(rgb.detach(), "rgb"), _f32(gathered, "gathered")
    B, dev = rgb.shape[0], rgb.device
    if len(biases) != 7 or packed.numel() != int(lib.tp_feat_chain_packed_floats()) or packed.dtype != torch.float32:
        raise ValueError("feat_chain: packed weights (feat_chain_pack) and the seven biases of VGG19 features[:15] expected")
    for b, co in zip(biases, (64, 64, 128, 128, 256, 256, 256)):
        if tuple(b.shape) != (co,) or b.dtype != torch.float32:
            raise ValueError("feat_chain: bias %s does not fit VGG19 features[:15]" % (tuple(b.shape),))
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream, B)
    scratch = _feat_chain_scratch.get(key)
    if scratch is None:
        n_cnt = C.c_int64(0)
        n_ws = lib.tp_feat_chain_workspace(B, int(hw[0]), int(hw[1]), C.byref(n_cnt))
        if n_ws < 0:
            check(-1, "tp_feat_chain_workspace")
        if torch.cuda.is_current_stream_capturing():
            raise _lib.TexposeLibraryError("tp_feat_chain: its workspace must exist before a hipGraph capture (run one eager step first)")
        scratch = _feat_chain_scratch[key] = (torch.empty(int(n_ws), device=dev), torch.zeros(int(n_cnt.value), dtype=torch.int32, device=dev))
    ws, cnt = scratch
    a = _lib.FeatChainArgs()
    f = _feat_args(rgb, gathered, mean, std)
    a.rgb, a.gathered, a.B, a.H, a.W, a.n_channels = f.rgb, f.gathered, B, int(hw[0]), int(hw[1]), f.n_channels
    a.c_image, a.c_image_syn, a.c_mask, a.c_mask_syn = f.c_image, f.c_image_syn, f.c_mask, f.c_mask_syn
    for c in range(3):
        a.mean[c], a.std[c] = f.mean[c], f.std[c]
    a.packed = packed.data_ptr()
    for l in range(7):
        a.bias[l] = biases[l].data_ptr()
    a.w2, a.scale = float(w2), float(scale)
    loss3, g_rgb = torch.empty(3, device=dev), torch.empty_like(rgb)
    a.loss, a.g_rgb = loss3.data_ptr(), g_rgb.data_ptr()
    a.workspace, a.workspace_floats, a.counters, a.n_counters = ws.data_ptr(), ws.numel(), cnt.data_ptr(), cnt.numel()
    check(lib.tp_feat_chain(C.byref(a), _stream()), "tp_feat_chain")
    return loss3, g_rgb


@_on_tensor_device
def disc_inputs(rgb: Tensor, gathered: Tensor, hw, geo: bool, stacked: bool = False):
    """(real, fake) [B, 3 or 9, h, w] of the discriminator step from the render output and the gathered patches (K13).
    ``stacked``: `real` is returned as the first half of a [2B, ...] buffer (returned in its place) whose second half the
    explicit discriminator-step schedule fills with the R1 cotangent: its first weight gradient then sums both pairs in one launch."""
    lib = _lib.load()
    rgb, gathered = _f32(rgb.detach(), "rgb"), _f32(gathered, "gathered")
    B, P = rgb.shape[0], rgb.shape[1]
    if gathered.numel() != B * 14 * P:
        raise ValueError("disc_inputs: gathered [B,14,h,w] expected")
    nc = 9 if geo else 3
    real, fake = torch.empty(2 * B if stacked else B, nc, hw[0], hw[1], device=rgb.device), torch.empty(B, nc, hw[0], hw[1], device=rgb.device)
    check(lib.tp_disc_inputs(rgb.data_ptr(), gathered.data_ptr(), B, P, int(bool(geo)), real.data_ptr(), fake.data_ptr(), _stream()),
          "tp_disc_inputs")
    return real, fake


@_on_tensor_device
def fake_patch_bwd(g_fake: Tensor, B: int, P: int) -> Tensor:
    """g_rgb [B,P,3] from the cotangent of the fake stack [B,nc,h,w] (channels 0..2, transposed)."""
    lib = _lib.load()
    g_fake = _f32(g_fake, "g_fake")
    g_rgb = torch.empty(B, P, 3, device=g_fake.device)
    check(lib.tp_fake_patch_bwd(g_fake.data_ptr(), B, P, g_fake.shape[1], g_rgb.data_ptr(), _stream()), "tp_fake_patch_bwd")
    return g_rgb


@_on_tensor_device
def feat_pair_loss_fwd(feat: Tensor, w2: float) -> Tensor:
    """feat [4B,...] = features of [fake1 | fake2 | real1 | real2] -> [l1 + w2 l2, l1, l2] (one launch)."""
    lib = _lib.load()
    feat = _f32(feat, "feat")
    out = torch.empty(3, device=feat.device)
    check(lib.tp_feat_pair_loss_fwd(feat.data_ptr(), feat.numel() // 4, float(w2), out.data_ptr(), _stream()), "tp_feat_pair_loss_fwd")
    return out


@_on_tensor_device
def feat_pair_loss_bwd(feat: Tensor, w2: float, g: Tensor) -> Tensor:
    lib = _lib.load()
    g = _f32(g, "g")
    g_feat = torch.empty_like(feat)
    check(lib.tp_feat_pair_loss_bwd(feat.data_ptr(), feat.numel() // 4, float(w2), g.data_ptr(), g_feat.data_ptr(), _stream()),
          "tp_feat_pair_loss_bwd")
    return g_feat


@_on_tensor_device
def sumsq_mean_fwd(g: Tensor) -> Tensor:
    """sum(g^2) / B for g [B,...] -> 0-dim tensor."""
    lib = _lib.load()
    g = _f32(g, "g")
    out = torch.empty((), device=g.device)
    check(lib.tp_sumsq_mean_fwd(g.data_ptr(), g.numel(), g.shape[0], out.data_ptr(), _stream()), "tp_sumsq_mean_fwd")
    return out


@_on_tensor_device
def sumsq_mean_bwd(g: Tensor, cot: Tensor) -> Tensor:
    lib = _lib.load()
    g, cot = _f32(g, "g"), _f32(cot, "cot")
    out = torch.empty_like(g)
    check(lib.tp_sumsq_mean_bwd(g.data_ptr(), g.numel(), g.shape[0], cot.data_ptr(), out.data_ptr(), _stream()), "tp_sumsq_mean_bwd")
    return out


@_on_tensor_device
def maxpool2_fwd(x: Tensor):
    """MaxPool2d(2, 2) of x [N,C,H,W] (H, W even) -> (y [N,C,H/2,W/2], arg uint8: window position of the maximum)."""
    lib = _lib.load()
    x = _f32(x, "x")
    N, Cc, H, W = x.shape
    y = torch.empty(N, Cc, H // 2, W // 2, device=x.device)
    arg = torch.empty(N, Cc, H // 2, W // 2, device=x.device, dtype=torch.uint8)
    check(lib.tp_maxpool2_fwd(x.data_ptr(), N * Cc, H, W, y.data_ptr(), arg.data_ptr(), _stream()), "tp_maxpool2_fwd")
    return y, arg


@_on_tensor_device
def maxpool2_bwd(gy: Tensor, arg: Tensor) -> Tensor:
    lib = _lib.load()
    gy = _f32(gy, "gy")
    N, Cc, oh, ow = gy.shape
    gx = torch.empty(N, Cc, 2 * oh, 2 * ow, device=gy.device)
    check(lib.tp_maxpool2_bwd(gy.data_ptr(), arg.data_ptr(), N * Cc, 2 * oh, 2 * ow, gx.data_ptr(), _stream()), "tp_maxpool2_bwd")
    return gx


@_on_tensor_device
def sumsq_mean_fwd_bwd(g: Tensor, w: float, out_g: Optional[Tensor] = None):
    """([sum(g^2) / B, w sum(g^2) / B], 2 w g / B) for g [B, ...] in one launch: value, weighted value (what the reference logs) and
    weighted gradient of the R1 penalty (K16)."""
    lib = _lib.load()
    g = _f32(g, "g")
    out, og = torch.empty(2, device=g.device), _out_like(out_g, g)
    check(lib.tp_sumsq_mean_fwd_bwd(g.data_ptr(), g.numel(), g.shape[0], float(w), out.data_ptr(), og.data_ptr(), _stream()),
          "tp_sumsq_mean_fwd_bwd")
    return out, og


@_on_tensor_device
def gan_disc_losses(d_real: Tensor, d_fake: Tensor, w_real: float, w_fake: float, g_real_out: Optional[Tensor] = None,
                    g_fake_out: Optional[Tensor] = None):
    """Both GAN-loss terms of the discriminator step and their weighted cotangents in one launch (K16):
    -> (out2 = [bce(d_real, 1), bce(d_fake, 0)], g_real, g_fake)."""
    lib = _lib.load()
    d_real, d_fake = _f32(d_real, "d_real"), _f32(d_fake, "d_fake")
    if d_real.numel() != d_fake.numel():
        raise ValueError("gan_disc_losses: d_real and d_fake must have the same number of elements")
    out2 = torch.empty(2, device=d_real.device)
    gr, gf = _out_like(g_real_out, d_real), _out_like(g_fake_out, d_fake)
    check(lib.tp_gan_disc_losses(d_real.data_ptr(), d_fake.data_ptr(), d_real.numel(), float(w_real), float(w_fake), out2.data_ptr(),
                                 gr.data_ptr(), gf.data_ptr(), _stream()), "tp_gan_disc_losses")
    return out2, gr, gf


_pending_total = {}          # "job": a loss total + gate waiting for the next nerf_losses_bwd launch on its stream (weighted_sum(defer=True))


def flush_pending_total() -> None:
    """Launch a total handed over with weighted_sum(defer=True) that no nerf_losses_bwd launch has taken (tp_weighted_sum_flags)."""
    job = _pending_total.pop("job", None)
    if job is None:
        return
    flags, bad = job["flags"], job["flags"]["bad"]
    with torch.cuda.device(job["out"].device):
        check(_lib.load().tp_weighted_sum_flags(job["ptrs"], job["ws"], job["n"], job["out"].data_ptr(), _ptr(flags.get("status")), bad.data_ptr(),
                                                bad.numel(), int(flags.get("word_status", 0)), int(flags["word_finite"]),
                                                flags["snapshot"].data_ptr(), _ptr(flags.get("step_counter")), _stream()), "tp_weighted_sum_flags")


@_on_tensor_device
def weighted_sum(terms, weights, flags=None, defer: bool = False) -> Tensor:
    """sum_k weights[k] * terms[k] for 0-dim float32 device tensors and host floats, one launch.  ``flags`` = dict(bad,
    word_finite, snapshot[, status, word_status, step_counter]): `step_flags` on the result in the same launch
    (tp_weighted_sum_flags); ``step_counter`` (int64 [1]) is incremented by it.
    ``defer`` (with flags): nothing is launched -- the next `nerf_losses_bwd` launch on this stream carries the job
    (tp_nerf_losses_bwd_total); the caller runs `flush_pending_total()` behind the backward pass in case there was none."""
    lib = _lib.load()
    n = len(terms)
    ts = [_f32(t.detach(), "term") for t in terms]
    ptrs = (C.c_void_p * n)(*[t.data_ptr() for t in ts])
    ws = (C.c_float * n)(*[float(w) for w in weights])
    out = torch.empty((), device=ts[0].device)
    if defer and flags is not None:
        flush_pending_total()
        _pending_total["job"] = dict(ptrs=ptrs, ws=ws, n=n, out=out, flags=flags, stream=_stream(), keep=ts)
        return out
    if flags is None:
        check(lib.tp_weighted_sum(ptrs, ws, n, out.data_ptr(), _stream()), "tp_weighted_sum")
    else:
        bad = flags["bad"]
        check(lib.tp_weighted_sum_flags(ptrs, ws, n, out.data_ptr(), _ptr(flags.get("status")), bad.data_ptr(), bad.numel(),
                                        int(flags.get("word_status", 0)), int(flags["word_finite"]), flags["snapshot"].data_ptr(),
                                        _ptr(flags.get("step_counter")), _stream()), "tp_weighted_sum_flags")
    return out


@_on_tensor_device
def latent_rows_fwd(w_trans: Tensor, w_light: Tensor, idx: Tensor, idx_copy: Optional[Tensor] = None, defer: bool = False):
    """``idx_copy`` (int64 [B], optional): the launch also writes idx there (a private copy for the backward).  ``defer``: nothing is
    launched; returns (out_trans, out_light, job) -- the ray-generation launch given the job fills them (`raygen(..., rows=job)`)."""
    lib = _lib.load()
    w_trans, w_light = _f32(w_trans, "w_trans"), _f32(w_light, "w_light")
    idx = idx.to(torch.int64).contiguous()
    B = idx.numel()
    ot, ol = torch.empty(B, w_trans.shape[1], device=idx.device), torch.empty(B, w_light.shape[1], device=idx.device)
    if idx_copy is not None and not (idx_copy.dtype == torch.int64 and idx_copy.is_contiguous() and idx_copy.numel() == B and idx_copy.device == idx.device):
        raise _lib.TexposeLibraryError("latent_rows_fwd: idx_copy must be a contiguous int64 device tensor of idx's length")
    if defer:
        return ot, ol, dict(w_trans=w_trans, w_light=w_light, idx=idx, out_trans=ot, out_light=ol, idx_copy=idx_copy)
    check(lib.tp_latent_rows_fwd(w_trans.data_ptr(), w_light.data_ptr(), idx.data_ptr(), B, w_trans.shape[1], w_light.shape[1], ot.data_ptr(),
                                 ol.data_ptr(), _ptr(idx_copy), _stream()), "tp_latent_rows_fwd")
    return ot, ol


@_on_tensor_device
def latent_rows_bwd(g_trans: Tensor, g_light: Tensor, idx: Tensor, n_rows: int):
    lib = _lib.load()
    g_trans, g_light = _f32(g_trans, "g_trans"), _f32(g_light, "g_light")
    if idx.dtype != torch.int64 or not idx.is_contiguous() or not idx.is_cuda:
        raise _lib.TexposeLibraryError("latent_rows_bwd: idx must be the contiguous int64 device tensor the forward used (got %s%s)"
                                       % (idx.dtype, "" if idx.is_contiguous() else ", non-contiguous"))
    B = idx.numel()
    gwt, gwl = torch.empty(n_rows, g_trans.shape[1], device=idx.device), torch.empty(n_rows, g_light.shape[1], device=idx.device)
    check(lib.tp_latent_rows_bwd(g_trans.data_ptr(), g_light.data_ptr(), idx.data_ptr(), B, int(n_rows), g_trans.shape[1], g_light.shape[1],
                                 gwt.data_ptr(), gwl.data_ptr(), _stream()), "tp_latent_rows_bwd")
    return gwt, gwl


@_on_tensor_device
def step_flags(total: Tensor, bad: Tensor, word_finite: int, snapshot: Tensor, status: Optional[Tensor] = None, word_status: int = 0) -> None:
    """bad[word_status] |= status & 1; bad[word_finite] |= !isfinite(total); snapshot = bad (K13 tp_step_flags, one launch)."""
    lib = _lib.load()
    check(lib.tp_step_flags(_ptr(status), total.data_ptr(), bad.data_ptr(), bad.numel(), int(word_status), int(word_finite),
                            snapshot.data_ptr(), _stream()), "tp_step_flags")


def stamp(slots: torch.Tensor, i: int) -> None:
    """Diagnostic: the device clock (100 MHz ticks) into ``slots[i]`` (int64 device tensor), one launch in stream order."""
    assert slots.dtype == torch.int64 and slots.is_cuda and 0 <= i < slots.numel()
    check(_lib.load().tp_stamp(slots.data_ptr() + 8 * i, _stream()), "tp_stamp")


def capture_node_count(stream=None):
    """Number of nodes (kernel launches, fills, copies) recorded so far in the hipGraph that ``stream`` (default: the current one) is
    capturing into, or None when it is not capturing (tp_capture_node_count: hipStreamGetCaptureInfo_v2 + hipGraphGetNodes on the HIP
    runtime the library is linked against -- the one this process already runs on).  The trainers report it per captured graph
    (`launch_counts`: the launches of one training iteration)."""
    stream = stream or torch.cuda.current_stream()
    n = int(_lib.load().tp_capture_node_count(stream.cuda_stream))
    return None if n < 0 else n


@_on_tensor_device
def grad_pack(grads, flat: Tensor, scale: float, words: Optional[Tensor] = None, tail: Optional[Tensor] = None) -> None:
    """The gradients of one optimiser step into the flat all-reduce buffer: flat[concatenation] = scale * grads[k] (``None`` entries:
    ``numel`` zeros -- pass (None, numel)), and the gate words into the sticky tail (K13 tp_grad_pack, one launch per 32 tensors).
    ``grads``: tensors, or (None, numel) pairs; ``words`` int32 device words, ``tail`` float32 view behind the gradients in ``flat``."""
    lib = _lib.load()
    rows, keep = [], []
    for g in grads:
        if isinstance(g, tuple):
            rows.append((None, int(g[1])))
            continue
        g = _f32(g, "grad")
        keep.append(g)
        rows.append((g.data_ptr(), g.numel()))
    if not flat.is_contiguous() or flat.dtype != torch.float32 or flat.numel() < sum(n for _, n in rows):
        raise _lib.TexposeLibraryError("grad_pack: the flat buffer must be contiguous float32 and hold every gradient")
    if words is not None and (words.dtype != torch.int32 or tail is None or tail.dtype != torch.float32 or tail.numel() < words.numel()):
        raise _lib.TexposeLibraryError("grad_pack: int32 gate words need a float32 tail of at least their length")
    off = 0
    for i0 in range(0, len(rows), _lib.GRAD_PACK_MAX_TENSORS):
        chunk = rows[i0:i0 + _lib.GRAD_PACK_MAX_TENSORS]
        ptrs = (C.c_void_p * len(chunk))(*[p for p, _ in chunk])
        numel = (C.c_int64 * len(chunk))(*[n for _, n in chunk])
        last = i0 + _lib.GRAD_PACK_MAX_TENSORS >= len(rows)
        n_words = 0 if (tail is None or not last) else (words.numel() if words is not None else tail.numel())
        check(lib.tp_grad_pack(ptrs, numel, len(chunk), flat.data_ptr() + 4 * off, float(scale), _ptr(words) if last else None, n_words,
                               _ptr(tail) if last else None, _stream()), "tp_grad_pack")
        off += sum(n for _, n in chunk)


def clock_probe(windows: int = 16, window_us: int = 5000) -> torch.Tensor:
    """Diagnostic: launch the clock sampler on the CURRENT stream (use a side stream, then launch the load on another one); returns
    the int64 device tensor [windows, 2] it fills with (shader cycles, 100-MHz ticks) per window -- read it after a synchronise;
    `clock_ghz_from_probe` folds it."""
    out = torch.zeros(windows, 2, dtype=torch.int64, device="cuda")
    check(_lib.load().tp_clock_probe(out.data_ptr(), int(windows), int(window_us), _stream()), "tp_clock_probe")
    return out


def clock_ghz_from_probe(out: torch.Tensor) -> float:
    """Median shader clock in GHz over the probe's windows (the first and the last one left out when there are more than four)."""
    w = out.cpu().double()
    ghz = (w[:, 0] / w[:, 1].clamp(min=1)) * 0.1
    if len(ghz) > 4:
        ghz = ghz[1:-1]
    return float(ghz.median())


def step_inputs(copies, scalars=(), words=None, words_host=None) -> None:
    """The per-iteration host -> device state of a replayed step in ONE launch (K13 tp_step_inputs): ``copies`` = [(dst, src)]
    device tensors of equal byte size (the batch into the static inputs), ``scalars`` = [(0-dim float32 device tensor, value)],
    ``words`` (int32 device tensor) copied out to ``words_host`` (pinned int32 host tensor of the same length)."""
    lib = _lib.load()
    copies, scalars = list(copies), list(scalars)
    if len(scalars) > _lib.STEP_INPUTS_MAX_SCALARS:
        raise ValueError("step_inputs: too many scalars")
    dev = (copies[0][0] if copies else scalars[0][0] if scalars else words).device
    with torch.cuda.device(dev):
        first = True
        for i0 in range(0, max(len(copies), 1), _lib.STEP_INPUTS_MAX_COPIES):
            chunk = copies[i0:i0 + _lib.STEP_INPUTS_MAX_COPIES]
            arr = (_lib.StepCopy * max(len(chunk), 1))()
            for a, (d, s) in zip(arr, chunk):
                nb = d.numel() * d.element_size()
                if not (d.is_contiguous() and s.is_contiguous() and s.numel() * s.element_size() == nb and d.device == s.device == dev):
                    raise _lib.TexposeLibraryError("step_inputs: copies need contiguous same-size tensors on one device")
                a.dst, a.src, a.bytes = d.data_ptr(), s.data_ptr(), nb
            sc = scalars if first else []
            sd = (C.c_void_p * max(len(sc), 1))(*[t.data_ptr() for t, _ in sc])
            sv = (C.c_float * max(len(sc), 1))(*[float(v) for _, v in sc])
            for t, _ in sc:
                if t.dtype != torch.float32 or t.device != dev:
                    raise _lib.TexposeLibraryError("step_inputs: scalars are float32 tensors on the batch's device")
            w = words if first else None
            if w is not None and not (w.dtype == torch.int32 and words_host.dtype == torch.int32 and words_host.is_pinned()
                                      and words_host.numel() == w.numel()):
                raise _lib.TexposeLibraryError("step_inputs: gate words need an int32 device tensor and a pinned int32 host tensor")
            check(lib.tp_step_inputs(arr, len(chunk), sd, sv, len(sc), _ptr(w), None if w is None else words_host.data_ptr(),
                                     0 if w is None else w.numel(), _stream()), "tp_step_inputs")
            first = False


@_on_tensor_device
def adam_step(params, grads, exp_avgs, exp_avg_sqs, steps, lr, beta1: float, beta2: float, eps: float, gate: Optional[Tensor] = None) -> None:
    """torch.optim.Adam's update of all tensors in one launch per 32 (K13 tp_adam_step); ``steps``: 0-dim float tensors with
    the step count BEFORE this update; a gated one-wave launch inside the same call adds 1 to each afterwards."""
    lib = _lib.load()
    lr_dev = lr.data_ptr() if isinstance(lr, torch.Tensor) else None
    lr_host = 0.0 if isinstance(lr, torch.Tensor) else float(lr)
    rows = list(zip(params, grads, exp_avgs, exp_avg_sqs, steps))
    for i0 in range(0, len(rows), _lib.ADAM_MAX_TENSORS):
        chunk = rows[i0:i0 + _lib.ADAM_MAX_TENSORS]
        arr = (_lib.AdamTensor * len(chunk))()
        for a, (p, g, m, v, st) in zip(arr, chunk):
            if not (p.is_contiguous() and g.is_contiguous() and m.is_contiguous() and v.is_contiguous()
                    and p.dtype == g.dtype == m.dtype == v.dtype == st.dtype == torch.float32 and st.is_cuda):
                raise _lib.TexposeLibraryError("adam_step needs contiguous float32 tensors and device step counters")
            a.param, a.grad, a.exp_avg, a.exp_avg_sq, a.step, a.numel = p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), st.data_ptr(), p.numel()
        check(lib.tp_adam_step(arr, len(chunk), lr_dev, lr_host, float(beta1), float(beta2), float(eps), _ptr(gate),
                               gate.numel() if gate is not None else 0, _ticket(chunk[0][0].device, "adam"), _stream()), "tp_adam_step")


# ------------------------------------------------------------------------------------------ K14
def _head_args(W1, W2, W3, B, C_z, L, slope):
    a = _lib.DiscHeadArgs()
    H = W2.shape[0]
    if W1.shape != (H, C_z + 2 * L + 1) or W2.shape != (H, H) or W3.numel() != H:
        raise ValueError("disc_head: W1 [H,C+2L+1], W2 [H,H], W3 [1,H] expected")
    a.W1, a.W2, a.W3 = W1.data_ptr(), W2.data_ptr(), W3.data_ptr()
    a.B, a.C, a.L, a.H, a.slope = int(B), int(C_z), int(L), int(H), float(slope)
    return a


@_on_tensor_device
def disc_head_fwd(z: Tensor, scale: Tensor, W1: Tensor, W2: Tensor, W3: Tensor, L: int, slope: float):
    """-> (out [B], t0 [B,C+2L+1], t1 [B,H], t2 [B,H]): the scale-conditioned head of the PatchGAN in one launch."""
    lib = _lib.load()
    z, scale, W1, W2, W3 = (_f32(t, n) for t, n in ((z, "z"), (scale, "scale"), (W1, "W1"), (W2, "W2"), (W3, "W3")))
    B, C_z = z.shape
    a = _head_args(W1, W2, W3, B, C_z, L, slope)
    dev, H = z.device, W2.shape[0]
    out, t0, t1, t2 = (torch.empty(B, device=dev), torch.empty(B, C_z + 2 * L + 1, device=dev), torch.empty(B, H, device=dev),
                       torch.empty(B, H, device=dev))
    a.z, a.scale, a.out, a.t0, a.t1, a.t2 = z.data_ptr(), scale.data_ptr(), out.data_ptr(), t0.data_ptr(), t1.data_ptr(), t2.data_ptr()
    check(lib.tp_disc_head_fwd(C.byref(a), _stream()), "tp_disc_head_fwd")
    return out, t0, t1, t2


@_on_tensor_device
def disc_head_bwd(g_out: Tensor, t0: Tensor, t1: Tensor, t2: Tensor, W1: Tensor, W2: Tensor, W3: Tensor, C_z: int, L: int, slope: float,
                  weight_grads: bool = True, accumulate_into=None, gz_out: Optional[Tensor] = None):
    """-> (gz [B,C], gW1, gW2, gW3, e1, e2).  ``weight_grads=False``: data gradient only (gW1..3 = None);
    ``accumulate_into=(gW1, gW2, gW3)``: the weight gradients are ADDED to these tensors (and they are what is returned)."""
    lib = _lib.load()
    g_out, W1, W2, W3 = _f32(g_out, "g_out"), _f32(W1, "W1"), _f32(W2, "W2"), _f32(W3, "W3")
    B, H, dev = t1.shape[0], t1.shape[1], t1.device
    a = _head_args(W1, W2, W3, B, C_z, L, slope)
    gz, e1, e2 = _out_like(gz_out, t1, (B, C_z)), torch.empty(B, H, device=dev), torch.empty(B, H, device=dev)
    gW1 = gW2 = gW3 = None
    if accumulate_into is not None:
        gW1, gW2, gW3 = (_out_like(g, W) for g, W in zip(accumulate_into, (W1, W2, W3)))
        a.accumulate_gw = 1
    elif weight_grads:
        gW1, gW2, gW3 = torch.empty_like(W1), torch.empty_like(W2), torch.empty_like(W3)
    a.g_out, a.t0, a.t1, a.t2, a.e1, a.e2, a.out = (g_out.data_ptr(), t0.data_ptr(), t1.data_ptr(), t2.data_ptr(), e1.data_ptr(),
                                                    e2.data_ptr(), gz.data_ptr())
    a.gW1, a.gW2, a.gW3 = _ptr(gW1), _ptr(gW2), _ptr(gW3)
    check(lib.tp_disc_head_bwd(C.byref(a), _stream()), "tp_disc_head_bwd")
    return gz, gW1, gW2, gW3, e1, e2


@_on_tensor_device
def disc_head_bwd_bwd(c_gz: Tensor, g_out: Tensor, t0: Tensor, t1: Tensor, t2: Tensor, e1: Tensor, e2: Tensor, W1: Tensor, W2: Tensor,
                      W3: Tensor, L: int, slope: float):
    """cotangent c_gz [B,C] of the backward's gz -> (d/d g_out [B], d/d W1, d/d W2, d/d W3)."""
    lib = _lib.load()
    c_gz, g_out, W1, W2, W3 = _f32(c_gz, "c_gz"), _f32(g_out, "g_out"), _f32(W1, "W1"), _f32(W2, "W2"), _f32(W3, "W3")
    B, C_z = c_gz.shape
    a = _head_args(W1, W2, W3, B, C_z, L, slope)
    gg = torch.empty(B, device=c_gz.device)
    gW1, gW2, gW3 = torch.empty_like(W1), torch.empty_like(W2), torch.empty_like(W3)
    a.c_gz, a.g_out, a.t0, a.t1, a.t2, a.e1, a.e2, a.out = (c_gz.data_ptr(), g_out.data_ptr(), t0.data_ptr(), t1.data_ptr(),
                                                            t2.data_ptr(), e1.data_ptr(), e2.data_ptr(), gg.data_ptr())
    a.gW1, a.gW2, a.gW3 = gW1.data_ptr(), gW2.data_ptr(), gW3.data_ptr()
    check(lib.tp_disc_head_bwd_bwd(C.byref(a), _stream()), "tp_disc_head_bwd_bwd")
    return gg, gW1, gW2, gW3


# ------------------------------------------------------------------------------------------ K17
DISC_TAIL_MAX_ROWS = _lib.DISC_TAIL_MAX_ROWS
_tail_ws = {}                # (device index, stream) -> workspace tensor of the split-K partial sums


def disc_tail_eligible(a: Tensor, W0: Tensor, extra_rows: int = 0) -> bool:
    """The fused tail (K17) takes up to 16 rows (and 16 extra weight-gradient rows), K a multiple of 4, fp32 device tensors."""
    return (a.is_cuda and a.dtype == torch.float32 and a.dim() == 2 and a.shape[0] <= DISC_TAIL_MAX_ROWS and extra_rows <= DISC_TAIL_MAX_ROWS
            and a.shape[1] % 4 == 0 and W0.shape[1] == a.shape[1] and not knobs.K.no_disc_tail)


def _tail_args(W0, W1, W2, W3, M, L, slope):
    a = _lib.DiscTailArgs()
    N, K = W0.shape
    H = W2.shape[0]
    if W1.shape != (H, N + 2 * L + 1) or W2.shape != (H, H) or W3.numel() != H:
        raise ValueError("disc_tail: W0 [N,K], W1 [H,N+2L+1], W2 [H,H], W3 [1,H] expected")
    a.W0, a.W1, a.W2, a.W3 = W0.data_ptr(), W1.data_ptr(), W2.data_ptr(), W3.data_ptr()
    a.M, a.M2, a.K, a.N, a.L, a.H, a.slope = int(M), 0, int(K), int(N), int(L), int(H), float(slope)
    return a


def _tail_workspace(dev, N):
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream, N, _pair_slot())
    ws = _tail_ws.get(key)
    if ws is None:
        ws = _tail_ws[key] = torch.empty(int(_lib.load().tp_disc_tail_workspace_bytes(N)) // 4, device=dev)
    return ws


@_on_tensor_device
def disc_tail_fwd(a: Tensor, W0: Tensor, scale: Tensor, W1: Tensor, W2: Tensor, W3: Tensor, L: int, slope: float):
    """a [M,K] -> (out [M], t0 [M,N+2L+1], t1 [M,H], t2 [M,H]): full-map convolution + scale-conditioned head, one launch."""
    lib = _lib.load()
    a, W0, scale, W1, W2, W3 = (_f32(t, n) for t, n in ((a, "a"), (W0, "W0"), (scale, "scale"), (W1, "W1"), (W2, "W2"), (W3, "W3")))
    M, dev = a.shape[0], a.device
    q = _tail_args(W0, W1, W2, W3, M, L, slope)
    out, t0, t1, t2 = (torch.empty(M, device=dev), torch.empty(M, q.N + 2 * L + 1, device=dev), torch.empty(M, q.H, device=dev),
                       torch.empty(M, q.H, device=dev))
    ws = _tail_workspace(dev, q.N)
    q.a, q.scale, q.out, q.t0, q.t1, q.t2 = a.data_ptr(), scale.data_ptr(), out.data_ptr(), t0.data_ptr(), t1.data_ptr(), t2.data_ptr()
    q.workspace, q.ticket = ws.data_ptr(), _ticket(dev, "disc_tail%d" % _pair_slot())
    q._keep = (a, W0, scale, W1, W2, W3, out, t0, t1, t2, ws)
    _launch("tp_disc_tail_fwd", q)
    return out, t0, t1, t2


@_on_tensor_device
def disc_tail_bwd(g_out: Tensor, t0: Tensor, t1: Tensor, t2: Tensor, W0: Tensor, W1: Tensor, W2: Tensor, W3: Tensor, L: int, slope: float,
                  a: Optional[Tensor] = None, want_c_a: bool = True, want_gW0: bool = True, head_weight_grads: bool = True,
                  accumulate_into=None, want_e: bool = False, gz_out: Optional[Tensor] = None, gy2: Optional[Tensor] = None,
                  a2: Optional[Tensor] = None, c_a_out: Optional[Tensor] = None, inorm=None):
    """The tail's backward in one launch -> dict(c_a [M,K], gW0 [N,K], gW1, gW2, gW3, gz [M,N], e1, e2) (absent entries None).
    ``a`` [M,K]: the ladder output (needed for gW0); ``gy2`` [M2,N] / ``a2`` [M2,K]: a second (cotangent, input) pair of the same
    weight whose rows join gW0's sum; ``accumulate_into=(gW1, gW2, gW3)``: the head's weight gradients are ADDED to these.
    ``inorm`` = dict(xhat [M,C,h,w], rstd [M*C], addend=None, out=None): the InstanceNorm + LeakyReLU backward of the ladder's last stage
    (ops.inorm_lrelu_bwd) applied to c_a inside the launch -> res["c_z"] (shaped like xhat); c_a itself only with ``want_c_a``."""
    lib = _lib.load()
    g_out, W0, W1, W2, W3 = (_f32(t, n) for t, n in ((g_out, "g_out"), (W0, "W0"), (W1, "W1"), (W2, "W2"), (W3, "W3")))
    M, dev = t1.shape[0], t1.device
    q = _tail_args(W0, W1, W2, W3, M, L, slope)
    res = dict(c_a=None, gW0=None, gW1=None, gW2=None, gW3=None, gz=None, e1=None, e2=None, c_z=None)
    keep = []
    if want_c_a:
        res["c_a"] = _out_like(c_a_out, t1, (M, q.K))
    if inorm is not None:
        xhat, rstd = _f32(inorm["xhat"], "xhat"), _f32(inorm["rstd"], "rstd")
        in_P = xhat.shape[-2] * xhat.shape[-1]
        if xhat.numel() != M * q.K or rstd.numel() * in_P != M * q.K:
            raise ValueError("disc_tail_bwd: xhat / rstd of the last ladder stage expected")
        res["c_z"] = _out_like(inorm.get("out"), xhat)
        q.in_xhat, q.in_rstd, q.c_z, q.in_P = xhat.data_ptr(), rstd.data_ptr(), res["c_z"].data_ptr(), int(in_P)
        if inorm.get("addend") is not None:
            ad = _f32(inorm["addend"], "addend")
            keep.append(ad)
            q.in_addend = ad.data_ptr()
        keep += [xhat, rstd]
    if want_gW0:
        if a is None:
            raise ValueError("disc_tail_bwd: the weight gradient of the full-map convolution needs the ladder output")
        a = _f32(a, "a")
        res["gW0"] = torch.empty(q.N, q.K, device=dev)
        q.a = a.data_ptr()
        if gy2 is not None:
            gy2, a2 = _f32(gy2, "gy2"), _f32(a2, "a2")
            keep += [gy2, a2]
            q.gy2, q.a2, q.M2 = gy2.data_ptr(), a2.data_ptr(), gy2.shape[0]
    if accumulate_into is not None:
        res["gW1"], res["gW2"], res["gW3"] = (_out_like(g, W) for g, W in zip(accumulate_into, (W1, W2, W3)))
        q.accumulate_gw = 1
    elif head_weight_grads:
        res["gW1"], res["gW2"], res["gW3"] = torch.empty_like(W1), torch.empty_like(W2), torch.empty_like(W3)
    if want_e:
        res["e1"], res["e2"] = torch.empty(M, q.H, device=dev), torch.empty(M, q.H, device=dev)
    if gz_out is not None:
        res["gz"] = _out_like(gz_out, t1, (M, q.N))
    q.g_out, q.t0, q.t1, q.t2 = g_out.data_ptr(), t0.data_ptr(), t1.data_ptr(), t2.data_ptr()
    q.c_a, q.gW0, q.gW1, q.gW2, q.gW3 = (_ptr(res[k]) for k in ("c_a", "gW0", "gW1", "gW2", "gW3"))
    q.gz, q.e1, q.e2 = _ptr(res["gz"]), _ptr(res["e1"]), _ptr(res["e2"])
    q._keep = (g_out, t0, t1, t2, W0, W1, W2, W3, a, keep, dict(res))
    _launch("tp_disc_tail_bwd", q)
    return res


@_on_tensor_device
def disc_tail_bwd_bwd(c: Tensor, g_out: Tensor, t0: Tensor, t1: Tensor, t2: Tensor, e1: Tensor, e2: Tensor, W0: Tensor, W1: Tensor,
                      W2: Tensor, W3: Tensor, L: int, slope: float, want_gg: bool = False):
    """R1 second pass through the tail: cotangent c [M,K] of the first pass' data gradient -> (gW1, gW2, gW3[, d/d g_out])."""
    lib = _lib.load()
    c, g_out, W0, W1, W2, W3 = (_f32(t, n) for t, n in ((c, "c"), (g_out, "g_out"), (W0, "W0"), (W1, "W1"), (W2, "W2"), (W3, "W3")))
    M, dev = c.shape[0], c.device
    q = _tail_args(W0, W1, W2, W3, M, L, slope)
    gW1, gW2, gW3 = torch.empty_like(W1), torch.empty_like(W2), torch.empty_like(W3)
    gg = torch.empty(M, device=dev) if want_gg else None
    ws = _tail_workspace(dev, q.N)
    q.a, q.g_out, q.t0, q.t1, q.t2, q.e1, q.e2 = (c.data_ptr(), g_out.data_ptr(), t0.data_ptr(), t1.data_ptr(), t2.data_ptr(), e1.data_ptr(),
                                                  e2.data_ptr())
    q.gW1, q.gW2, q.gW3, q.out = gW1.data_ptr(), gW2.data_ptr(), gW3.data_ptr(), _ptr(gg)
    q.workspace, q.ticket = ws.data_ptr(), _ticket(dev, "disc_tail")
    check(lib.tp_disc_tail_bwd_bwd(C.byref(q), _stream()), "tp_disc_tail_bwd_bwd")
    return (gW1, gW2, gW3, gg) if want_gg else (gW1, gW2, gW3)


# ------------------------------------------------------------------------------------------ K15
SKINNY_MAX_ROWS = 256


@_on_tensor_device
def skinny_linear_fwd(x: Tensor, w: Tensor, out: Optional[Tensor] = None) -> Tensor:
    """x [M,K] @ w [N,K]^T -> [M,N] for a handful of rows (K15)."""
    lib = _lib.load()
    x, w = _f32(x, "x"), _f32(w, "w")
    y = _out_like(out, x, (x.shape[0], w.shape[0]))
    check(lib.tp_skinny_linear_fwd(x.data_ptr(), w.data_ptr(), y.data_ptr(), x.shape[0], w.shape[0], x.shape[1], _stream()),
          "tp_skinny_linear_fwd")
    return y


SKINNY_DGRAD_MAX_ROWS = 16


@_on_tensor_device
def skinny_linear_dgrad(gy: Tensor, w: Tensor, out: Optional[Tensor] = None) -> Tensor:
    """gy [M,N] @ w [N,K] -> [M,K]: K15's own kernel (tp_skinny_linear_dgrad) for up to 16 rows -- every batch size of BASELINE's
    configurations -- so that the discriminator pass contains no library kernel in its autograd form either (the captured default
    iteration never reaches this function: its explicit schedule uses the fused tail K17).  TP_SKINNY_DGRAD_MM=1 selects rocBLAS
    (0.7 % faster per autograd-form iteration: 671-677 vs 664-669 it/s, round 5); more than 16 rows always take it."""
    if gy.shape[0] > SKINNY_DGRAD_MAX_ROWS or knobs.K.skinny_dgrad_mm:
        return torch.mm(gy, w, out=out) if out is not None else torch.mm(gy, w)
    lib = _lib.load()
    gy, w = _f32(gy, "gy"), _f32(w, "w")
    gx = _out_like(out, gy, (gy.shape[0], w.shape[1]))
    check(lib.tp_skinny_linear_dgrad(gy.data_ptr(), w.data_ptr(), gx.data_ptr(), gy.shape[0], w.shape[0], w.shape[1], _stream()),
          "tp_skinny_linear_dgrad")
    return gx


@_on_tensor_device
def skinny_linear_wgrad(gy: Tensor, x: Tensor, out: Optional[Tensor] = None) -> Tensor:
    """gy [M,N]^T @ x [M,K] -> [N,K]."""
    lib = _lib.load()
    gy, x = _f32(gy, "gy"), _f32(x, "x")
    if x.shape[0] > SKINNY_MAX_ROWS:
        raise ValueError("skinny_linear_wgrad: at most %d rows" % SKINNY_MAX_ROWS)
    gw = _out_like(out, x, (gy.shape[1], x.shape[1]))
    check(lib.tp_skinny_linear_wgrad(gy.data_ptr(), x.data_ptr(), gw.data_ptr(), x.shape[0], gy.shape[1], x.shape[1], _stream()),
          "tp_skinny_linear_wgrad")
    return gw
