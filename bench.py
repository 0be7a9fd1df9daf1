#!/usr/bin/env python3
"""Headline benchmark: rendered rays/s of the ray-marching hot path at 480x640 x 128 samples.

    python bench.py --gpus N --steps K --warmup W
        N>1 without a torchrun environment: this process starts N fresh children (python -m torch.distributed.run, one
        rank per GPU) BEFORE it touches the GPU and relays rank 0's line; launched by torch.distributed.run it is a rank.
    python bench.py --config c5 [--gpus N]                 BASELINE config C5 (8 objects, mixed resolution, N=256):
                                                           tools/eval_multi_object.py

A step = one pass of the hot path over one full 480x640 image per GPU through the product API
(Graph.render_by_slices, mode='val': fused ray-gen + bounds + stratified samples -> fused
posenc + static/transient/light MLP -> per-ray composite), synthetic Duck-like scene, inputs
resident in HBM.  Images shard by batch across GPUs (weak scaling, no data-path collective: rays
are independent).  Rank 0 prints ONE JSON line with the contract fields plus
  roofline     : the dominant kernel (fused MLP, MFMA-bound), algorithmic FLOP / HIP-event time; its FLAT scalar members also
                 carry the clock held under the kernel (clock_ghz), the exact-fp32 kernel (exact_fp32_*), the HBM-bound kernels
                 (hbm_<kernel>_frac / _ms / _traffic_ratio) and the training iteration at the BASELINE size (train_c3_*)
  cpu_baseline : the CPU oracle ("port" of the reference path) timed on this box's host cores on a
                 bounded sample of the same workload
  summary      : the same figures once more, compact, as the LAST member of the line (what survives a truncated log).
The verbose per-leg objects come first in the line, the contract keys and `summary` last.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

H, W, N_SAMPLES = 480, 640, 128
MLP_FLOP_PER_SAMPLE = 1_821_184          # 2 x 910,592 MAC
# dense MFMA peaks (MI355X_MICROARCH.md): exact-fp32 v_mfma_f32_32x32x2_f32, and f16 v_mfma_f32_32x32x16_f16
MFMA_PEAK_TFLOPS = {"fp32": 157.3, "f16x3": 2500.0}
TRAIN_FLOP_PER_SAMPLE = 3_220_992        # recording forward (1,821,184) + backward of the two heads (SURVEY 8d / DESIGN 4)
ISSUED_PER_ALGORITHMIC = {"fp32": 1, "f16x3": 3}    # f16x3 issues hi*hi + hi*lo + lo*hi per product


def build_scene(device, seed):
    """Synthetic evaluation scene through the PRODUCT path only: numpy recipes for intrinsics / pose / box / weights
    (texpose_amd.synthetic) and per-pixel depth bounds from the HIP ray-gen + slab test.  Returns CPU tensors."""
    from texpose_amd import synthetic
    sc = synthetic.eval_scene(H, W, B=1, seed=seed)
    near, far = synthetic.scene_bounds(sc, H, W, device)
    sc["z_near"], sc["z_far"] = near.cpu(), far.cpu()
    params = synthetic.network_weights(0, bias_scale=0.0)
    rs = np.random.RandomState(1)
    emb_t = torch.from_numpy(rs.normal(size=(189, 16)).astype(np.float32))
    emb_l = torch.from_numpy(rs.normal(size=(189, 48)).astype(np.float32))
    return sc, params, emb_t, emb_l


def make_graph(device, params, emb_t, emb_l, precision=None):
    from texpose_amd.graph import Graph
    from texpose_amd.options import default_options
    opt = default_options(H=H, W=W, device=str(device))
    opt.nerf.sample_intvs = N_SAMPLES
    opt.batch_size = 1
    g = Graph(opt).to(device)
    g.nerf.load_state_dict({**g.nerf.state_dict(), **{k: v.to(device) for k, v in params.items()}})
    g.attach_latents(189, opt)
    with torch.no_grad():
        g.latent_vars_trans.weight.copy_(emb_t)
        g.latent_vars_light.weight.copy_(emb_l)
    if precision is not None:
        g.nerf.precision = precision
    g.eval()
    return g, opt


def host_topology():
    """(sockets, physical cores, hardware threads available to this process, model name) from /proc/cpuinfo."""
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores, sockets, model, phys, core = set(), set(), "unknown", None, None
    try:
        for ln in open("/proc/cpuinfo"):
            key, _, val = ln.partition(":")
            key, val = key.strip(), val.strip()
            if key == "model name" and model == "unknown":
                model = val
            elif key == "physical id":
                phys = val
                sockets.add(val)
            elif key == "core id":
                core = val
            elif key == "" and phys is not None and core is not None:
                cores.add((phys, core))
                phys = core = None
        if phys is not None and core is not None:
            cores.add((phys, core))
    except OSError:
        pass
    n_phys = min(len(cores), avail) if cores else avail
    return max(len(sockets), 1), n_phys, avail, model


def cpu_baseline(sc, params, emb_t, emb_l, min_chunks=20, chunk=2048):
    """CPU oracle (plain PyTorch restatement of the reference path; the ONLY place this file touches oracle/) on 2048-ray
    chunks of the same 480x640x128 image, as BASELINE.md section 3 plans it: the thread count is chosen ON the timed chunk
    size (a 512-ray warm-up + one timed 2048-ray chunk per candidate; all hardware threads is usually NOT the fastest for 256-wide
    GEMMs), then >= 20 chunks are timed with the best count (`value`, `cores`), and the all-physical-cores figure is
    reported beside it."""
    from oracle import texpose_oracle as O
    sockets, n_phys, avail, cpu_model = host_topology()
    dr = (sc["z_near"][:, :, None], sc["z_far"][:, :, None])
    centre_row = (H // 2) * W

    def run_chunk(i, n=chunk):
        idx = (torch.arange(centre_row + i * n, centre_row + (i + 1) * n)[None]) % (H * W)
        rand = torch.rand(1, n, N_SAMPLES, 1)
        t0 = time.perf_counter()
        O.render(params, emb_t, emb_l, sc["pose"], sc["intr"], idx, dr, None, "val", H, W, N_SAMPLES, rand=rand)
        return time.perf_counter() - t0

    trial = {}
    with torch.no_grad():
        for nt in sorted({t for t in (8, 16, 32, 64, n_phys // 2, n_phys) if 1 <= t <= avail}):
            torch.set_num_threads(nt)
            run_chunk(0, 512)                                  # warm-up (thread pool, allocator) for this thread count
            trial[nt] = run_chunk(1)                           # one timed 2048-ray chunk
        best = min(trial, key=trial.get)

        def timed(nt, n_chunks):
            torch.set_num_threads(nt)
            run_chunk(0)
            ts = [run_chunk(i) for i in range(1, n_chunks + 1)]
            return n_chunks * chunk / sum(ts), chunk / min(ts)

        mean_rate, best_rate = timed(best, min_chunks)
        all_cores = None
        if n_phys != best:
            r, rb = timed(n_phys, 5)
            all_cores = dict(value=r, best_chunk=rb, cores=n_phys, chunks=5)
        else:
            all_cores = dict(value=mean_rate, best_chunk=best_rate, cores=n_phys, chunks=min_chunks)
        threads_set = torch.get_num_threads()
    return dict(value=mean_rate, unit="rays/s", cores=best, kind="port", cpu_model=cpu_model, sockets=sockets,
                physical_cores=n_phys, hardware_threads=avail, best_chunk=best_rate, chunks=min_chunks,
                all_physical_cores=all_cores, torch_num_threads_last=threads_set,
                thread_trial={str(k): chunk / v for k, v in sorted(trial.items())},
                sample="%d chunks of %d rays x %d samples of the 480x640 image (1 warm chunk before them), torch %s CPU fp32, "
                       "%d threads = the fastest of %s on timed %d-ray chunks (rays/s per candidate in thread_trial); "
                       "all_physical_cores: the same with torch.set_num_threads(%d) on %d socket(s), %d chunks"
                       % (min_chunks, chunk, N_SAMPLES, torch.__version__, best, sorted(trial), chunk, n_phys, sockets,
                          all_cores["chunks"]))


def train_cpu_baseline(min_iters=10, B=4, patch=16, n=64, hw=128):
    """BASELINE.md section 3, second half: "Train: B=4, 16x16 patches, N=64, fwd+bwd, >= 10 iterations" on the host cores --
    the CPU oracle's render(mode='train') of config C3 with autograd through the two heads and the latent rows (the trunk
    runs without a graph, as in the reference: layers/nerf_static_transient_light.py:87-100), the photometric / uncertainty /
    transient terms and their backward (reference path: model/nerf_adapt_st_gan.py:108-127 with the feature and GAN terms
    off -- VGG / PatchGAN are not part of the oracle's render).  Thread count chosen on a timed iteration like cpu_baseline."""
    from oracle import texpose_oracle as O
    from texpose_amd.synthetic import network_weights, training_batch
    sockets, n_phys, avail, cpu_model = host_topology()
    params = network_weights(0)
    leaves = []
    for k, v in params.items():
        if k.startswith(("mlp_rgb.", "mlp_trans.")):
            v.requires_grad_(True)
            leaves.append(v)
    rs = np.random.RandomState(1)
    emb_t = torch.from_numpy(rs.normal(size=(189, 16)).astype(np.float32)).requires_grad_(True)
    emb_l = torch.from_numpy(rs.normal(size=(189, 48)).astype(np.float32)).requires_grad_(True)
    leaves += [emb_t, emb_l]
    var = training_batch(B, hw, hw, seed=0, device="cpu")
    dr = (var.z_near[:, :, None], var.z_far[:, :, None])
    weights = dict(render=0, uncert=0, trans_reg=-2)                # options/nerf_lm_adapt_gan.yaml:65-79

    def iteration(it):
        g = torch.Generator().manual_seed(it)
        u = torch.rand(3, B, generator=g)
        coords = O.patch_coords(patch, u[0], u[1], u[2], lo=0.25)[0]
        rand = torch.rand(B, patch * patch, n, 1, generator=g)
        t0 = time.perf_counter()
        out = O.render(params, emb_t, emb_l, var.pose, var.intr, coords, dr, var.idx, "train", hw, hw, n, rand=rand)
        gath = O.patch_gather(coords, var.image, var.image_syn, var.nocs_pred, var.normal_pred, var.obj_mask, var.mask_syn)
        total = O.summarize(O.nerf_losses(out["rgb"], out["uncert"], out["density"], gath), weights)
        t1 = time.perf_counter()
        grads = torch.autograd.grad(total, leaves)
        t2 = time.perf_counter()
        assert all(bool(torch.isfinite(x).all()) for x in grads)
        return t1 - t0, t2 - t1

    trial = {}
    for nt in sorted({t for t in (8, 16, 32, 64, n_phys // 2, n_phys) if 1 <= t <= avail}):
        torch.set_num_threads(nt)
        iteration(0)
        trial[nt] = sum(iteration(1))
    best = min(trial, key=trial.get)
    torch.set_num_threads(best)
    iteration(0)
    ts = [iteration(i) for i in range(2, 2 + min_iters)]
    fwd, bwd = sum(a for a, _ in ts) / len(ts), sum(b for _, b in ts) / len(ts)
    return dict(value=1.0 / (fwd + bwd), unit="iterations/s", cores=best, kind="port", fwd_s=fwd, bwd_s=bwd, iterations=min_iters,
                cpu_model=cpu_model, sockets=sockets, physical_cores=n_phys, hardware_threads=avail,
                thread_trial={str(k): 1.0 / v for k, v in sorted(trial.items())},
                sample="%d iterations (after 1 warm one) of the oracle's render(mode='train') fwd + autograd bwd at B=%d, %dx%d patches of "
                       "%dx%d crops, N=%d (%d samples), photometric / uncert / trans_reg terms; the feature and GAN terms (VGG, "
                       "PatchGAN) are NOT in this baseline, so it is a render-only lower bound of a reference iteration; torch %s CPU fp32, "
                       "%d threads = the fastest of %s" % (min_iters, B, patch, patch, hw, hw, n, B * patch * patch * n,
                                                           torch.__version__, best, sorted(trial)))


CSRC = os.path.join("texpose_amd", "csrc")
# which sources decide the traffic of which profiles/traffic.json entry (the PMC figures are committed measurements: bench.py
# cannot run rocprofv3 on itself, so every entry carries the sha of the sources it was measured on and a mismatch is reported
# as "traffic_stale": true instead of silently quoting a figure of another kernel)
_MLP_COMMON = ["mlp_mma.h", "mlp_layout.h", "tp_common.h", "mlp_pack.hip"]
TRAFFIC_SOURCES = {
    "f16x3": ["mlp_fwd_f16x3.hip", "gen_wide_asm.py"] + _MLP_COMMON,
    "fp32": ["mlp_fwd.hip", "gen_fp32_asm.py"] + _MLP_COMMON,
    "train_b32": ["mlp_fwd_f16x3.hip", "gen_wide_asm.py", "mlp_bwd.hip"] + _MLP_COMMON,
    "hbm_kernels.composite_fwd": ["composite.hip", "tp_common.h"],
    "hbm_kernels.composite_bwd": ["composite.hip", "tp_common.h"],
    "hbm_kernels.raygen": ["raygen.hip", "tp_common.h"],
    "hbm_kernels.patch_gather": ["patch_gather.hip", "tp_common.h"],
    "hbm_kernels.patch_gather_b32_p64": ["patch_gather.hip", "tp_common.h"],
}


def sources_sha16(entry):
    """sha256 (first 16 hex digits) over the sources named for a traffic.json entry, in the listed order."""
    import hashlib
    h = hashlib.sha256()
    for name in TRAFFIC_SOURCES[entry]:
        with open(os.path.join(REPO, CSRC, name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read() + b"\0")
    return h.hexdigest()[:16]


def traffic_entry(entry):
    """(entry dict of profiles/traffic.json, stale flag): stale = the sources the figure was measured on are not today's
    (None when the entry carries no sha yet)."""
    try:
        node = json.load(open(os.path.join(REPO, "profiles", "traffic.json")))
        for part in entry.split("."):
            node = node[part]
    except Exception:
        return None, None
    sha = node.get("sources_sha16")
    return node, (None if sha is None else sha != sources_sha16(entry))


def _train_traffic():
    node, stale = traffic_entry("train_b32")
    return (None, None) if node is None else (float(node["total_per_step"]), stale)


HBM_PEAK_GBS = 8000.0                    # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable by a float4 copy)


def hbm_bytes(kernel, rays, n=0, pixels=0):
    """ALGORITHMIC bytes per launch of the HBM-bound kernels:
      composite_fwd  40 B read (rgb 24, density 8, depth 4, uncert 4) + 8 B written (the two alphas; `prob` is never written)
                     per sample, 12 B read (ray) + 56 B written (14 per-ray outputs) per ray
      composite_bwd  the same 40 B + 12 B read again, 56 B of per-ray cotangents read, 36 B of per-sample gradients written
                     (rgb 24, density 8, uncert 4); alpha cotangents absent (the losses do not use them)
      raygen         8 B read (int64 pixel index) + 24 B (centre, ray) + 4 N B (depths) written per ray
      patch_gather   8 B of coordinates + 12 bilinear channels x 4 taps x 4 B + 2 nearest taps x 4 B read, 56 B written
                     per patch pixel."""
    if kernel == "composite_fwd":
        return rays * (n * 48 + 68)
    if kernel == "composite_bwd":
        return rays * (n * 76 + 68)
    if kernel == "raygen":
        return rays * (8 + 24 + 4 * n)
    if kernel == "patch_gather":
        return pixels * (8 + 12 * 4 * 4 + 2 * 4 + 56)
    raise KeyError(kernel)


def _event_ms(fn, reps):
    """Average GPU time of `fn` (one C-ABI launch) over `reps` launches REPLAYED from a hipGraph between two HIP events: no
    host work between the launches (an eager loop of a 6 us kernel measures the ~50 us Python / ctypes call)."""
    fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        for _ in range(reps):
            fn()                                           # (outputs dropped: the graph's pool reuses their memory)
    graph.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    graph.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def hbm_rooflines(device, in_situ):
    """HBM-side rooflines of the kernels around the MLP (north_star: "achieved HBM GB/s on ray-gen / composite"): ALGORITHMIC
    bytes / HIP-event time against the 8 TB/s HBM3E peak, at the C2 launch size (one 480x640x128 image) and, for the gather,
    at the training sizes.  `in_situ_ms`: the same kernel bracketed inside the timed render loop; `standalone_ms`: 10-20 launches
    in one hipGraph replay between two events (no host gaps).  `traffic`: fabric bytes per launch from the committed PMC passes
    (profiles/traffic.json: rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in separate runs of tools/hbm_kernels.py)."""
    from texpose_amd import ops, synthetic
    R, N = H * W, N_SAMPLES
    g = torch.Generator(device=device).manual_seed(0)
    rnd = lambda *s: torch.rand(*s, device=device, generator=g)
    ray = rnd(1, R, 3) + 0.5
    rgb, den, unc = rnd(1, R, N, 3, 2), rnd(1, R, N, 2) * 3, rnd(1, R, N, 1) + 0.1
    depth = torch.sort(rnd(1, R, N) * 2 + 7, dim=-1).values[..., None].contiguous()
    g_out = rnd(1, R, 14)
    sc = synthetic.eval_scene(H, W, B=1, seed=0)
    near, far = synthetic.scene_bounds(sc, H, W, device)
    intr, pose = sc["intr"].to(device), sc["pose"].to(device)
    idx = torch.arange(R, device=device)[None]
    out = {"peak": HBM_PEAK_GBS, "unit": "GB/s", "bound": "hbm",
           "note": "achieved = ALGORITHMIC bytes per launch (bench.hbm_bytes) / HIP-event time; traffic = PMC fabric bytes per launch"}

    def entry(name, kernel, ms, nbytes, workload, in_situ_ms=None):
        """`ms`: the kernel bracketed inside the timed render loop where it is part of it (composite fwd, ray-gen: what the
        product path sees, right behind / in front of the MLP kernel), else the graph-replayed figure; both are reported."""
        used = in_situ_ms if in_situ_ms else ms
        node, stale = traffic_entry("hbm_kernels." + name)
        e = {"kernel": kernel, "ms": used, "bytes": nbytes, "achieved": nbytes / (used * 1e-3) / 1e9, "workload": workload,
             "traffic": (node or {}).get("total"), "traffic_stale": stale, "standalone_ms": ms, "in_situ_ms": in_situ_ms,
             "timed": "in the timed render loop (HIP events around the launch)" if in_situ_ms else
                      "hipGraph replay of 10-20 launches between two HIP events"}
        e["frac"] = e["achieved"] / HBM_PEAK_GBS
        out[name] = e

    ms = _event_ms(lambda: ops.composite_fwd(ray, rgb, den, depth, unc, 0.05, per_sample=True, want_prob=False), 10)
    entry("composite_fwd", "composite_fwd_kernel", ms, hbm_bytes("composite_fwd", R, N), "480x640 rays x 128 samples, alphas written",
          in_situ.get("composite_fwd_ms"))
    ms = _event_ms(lambda: ops.composite_bwd(ray, rgb, den, depth, unc, g_out), 10)
    entry("composite_bwd", "composite_bwd_kernel", ms, hbm_bytes("composite_bwd", R, N), "480x640 rays x 128 samples")
    ms = _event_ms(lambda: ops.raygen(intr, pose, H=H, W=W, n_samples=N, ray_idx=idx, z_near=near, z_far=far,
                                      jitter=ops.JITTER_PHILOX, seed=1, offset=0), 20)
    entry("raygen", "raygen_kernel", ms, hbm_bytes("raygen", R, N), "480x640 rays x 128 depths, Philox jitter, bounds from maps",
          None)
    out["raygen"]["in_situ_ms"] = in_situ.get("raygen_ms")
    out["raygen"]["note"] = ("in_situ_ms brackets the FIRST launch of a step on an idle GPU (the previous image ended with the "
                            "host read of the range flag): it contains the host's launch latency, so the replayed figure is used")
    del rgb, den, unc, depth
    for name, B, p, hw in (("patch_gather", 4, 16, 128), ("patch_gather_b32_p64", 32, 64, 128)):
        var = synthetic.training_batch(B, hw, hw, seed=0, device=device)
        coords = rnd(B, p, p, 2) * 1.6 - 0.8
        fn = lambda: ops.patch_gather(coords, var.image, var.image_syn, var.nocs_pred, var.normal_pred, var.obj_mask, var.mask_syn)
        ms = _event_ms(fn, 20)
        entry(name, "patch_gather_kernel", ms, hbm_bytes("patch_gather", 0, pixels=B * p * p),
              "%d images x %dx%d patch pixels of %dx%d crops%s" % (B, p, p, hw, hw, " (C3 size: launch-latency sized)" if B == 4 else ""))
    return out


def train_kernel_times(device, B=32, reps=10):
    """(recording forward ms, backward ms) of the training MLP kernels at C4's per-launch size (B images x 256 rays x 64
    samples): tp_mlp_fwd(save) + tp_mlp_bwd replayed `reps` times from a hipGraph (their sum), split by eager brackets."""
    from texpose_amd import ops
    from texpose_amd.graph import Graph
    from texpose_amd.options import default_options
    from texpose_amd.synthetic import training_batch
    opt = default_options(H=128, W=128, device=str(device))
    opt.batch_size, opt.patch_size, opt.nerf.sample_intvs = B, 16, 64
    torch.manual_seed(0)
    g = Graph(opt).to(device)
    g.attach_latents(189, opt)
    var = training_batch(B, 128, 128, seed=0, device=device)
    coords, _ = g.patch_sampler(nbatch=B, patch_size=16, device=device)
    center, ray, _, _, depth = ops.raygen(var.intr, var.pose, H=128, W=128, n_samples=64, coords=coords, z_near=var.z_near,
                                          z_far=var.z_far, jitter=ops.JITTER_PHILOX, seed=1, offset=0)
    lt, ll = g.latent_vars_trans.weight[var.idx].detach(), g.latent_vars_light.weight[var.idx].detach()
    packed = g.nerf.packed_weights("f16x3")

    def fwd():
        return ops.mlp_forward(packed, lt, ll, center=center, ray=ray, depth=depth, save=True, precision="f16x3")

    rgb, den, unc, saved = fwd()
    g_rgb, g_den, g_unc = torch.randn_like(rgb) * 1e-3, torch.randn_like(den) * 1e-3, torch.randn_like(unc) * 1e-3

    def bwd():
        return ops.mlp_backward(g.nerf, lt, ll, saved, rgb, den, unc, g_rgb, g_den, g_unc, wgrad_precision="f16x3")

    # (1) the pair (recording forward, backward) x reps REPLAYED from one hipGraph between two HIP events: the kernels as
    # the captured training step runs them -- alternating, no host work and no event packets between them;
    # (2) eager alternating launches with an event after each give the forward : backward split of that time (their sum
    # comes out ABOVE the replayed step: every event record costs the queue a few us).
    pair_ms = _event_ms(lambda: (fwd(), bwd()), reps)
    torch.cuda.synchronize(device)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2 * reps + 1)]
    ev[0].record()
    for i in range(reps):
        fwd()
        ev[2 * i + 1].record()
        bwd()
        ev[2 * i + 2].record()
    torch.cuda.synchronize(device)
    f_ev = sum(ev[2 * i].elapsed_time(ev[2 * i + 1]) for i in range(reps)) / reps
    b_ev = sum(ev[2 * i + 1].elapsed_time(ev[2 * i + 2]) for i in range(reps)) / reps
    return pair_ms * f_ev / (f_ev + b_ev), pair_ms * b_ev / (f_ev + b_ev)


def run_leg(failed, name, fn):
    """An untimed leg: its result, or {"error": ...} with ``name`` appended to ``failed`` (-> "legs_failed" in the line and a
    non-zero exit code after the line is printed) and the traceback on stderr."""
    try:
        return fn()
    except Exception as exc:
        import traceback
        traceback.print_exc(file=sys.stderr)
        failed.append(name)
        return {"error": repr(exc)[:400]}


def captured_or_eager(measure, device, world, graphed=True, sync=None):
    """``measure(graphed)`` with the captured step, falling back to the eager loop when the capture raised (e.g. a collective
    that cannot be captured on this stack).  The fallback is a JOB-wide decision: a rank retrying alone would sit in
    collectives its peers never issue (they are still in the captured loop), so the ranks agree on "somebody failed" first
    (one MAX all-reduce) and then ALL take the eager retry.  Returns (result, note or None)."""
    full, err = None, None
    try:
        full = measure(graphed)
    except Exception as exc:
        if not graphed:
            raise
        err = exc
    failed = torch.tensor([0 if err is None else 1], device=device, dtype=torch.int32)
    if world > 1:
        torch.distributed.all_reduce(failed, op=torch.distributed.ReduceOp.MAX)
    if not int(failed):
        return full, None
    if sync is not None:
        sync()
    return measure(False), (repr(err)[:300] if err is not None else "another rank failed to capture the step")


def train_leg(device, rank, world):
    """BASELINE's metric has a second half, "train iters/sec" (config C3: full GAN loop, batch 4, 128x128 crops, 16x16
    patches, 64 samples per ray; C4 = the same per-GPU batch sharded over the GPUs with one RCCL all-reduce per
    optimiser step).  Measured OUTSIDE the rays/s timed region with tools/train_dp.measure (the training entry point's
    own loop); at N=1 also the nerf step alone (B=4, B=32) and a roofline of the training MLP kernels from HIP events
    around tp_mlp_fwd (recording) and tp_mlp_bwd (dgrad + wgrad + finalize) on the stream they run on."""
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import train_dp
    from texpose_amd import ops
    out = {"workload": "C3/C4: Duck-like synthetic crops 128x128, 16x16 patches, 64 samples/ray, 4 images per GPU, "
                       "hipGraph-replayed iteration (several GPUs: two replays with the gradient all-reduces between them); "
                       "random-init VGG19[:15] feature network (weights unavailable offline)"}
    # The captured two-branch step.  On several GPUs the trainer keeps the RCCL all-reduces OUT of the graphs: replay A (render,
    # losses, all backward passes), two eager stream-ordered collectives, replay B (optimiser steps) -- collectives inside a
    # replayed hipGraph are not supported (ProcessGroupNCCL's watchdog aborts on the captured work event).
    graphed = os.environ.get("TP_BENCH_TRAIN_EAGER", "0") != "1"
    full, note = captured_or_eager(lambda g: train_dp.measure(device, rank, world, global_batch=4 * world, iters=40, warm=4,
                                                              graphed=g, full=True), device, world, graphed,
                                   sync=torch.cuda.synchronize)
    if note is not None:
        out["graph_capture_error"] = note
    out["full_gan_loop"] = {k: full[k] for k in ("value", "ms_per_iter", "global_batch", "per_gpu_batch", "launch", "launches",
                                                 "launch_counts", "queues", "ranks_seen",
                                                 "recording_forward", "collective", "loop", "finite", "skipped_steps")}
    out["queues"] = full["queues"]           # the stream -> hardware-queue probe of the captured step (None: no probe in this form)
    out["ranks_seen"] = full["ranks_seen"]
    out["unit"] = "iterations/s"
    # HIP events around the step's two gradient all-reduces (FlatGradAllReducer.reduce x 2 between the two graph replays): the
    # xGMI figure of C4; None on one GPU (the single-graph form has no collective)
    out["collective_ms"] = full.get("collective_ms")
    if world == 1:
        for B in (4, 32):
            r = train_dp.measure(device, 0, 1, global_batch=B, iters=40, warm=4, graphed=True, full=False)
            out["nerf_step_b%d" % B] = {k: r[k] for k in ("value", "ms_per_iter", "global_batch", "launch", "recording_forward")}
        # the same two loops with the REFERENCE-PRECISION recording forward (exact fp32 MFMA products, fp32 backward kernels:
        # arch.mlp_train_precision = 'fp32'), like `exact_fp32_kernel` beside the rays/s value
        for name, kw in (("nerf_step_b4_fp32", dict(global_batch=4, full=False)), ("full_gan_loop_fp32", dict(global_batch=4, full=True))):
            r = train_dp.measure(device, 0, 1, iters=30, warm=4, graphed=True, train_precision="fp32", **kw)
            out[name] = {k: r[k] for k in ("value", "ms_per_iter", "global_batch", "launch", "recording_forward", "finite", "skipped_steps")}
        # kernel-level: the two C-ABI calls of the B=32 nerf step (recording forward; dgrad + wgrad + finalize) replayed 10x
        # from a hipGraph between two HIP events, i.e. timed as the captured step runs them (round 2 bracketed single calls
        # of an eager step: their sum left the replayed step 7 us for its other kernels)
        f_ms, b_ms = train_kernel_times(device)
        samples = 32 * 256 * 64
        # The denominator is the WHOLE replayed B=32 step (MLP kernels + ray-gen, composite fwd/bwd, losses, gathers, Adam, pack:
        # ~80 us of other kernels, profiles/r3/13): a lower bound of the kernels' own fraction that needs no bracket inside the
        # graph.  The pair replay below (the two C-ABI calls alone, back to back) comes out 1-2 % SLOWER than the step that
        # contains it -- nothing but MFMA + HBM-write work in the loop lowers the clock (profiles/r3/07) -- so it only splits
        # the time into forward : backward.
        step_ms = out["nerf_step_b32"]["ms_per_iter"]
        achieved = TRAIN_FLOP_PER_SAMPLE * samples / (step_ms * 1e-3) / 1e12
        out["roofline"] = {"kernels": "mlp_fwd_f16x3_kernel<recording> + mlp_dgrad_f16x3_asm_kernel + mlp_wgrad_f16x3_kernel + "
                                      "mlp_wgrad_finalize + finalize2", "bound": "mfma", "achieved": achieved, "peak": MFMA_PEAK_TFLOPS["f16x3"],
                           "unit": "TFLOP/s", "frac": achieved / MFMA_PEAK_TFLOPS["f16x3"],
                           "issued_frac": 3 * achieved / MFMA_PEAK_TFLOPS["f16x3"], "step_ms_replayed": step_ms,
                           "pair_replay": {"fwd_ms": f_ms, "bwd_ms": b_ms},
                           "samples_per_launch": samples, "flop_per_sample": TRAIN_FLOP_PER_SAMPLE, "traffic": _train_traffic()[0],
                           "traffic_stale": _train_traffic()[1],
                           "traffic_unit": "bytes per B=32 step over the three MLP kernels, L2<->fabric (profiles/traffic.json: "
                                           "train_b32; the weight gradient reads 7.7 GB: it is the HBM-bound one of the three)",
                           "note": "B=32 nerf step; ALGORITHMIC FLOP (recording forward + head backward) / time of the WHOLE hipGraph-"
                                   "replayed step (its other kernels, ~80 us, included: a lower bound); every product is three f16 "
                                   "MFMAs; pair_replay = tp_mlp_fwd + tp_mlp_bwd alone, 10 pairs replayed from a hipGraph, split by "
                                   "eager HIP-event brackets"}
        try:
            out["c4_form"] = c4_form_leg(device)
        except Exception as exc:                     # (reported in the line; the one-rank figures above stand on their own)
            import traceback
            traceback.print_exc(file=sys.stderr)
            out["c4_form"] = {"error": repr(exc)[:400]}
    return out


def c4_form_leg(device, iters=40):
    """The training step SEVERAL ranks run (config C4's per-GPU share: 4 images), timed on this ONE GPU: a 1-rank RCCL communicator
    with the gradient all-reduces forced on, i.e. the linear graphs with [gradients, pack] | real RCCL call | [optimiser] for both
    optimiser steps.  What the form costs against `full_gan_loop` (the one-rank form) before a byte crosses xGMI; `collective_ms` =
    HIP events around the two RCCL calls of an iteration."""
    import gc
    import socket
    import torch.distributed as dist
    import train_dp
    made = not dist.is_initialized()
    if made:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=device)
    try:
        r = train_dp.measure(device, 0, 1, global_batch=4, iters=iters, warm=4, graphed=True, full=True, force_collectives=True)
        keep = {k: r[k] for k in ("value", "ms_per_iter", "form", "launches", "launch_counts", "collective_ms", "collective_ms_by_step",
                                  "ranks_seen", "launch", "collective", "finite", "skipped_steps")}
        keep["note"] = ("ONE GPU, 1-rank RCCL communicator, collectives forced: the step several ranks run (per-GPU share of C4), "
                        "nothing crosses xGMI")
        return keep
    finally:
        gc.collect()
        torch.cuda.synchronize()
        if made:
            dist.destroy_process_group()


def eval_masked_leg(device, graph, opt, sc, images=3, mask_frac=0.10):
    """The path `evaluate.py` takes (reference model/nerf_adapt_st_gan.py:337-364, 652-680): render_by_slices(mode='eval_noalign')
    on the OBJECT-MASK pixels only (a centred disk covering ~10 % of the 480x640 image), scatter into the default-filled maps,
    then PSNR / SSIM of the static render against the masked image (Graph.evaluate_metrics = tp_eval_metrics).  Per image: one
    nonzero() host sync for the pixel list, one blocking read of the range flag, one blocking read of the two metric values
    (the reference reads them with .item() as well).  rays/s counts OBJECT rays."""
    from texpose_amd.options import AttrDict
    yy, xx = np.mgrid[0:H, 0:W]
    r2 = mask_frac * H * W / np.pi
    mask = torch.from_numpy((((yy - H / 2 + 0.5) ** 2 + (xx - W / 2 + 0.5) ** 2) < r2).astype(np.float32))[None].to(device)
    n_obj = int(mask.sum())
    pose, intr = sc["pose"].to(device), sc["intr"].to(device)
    dr = (sc["z_near"].to(device)[:, :, None], sc["z_far"].to(device)[:, :, None])
    image = torch.rand(1, 3, H, W, device=device)
    light_idx = torch.tensor(3, device=device)

    def one():
        with torch.no_grad():
            ret = graph.render_by_slices(opt, pose, intr=intr, depth_range=dr, object_mask=mask, sample_idx=light_idx, mode="eval_noalign")
            var = AttrDict(dict(ret))
            var.image, var.obj_mask = image, mask
            m = graph.evaluate_metrics(opt, var)
            return float(m.psnr), float(m.ssim)

    res = {}
    for per_sample in (True, False):                        # reference contract (per-sample maps filled) / per-ray maps only
        opt.render.per_sample = per_sample
        one()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(images):
            psnr, ssim = one()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / images
        res["per_sample_maps" if per_sample else "per_ray_maps_only"] = dict(value=n_obj / dt, unit="object rays/s", ms_per_image=dt * 1e3,
                                                                             psnr=psnr, ssim=ssim)
    opt.render.per_sample = True
    res.update(object_rays=n_obj, mask="centred disk, %.1f %% of 480x640" % (100.0 * n_obj / (H * W)), images=images,
               workload="render_by_slices(mode='eval_noalign') on object pixels + scatter into default-filled maps + evaluate_metrics "
                        "(PSNR / SSIM at 480x640), %d samples per ray" % N_SAMPLES,
               note="per_sample_maps fills density [1,HW,N,2] and the two alpha maps [1,HW,N] with their defaults for every image as the "
                    "reference does (:657-667: 630 MB of fills at this size); per_ray_maps_only is opt.render.per_sample=False")
    return res


_REAL_STDOUT = None


def protect_stdout():
    """The contract is ONE JSON line on stdout.  Libraries write to file descriptor 1 behind Python's back -- RCCL prints a version
    block (from C stdio, flushed at exit) when a communicator is created -- so everything but `emit()` is sent to stderr: fd 1 is
    pointed at fd 2 for the life of the process and the line goes to a private duplicate of the original stdout."""
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)


def emit(line):
    """Print the result line (a dict) on the process's original stdout."""
    out = _REAL_STDOUT if _REAL_STDOUT is not None else sys.stdout
    out.write(json.dumps(line) + "\n")
    out.flush()


def spawn_ranks(n_gpus, argv):
    """`python bench.py --gpus N` without a torchrun environment: start N fresh rank processes as CHILDREN
    (python -m torch.distributed.run, rendezvous on 127.0.0.1 with a free port), pass their output through and return the
    launcher's exit code (non-zero if any rank failed).  The parent never initialises HIP and never exec()s."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n_gpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train", action="store_true", help="skip the training leg (train iters/s)")
    ap.add_argument("--precision", choices=["f16x3", "fp32"], default="f16x3",
                    help="MLP arithmetic: f16x3 = split-fp16 products on the f16 matrix cores (fp32-grade accuracy, "
                         "default); fp32 = exact fp32 MFMA")
    ap.add_argument("--config", choices=["c2", "c5"], default="c2",
                    help="c2 (default): BASELINE's headline workload, one 480x640x128 image per GPU and step; c5: 8 objects, "
                         "mixed 240x320 / 480x640, 256 samples per ray, 64 images (tools/eval_multi_object.py)")
    ap.add_argument("--spawn-check", action="store_true",
                    help="launcher self-test without a GPU: the ranks meet in a gloo all-reduce and rank 0 prints the world size")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # driver-style call `python bench.py --gpus N`: nothing in this process has touched the GPU yet (importing torch
        # does not); start N fresh rank processes and relay rank 0's line.  Never exec: see spawn_ranks.
        raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("--gpus %d but the launcher started %d rank(s)" % (args.gpus, world))
    protect_stdout()
    if args.spawn_check and args.config == "c5":
        # the C5 entry point's sharding and aggregation with a stub renderer, on CPU under gloo (no GPU touched)
        import torch.distributed as dist
        sys.path.insert(0, os.path.join(REPO, "tools"))
        import eval_multi_object
        line = eval_multi_object.spawn_check(rank, world)
        if world > 1:
            dist.destroy_process_group()
        if rank == 0:
            emit({"spawn_check": world, "config": "c5", "n_gpus": line["n_gpus"],
                  "per_object_ms": [round(o["ms"], 1) for o in line["per_object"]],
                  "samples_all_ranks": line["roofline"]["samples_all_ranks"],
                  "parallelism": line["config"]["parallelism"]})
        return
    if args.spawn_check:
        import torch.distributed as dist
        if world > 1:
            dist.init_process_group("gloo")
        t = torch.tensor([float(rank + 1)])
        if world > 1:
            dist.all_reduce(t)
            dist.destroy_process_group()
        if rank == 0:
            emit({"spawn_check": world, "rank_sum": float(t)})
        return
    if args.config == "c5":
        sys.path.insert(0, os.path.join(REPO, "tools"))
        import eval_multi_object
        eval_multi_object.run(["--steps", str(max(1, min(args.steps, 2))), "--warmup", str(min(args.warmup, 1)),
                               "--precision", args.precision], emit=emit)
        return
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=device)

    from texpose_amd import dist as tdist
    ranks_seen = tdist.ranks_seen(device=device)                    # SUM all-reduce of ones: RCCL's own view of the job (1 rank: 1)

    sc, params, emb_t, emb_l = build_scene(device, seed=rank)       # one image per rank
    graph, opt = make_graph(device, params, emb_t, emb_l, args.precision)
    pose, intr = sc["pose"].to(device), sc["intr"].to(device)
    dr = (sc["z_near"].to(device)[:, :, None], sc["z_far"].to(device)[:, :, None])
    mask = torch.ones(1, H, W, device=device)

    # time the dominant kernel with HIP events on the stream it is launched on
    from texpose_amd import ops
    mlp_events, raygen_events, comp_events = [], [], []
    orig_mlp, orig_raygen, orig_comp = ops.mlp_forward, ops.raygen, ops.composite_fwd

    def timing_into(store, fn=None):
        fn = fn or orig_mlp

        def timed(*a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = fn(*a, **k)
            e1.record()
            store.append((e0, e1))
            return out
        return timed

    def step():
        with torch.no_grad():
            return graph.render_by_slices(opt, pose, intr=intr, depth_range=dr, object_mask=mask, sample_idx=None,
                                          mode="val")

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        ret = step()
    ops.mlp_forward = timing_into(mlp_events)
    ops.raygen, ops.composite_fwd = timing_into(raygen_events, orig_raygen), timing_into(comp_events, orig_comp)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ret = step()
    barrier()
    dt = time.perf_counter() - t0
    ops.mlp_forward, ops.raygen, ops.composite_fwd = orig_mlp, orig_raygen, orig_comp
    assert torch.isfinite(ret.rgb).all() and ret.rgb.shape == (1, H * W, 3)

    t = torch.tensor([dt], device=device, dtype=torch.float64)
    if world > 1:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    dt = float(t.item())
    mlp_ms = float(np.mean([a.elapsed_time(b) for a, b in mlp_events]))
    launches_per_step = len(mlp_events) / args.steps
    samples_per_launch = H * W * N_SAMPLES / launches_per_step

    def measured_traffic(precision, samples):
        """(fabric bytes per launch from the committed PMC profile, stale flag): bench.py cannot run rocprofv3 on itself; the
        entry names the sha of the kernel sources it was measured on (traffic_entry)."""
        node, stale = traffic_entry(precision)
        return (None, None) if node is None else (node["total"] * samples / (H * W * N_SAMPLES), stale)

    def roofline(precision, ms, samples):
        achieved = MLP_FLOP_PER_SAMPLE * samples / (ms * 1e-3) / 1e12
        peak = MFMA_PEAK_TFLOPS[precision]
        k = ISSUED_PER_ALGORITHMIC[precision]
        rb = precision == "f16x3" and ops.ray_bias_applies("f16x3", N_SAMPLES, False, True)
        if rb:
            # ray-bias variant: the 75 ray-constant input columns of mlp_rgb.0 and the 16 of mlp_trans.0 (23,296 of the 910,592 MACs
            # per sample) are contracted once per ray in fp32 by the pre-kernels -- inside the event window --, and the three narrow
            # output layers (9 rows x 256 = 2,304 MACs) run as fp32 dot products on the vector ALU: neither is on the matrix cores
            k = k * (1.0 - (23_296 + 2_304) / 910_592)
        return {"kernel": "mlp_fwd_exact_asm_kernel" if precision == "fp32" else
                ("mlp_fwd_f16x3_kernel<false, true> (+ rb_image_bias / rb_ray_bias pre-kernels, in the event window)" if rb else "mlp_fwd_f16x3_kernel"),
                "bound": "mfma",
                "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak, "traffic": measured_traffic(precision, samples)[0],
                "traffic_stale": measured_traffic(precision, samples)[1],
                "traffic_unit": "bytes/launch, L2<->fabric (FETCH_SIZE x2 + WRITE_SIZE, profiles/traffic.json)",
                "kernel_ms": ms, "samples_per_launch": samples, "flop_per_sample": MLP_FLOP_PER_SAMPLE,
                "mfma_issued_per_algorithmic": k, "issued_frac": k * achieved / peak,
                "note": "achieved = ALGORITHMIC FLOP / HIP-event time; f16x3 issues 3 f16 MFMAs per algorithmic product"}

    # secondary leg (outside the timed region, rank 0 only): the exact-fp32 kernel on the same image
    exact = None
    if rank == 0 and world == 1 and args.precision != "fp32":
        graph.nerf.precision = "fp32"
        step()
        ev = []
        ops.mlp_forward = timing_into(ev)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        step()
        torch.cuda.synchronize()
        dt1 = time.perf_counter() - t1
        ops.mlp_forward = orig_mlp
        graph.nerf.precision = args.precision
        ms1 = float(np.mean([a.elapsed_time(b) for a, b in ev]))
        exact = {"value": H * W / dt1, "unit": "rays/s", "roofline": roofline("fp32", ms1, H * W * N_SAMPLES / len(ev))}
    # secondary leg (outside the timed region): the same image with opt.render.per_sample = False -- alpha_static /
    # alpha_transient [HW,N] not materialised (evaluate_full / validate only read per-ray maps)
    per_ray = None
    if rank == 0:
        opt.render.per_sample = False
        step()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(3):
            r2 = step()
        torch.cuda.synchronize()
        dt2 = (time.perf_counter() - t1) / 3
        assert r2.alpha_static is None and torch.isfinite(r2.rgb).all()      # (stratified jitter: every call draws anew)
        per_ray = {"value": H * W / dt2, "unit": "rays/s", "ms_per_step": dt2 * 1e3,
                   "note": "opt.render.per_sample=False: per-sample alphas not written (prob is never written); MLP outputs "
                           "rgb/density/uncert per sample still pass through HBM between the MLP and the composite kernel"}
        opt.render.per_sample = True
        del r2
    # secondary leg (outside the timed region): the shader clock the chip HOLDS while the MLP kernel runs -- one sleeping wave on a
    # side stream samples s_memtime against the 100 MHz clock in 5 ms windows across one render (tp_clock_probe).  The MFMA peaks are
    # quoted at 2.4 GHz; this kernel is power-limited below that, and `frac` scales with the clock (it holds one SIMD slot of one CU
    # meanwhile, which is why it is NOT in the timed region).
    clock_ghz = None
    if rank == 0:
        def _clock():
            probe_stream = torch.cuda.Stream()
            torch.cuda.synchronize()
            with torch.cuda.stream(probe_stream):
                words = ops.clock_probe(windows=24, window_us=5000)
            step()
            torch.cuda.synchronize()
            return ops.clock_ghz_from_probe(words)
        try:
            clock_ghz = _clock()
        except Exception:
            import traceback
            traceback.print_exc(file=sys.stderr)
    ops.check_mlp_status(device)
    del ret
    torch.cuda.empty_cache()
    # The untimed legs: the rays/s line must come out whatever happens to them, but a leg that raised is NAMED in the line
    # ("legs_failed") and makes the process exit non-zero AFTER the line is printed -- a failed leg is never just an "error"
    # string inside an rc-0 run.
    legs_failed = []
    leg = lambda name, fn: run_leg(legs_failed, name, fn)

    # The train leg runs FIRST among the untimed legs: every stream a process creates takes the next hardware queue round-robin,
    # and the trainer's graphs are fastest when its own streams are the first ones after the timed region's (DESIGN section 9,
    # 'queues'); the roofline and masked-evaluation legs create timing streams of their own.
    train = None
    if not args.no_train:
        train = leg("train", lambda: train_leg(device, rank, world))
        if world == 1 and not args.no_cpu_baseline and "error" not in train:
            train["cpu_baseline"] = leg("train.cpu_baseline", train_cpu_baseline)
            if "error" not in train["cpu_baseline"]:
                train["gpu_over_cpu"] = {"full_gan_loop_vs_cpu_render_only": train["full_gan_loop"]["value"] / train["cpu_baseline"]["value"],
                                         "nerf_step_b4_vs_cpu": train["nerf_step_b4"]["value"] / train["cpu_baseline"]["value"]}

    hbm = None
    if rank == 0:
        hbm = leg("roofline_hbm", lambda: hbm_rooflines(device, dict(
            raygen_ms=float(np.mean([a.elapsed_time(b) for a, b in raygen_events])),
            composite_fwd_ms=float(np.mean([a.elapsed_time(b) for a, b in comp_events])))))
        torch.cuda.empty_cache()
    eval_masked = None
    if rank == 0 and world == 1:
        eval_masked = leg("eval_masked", lambda: eval_masked_leg(device, graph, opt, sc))
        torch.cuda.empty_cache()
    trained = None
    if rank == 0 and world == 1 and not args.no_train and args.precision == "f16x3":
        # untimed leg: the f16x3 kernel on a TRAINED network (500 product-trainer iterations) and with the trunk feature
        # scaled x4 / x16: rays/s, range-flag count, largest hidden activation, f16x3-vs-fp32 error (tools/trained_weights.py)
        def _trained():
            sys.path.insert(0, os.path.join(REPO, "tools"))
            import trained_weights
            return trained_weights.run(device, iters=500)
        trained = leg("trained_weights", _trained)
        torch.cuda.empty_cache()
    # a leg that failed on ANOTHER rank must show in rank 0's line and exit code too
    if world > 1:
        nf = torch.tensor([len(legs_failed)], device=device, dtype=torch.int32)
        torch.distributed.all_reduce(nf, op=torch.distributed.ReduceOp.MAX)
        if int(nf) and not legs_failed:
            legs_failed.append("a leg on another rank")

    if rank == 0:
        value = world * H * W * args.steps / dt
        rl = roofline(args.precision, mlp_ms, samples_per_launch)
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = leg("cpu_baseline", lambda: cpu_baseline(sc, params, emb_t, emb_l))
        # ---- compact, FLAT figures inside `roofline` (the driver's record keeps this object's scalar members): the clock, the
        # reference-arithmetic kernel, the HBM-bound kernels around the MLP, and the training iteration at the BASELINE size (C3 / C4's
        # per-GPU share); the full objects follow under their own keys.
        rl["clock_ghz"] = clock_ghz
        rl["clock_note"] = "shader clock held during the MLP kernel (tp_clock_probe, median of 5 ms windows); peaks are quoted at 2.4 GHz"
        if clock_ghz:
            rl["frac_of_peak_at_held_clock"] = rl["frac"] * 2.4 / clock_ghz
        if exact is not None:
            rl["exact_fp32_rays_per_s"] = exact["value"]
            rl["exact_fp32_frac"] = exact["roofline"]["frac"]
            rl["exact_fp32_kernel_ms"] = exact["roofline"]["kernel_ms"]
        if hbm is not None and "error" not in hbm:
            for name, short in (("composite_fwd", "composite_fwd"), ("composite_bwd", "composite_bwd"), ("raygen", "raygen"),
                                ("patch_gather_b32_p64", "gather")):
                e = hbm.get(name)
                if e:
                    rl["hbm_%s_frac" % short] = e["frac"]
                    rl["hbm_%s_ms" % short] = e["ms"]
                    rl["hbm_%s_standalone_frac" % short] = e["bytes"] / (e["standalone_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
                    rl["hbm_%s_traffic_ratio" % short] = (e["traffic"] / e["bytes"]) if e.get("traffic") else None
        if train is not None and "error" not in train:
            full = train["full_gan_loop"]
            samples_it = full["per_gpu_batch"] * 256 * 64
            tf = TRAIN_FLOP_PER_SAMPLE * samples_it / (full["ms_per_iter"] * 1e-3) / 1e12
            rl["train_c3_it_per_s"] = full["value"]
            rl["train_c3_ms_per_iter"] = full["ms_per_iter"]
            rl["train_c3_frac"] = tf / MFMA_PEAK_TFLOPS["f16x3"]
            rl["train_c3_launches"] = full.get("launches")
            if "nerf_step_b4" in train:
                rl["train_c3_nerf_step_it_per_s"] = train["nerf_step_b4"]["value"]
            c4 = train.get("c4_form")
            if c4 and "error" not in c4:
                # the several-rank form of the same iteration on this one GPU (1-rank RCCL communicator, collectives forced)
                rl["train_c4_form_it_per_s"] = c4["value"]
                rl["train_c4_form_launches"] = c4.get("launches")
                rl["train_c4_form_collective_ms"] = c4.get("collective_ms")
                rl["train_c4_form_over_c3"] = c4["value"] / full["value"]
        line = {}
        # the verbose legs FIRST, the contract keys and a compact summary LAST: the driver keeps the tail of stdout
        if hbm is not None:
            line["roofline_hbm"] = hbm
        if exact is not None:
            line["exact_fp32_kernel"] = exact
        if per_ray is not None:
            line["per_ray_outputs_only"] = per_ray
        if eval_masked is not None:
            line["eval_masked"] = eval_masked
        if trained is not None:
            line["trained_weights"] = trained
        if train is not None:
            line["train"] = train
        if cpu is not None:
            line["cpu_baseline"] = cpu
            if "error" not in cpu:
                line["gpu_over_cpu"] = value / cpu["value"]
        line["legs_failed"] = list(legs_failed)
        line.update({
            "metric": "rendered rays/sec (480x640x128 samples) + train iters/sec",
            "value": value,
            "unit": "rays/s",
            "n_gpus": world,
            "ranks_seen": ranks_seen,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            # BASELINE.md holds no published number for this metric (section 1); the figure beside it that north_star asks for is the
            # CPU path timed on this box's host cores in this very run (>= 20x is north_star's bar), so that ratio is reported here
            "vs_baseline": (value / cpu["value"]) if cpu is not None and "error" not in cpu else None,
            "vs_baseline_note": "no published reference number exists (BASELINE.md section 1): value / cpu_baseline.value of this run",
            "dtype": "f32" if args.precision == "fp32" else "f32 carried as 2xf16 (f16x3 products, f32 accumulate)",
            "data": "synthetic",
            "config": {"workload": "Duck-like synthetic scene 480x640, 128 samples/ray, batch=1 image per GPU, "
                                   "forward render (render_by_slices mode='val', all pixels), per-sample outputs "
                                   "materialised", "rays_per_step_per_gpu": H * W, "samples_per_ray": N_SAMPLES,
                       "mlp_precision": args.precision,
                       "parallelism": "images sharded across %d GPU(s), no collective" % world},
            "roofline": rl,
        })
        r3 = lambda v: None if v is None else float("%.4g" % v)
        line["summary"] = {
            "rays_per_s": r3(value), "ms_per_step": r3(dt / args.steps * 1e3), "n_gpus": world, "ranks_seen": ranks_seen,
            "mlp": {"frac": r3(rl["frac"]), "issued_frac": r3(rl["issued_frac"]), "kernel_ms": r3(rl["kernel_ms"]), "clock_ghz": r3(clock_ghz),
                    # algorithmic bytes of the MLP launch: 36 B of outputs + 4 B of depth per sample, 24 B of centre / ray per ray
                    "traffic_ratio": r3(rl["traffic"] / (samples_per_launch * (40.0 + 24.0 / N_SAMPLES))) if rl.get("traffic") else None},
            "exact_fp32": None if exact is None else {"rays_per_s": r3(exact["value"]), "frac": r3(exact["roofline"]["frac"])},
            "hbm": {k[4:]: r3(v) for k, v in rl.items() if k.startswith("hbm_") and (k.endswith("_frac") or k.endswith("_ms"))},
            "train_c3": None if "train_c3_it_per_s" not in rl else {
                "it_per_s": r3(rl["train_c3_it_per_s"]), "frac": r3(rl["train_c3_frac"]), "launches": rl["train_c3_launches"],
                "nerf_step_it_per_s": r3(rl.get("train_c3_nerf_step_it_per_s")),
                "queues": (train["full_gan_loop"].get("queues") or {}).get("concurrent"),
                "collective_ms": r3(train.get("collective_ms"))},
            "c4_form": None if "train_c4_form_it_per_s" not in rl else {
                "it_per_s": r3(rl["train_c4_form_it_per_s"]), "launches": rl["train_c4_form_launches"],
                "collective_ms": r3(rl["train_c4_form_collective_ms"]), "over_train_c3": r3(rl["train_c4_form_over_c3"])},
            "cpu": None if cpu is None or "error" in cpu else {"rays_per_s": r3(cpu["value"]), "cores": cpu["cores"],
                                                               "gpu_over_cpu": r3(value / cpu["value"])},
            "legs_failed": list(legs_failed),
        }
        emit(line)
    if world > 1:
        torch.distributed.destroy_process_group()
    if legs_failed:
        raise SystemExit(3)                        # (after the line: the driver's rc shows that an untimed leg raised)


if __name__ == "__main__":
    main()
