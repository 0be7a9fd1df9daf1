#!/usr/bin/env python3
"""Headline benchmark: rendered rays/s of the ray-marching hot path at 480x640 x 128 samples.

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

A step = one pass of the hot path over one full 480x640 image per GPU through the product API
(Graph.render_by_slices, mode='val': fused ray-gen + bounds + stratified samples -> fused
posenc + static/transient/light MLP -> per-ray composite), synthetic Duck-like scene, inputs
resident in HBM.  Images shard by batch across GPUs (weak scaling, no data-path collective: rays
are independent).  Rank 0 prints ONE JSON line with the contract fields plus
  roofline     : the dominant kernel (fused MLP, MFMA-bound), algorithmic FLOP / HIP-event time
  cpu_baseline : the CPU oracle ("port" of the reference path) timed on this box's host cores on a
                 bounded sample of the same workload.
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


def cpu_baseline(sc, params, emb_t, emb_l, budget_s=20.0, chunk=2048):
    """CPU oracle (plain PyTorch restatement of the reference path; the ONLY place this file touches oracle/) on
    2048-ray chunks of the same image.
    The thread count is picked by a short trial (all hardware threads is usually NOT the fastest for
    256-wide GEMMs); `cores` reports the count actually used."""
    from oracle import texpose_oracle as O
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    dr = (sc["z_near"][:, :, None], sc["z_far"][:, :, None])
    centre_row = (H // 2) * W

    def run_chunk(i, n=chunk):
        idx = (torch.arange(centre_row + i * n, centre_row + (i + 1) * n)[None]) % (H * W)
        rand = torch.rand(1, n, N_SAMPLES, 1)
        t0 = time.perf_counter()
        O.render(params, emb_t, emb_l, sc["pose"], sc["intr"], idx, dr, None, "val", H, W, N_SAMPLES, rand=rand)
        return time.perf_counter() - t0

    with torch.no_grad():
        best, best_t = None, float("inf")
        for nt in sorted({t for t in (8, 16, 32, 64, avail // 2, avail) if 1 <= t <= avail}):
            torch.set_num_threads(nt)
            run_chunk(0, 256)                                  # warm-up for this thread count
            t = run_chunk(1, 512)
            if t < best_t:
                best, best_t = nt, t
        torch.set_num_threads(best)
        run_chunk(0)                                           # warm-up chunk
        done, t_used, n_chunks = 0, 0.0, 0
        for i in range(1, 64):
            t_used += run_chunk(i)
            done += chunk
            n_chunks += 1
            if t_used > budget_s:
                break
    cpu_model = "unknown"
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                cpu_model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return dict(value=done / t_used, unit="rays/s", cores=best, kind="port", cpu_model=cpu_model,
                sample="%d chunks of %d rays x %d samples of the 480x640 image, torch %s CPU fp32, %d threads "
                       "(fastest of a short trial; %d hardware threads available)"
                       % (n_chunks, chunk, N_SAMPLES, torch.__version__, best, avail))


def _train_traffic():
    try:
        return float(json.load(open(os.path.join(REPO, "profiles", "traffic.json")))["train_b32"]["total_per_step"])
    except Exception:
        return None


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
    # replayed hipGraph have only ever run in a 1-rank group here (TP_COLLECTIVES_IN_GRAPH=1 opts in).
    graphed = os.environ.get("TP_BENCH_TRAIN_EAGER", "0") != "1"
    try:
        full = train_dp.measure(device, rank, world, global_batch=4 * world, iters=40, warm=4, graphed=graphed, full=True)
    except Exception as exc:                       # (e.g. a collective that cannot be captured on this stack): eager loop
        if not graphed:
            raise
        out["graph_capture_error"] = repr(exc)[:300]
        torch.cuda.synchronize()
        full = train_dp.measure(device, rank, world, global_batch=4 * world, iters=40, warm=4, graphed=False, full=True)
    out["full_gan_loop"] = {k: full[k] for k in ("value", "ms_per_iter", "global_batch", "per_gpu_batch", "launch",
                                                 "recording_forward", "collective", "loop", "finite", "skipped_steps")}
    out["unit"] = "iterations/s"
    if world == 1:
        for B in (4, 32):
            r = train_dp.measure(device, 0, 1, global_batch=B, iters=40, warm=4, graphed=True, full=False)
            out["nerf_step_b%d" % B] = {k: r[k] for k in ("value", "ms_per_iter", "global_batch", "launch", "recording_forward")}
        # kernel-level: eager nerf step at B=32 with events around the two C-ABI calls
        ev = {"fwd": [], "bwd": []}
        orig_f, orig_b = ops.mlp_forward, ops.mlp_backward

        def timed(fn, store):
            def call(*a, **k):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                res = fn(*a, **k)
                e1.record()
                store.append((e0, e1))
                return res
            return call

        ops.mlp_forward, ops.mlp_backward = timed(orig_f, ev["fwd"]), timed(orig_b, ev["bwd"])
        try:
            r = train_dp.measure(device, 0, 1, global_batch=32, iters=12, warm=3, graphed=False, full=False)
        finally:
            ops.mlp_forward, ops.mlp_backward = orig_f, orig_b
        torch.cuda.synchronize()
        samples = 32 * 256 * 64
        f_ms = float(np.mean([a.elapsed_time(b) for a, b in ev["fwd"][3:]]))
        b_ms = float(np.mean([a.elapsed_time(b) for a, b in ev["bwd"][3:]]))
        achieved = TRAIN_FLOP_PER_SAMPLE * samples / ((f_ms + b_ms) * 1e-3) / 1e12
        out["roofline"] = {"kernels": "mlp_fwd_f16x3_kernel<recording> + mlp_dgrad_f16x3_kernel + mlp_wgrad_f16x3_kernel + "
                                      "mlp_wgrad_finalize + finalize2", "bound": "mfma", "achieved": achieved, "peak": MFMA_PEAK_TFLOPS["f16x3"],
                           "unit": "TFLOP/s", "frac": achieved / MFMA_PEAK_TFLOPS["f16x3"],
                           "issued_frac": 3 * achieved / MFMA_PEAK_TFLOPS["f16x3"], "fwd_ms": f_ms, "bwd_ms": b_ms,
                           "samples_per_launch": samples, "flop_per_sample": TRAIN_FLOP_PER_SAMPLE, "traffic": _train_traffic(),
                           "traffic_unit": "bytes per B=32 step over the three MLP kernels, L2<->fabric (profiles/traffic.json: "
                                           "train_b32; the weight gradient reads 7.7 GB in 1.40-1.43 ms = 5.4-5.5 TB/s: it is the HBM-bound one of the three)",
                           "note": "B=32 nerf step, eager; ALGORITHMIC FLOP (recording forward + head backward) / HIP-event "
                                   "time of tp_mlp_fwd + tp_mlp_bwd; every product is three f16 MFMAs"}
    return out


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
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d ..."
                         % (args.gpus, args.gpus))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=device)

    sc, params, emb_t, emb_l = build_scene(device, seed=rank)       # one image per rank
    graph, opt = make_graph(device, params, emb_t, emb_l, args.precision)
    pose, intr = sc["pose"].to(device), sc["intr"].to(device)
    dr = (sc["z_near"].to(device)[:, :, None], sc["z_far"].to(device)[:, :, None])
    mask = torch.ones(1, H, W, device=device)

    # time the dominant kernel with HIP events on the stream it is launched on
    from texpose_amd import ops
    mlp_events = []
    orig_mlp = ops.mlp_forward

    def timing_into(store):
        def timed_mlp(*a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = orig_mlp(*a, **k)
            e1.record()
            store.append((e0, e1))
            return out
        return timed_mlp

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
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ret = step()
    barrier()
    dt = time.perf_counter() - t0
    ops.mlp_forward = orig_mlp
    assert torch.isfinite(ret.rgb).all() and ret.rgb.shape == (1, H * W, 3)

    t = torch.tensor([dt], device=device, dtype=torch.float64)
    if world > 1:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    dt = float(t.item())
    mlp_ms = float(np.mean([a.elapsed_time(b) for a, b in mlp_events]))
    launches_per_step = len(mlp_events) / args.steps
    samples_per_launch = H * W * N_SAMPLES / launches_per_step

    def measured_traffic(precision, samples):
        """Fabric bytes per launch from the committed PMC profile (bench.py cannot run rocprofv3 on itself)."""
        try:
            t = json.load(open(os.path.join(REPO, "profiles", "traffic.json")))[precision]
            return t["total"] * samples / (H * W * N_SAMPLES)
        except Exception:
            return None

    def roofline(precision, ms, samples):
        achieved = MLP_FLOP_PER_SAMPLE * samples / (ms * 1e-3) / 1e12
        peak = MFMA_PEAK_TFLOPS[precision]
        k = ISSUED_PER_ALGORITHMIC[precision]
        return {"kernel": "mlp_fwd_kernel" if precision == "fp32" else "mlp_fwd_f16x3_kernel", "bound": "mfma",
                "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak, "traffic": measured_traffic(precision, samples),
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
    ops.check_mlp_status(device)
    del ret
    torch.cuda.empty_cache()
    train = None
    if not args.no_train:
        try:
            train = train_leg(device, rank, world)
        except Exception as exc:                   # the rays/s line must come out whatever happens to the untimed legs
            train = {"error": repr(exc)[:400]}

    if rank == 0:
        line = {
            "metric": "rendered rays/sec (480x640x128 samples) + train iters/sec",
            "value": world * H * W * args.steps / dt,
            "unit": "rays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32" if args.precision == "fp32" else "f32 carried as 2xf16 (f16x3 products, f32 accumulate)",
            "data": "synthetic",
            "config": {"workload": "Duck-like synthetic scene 480x640, 128 samples/ray, batch=1 image per GPU, "
                                   "forward render (render_by_slices mode='val', all pixels), per-sample outputs "
                                   "materialised", "rays_per_step_per_gpu": H * W, "samples_per_ray": N_SAMPLES,
                       "mlp_precision": args.precision,
                       "parallelism": "images sharded across %d GPU(s), no collective" % world},
            "roofline": roofline(args.precision, mlp_ms, samples_per_launch),
        }
        if exact is not None:
            line["exact_fp32_kernel"] = exact
        if per_ray is not None:
            line["per_ray_outputs_only"] = per_ray
        if train is not None:
            line["train"] = train
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(sc, params, emb_t, emb_l)
            line["gpu_over_cpu"] = line["value"] / line["cpu_baseline"]["value"]
        print(json.dumps(line))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
