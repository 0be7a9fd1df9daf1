#!/usr/bin/env python3
"""Multi-object evaluation entry point (BASELINE config C5: LineMOD 8-object multi-scene, 256 samples per ray, mixed
240x320 / 480x640 images, batch 64 over 8 GPUs):

    python tools/eval_multi_object.py [--objects 8] [--images-per-object 8] [--samples 256]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        tools/eval_multi_object.py ...

The reference trains and evaluates ONE model per object (`data.object`, options/nerf_lm_adapt_gan.yaml:44; one GPU by
assertion, options.py:112), so "8 objects" = 8 independent Graphs with their own weights, latents, poses and boxes.  Objects
shard over the ranks (one object per GPU at N=8; on fewer GPUs a rank renders its objects one after the other) and there is no
data-path collective.  Each object renders `images-per-object` evaluation images through Graph.render_by_slices (mode 'val':
every pixel), alternating 240x320 (LineMOD intrinsics scaled by 1/2, data/lmsyn2real.py:329-338) and 480x640; the reference's
eval path takes one image per call (model/nerf_adapt_st_gan.py:679), so a mixed-resolution batch is a sequence of calls with the
image size in `opt.H / opt.W`.  Rank 0 prints one JSON line: per-object and aggregate rays / s, and the MLP kernel's roofline
from HIP events."""
import argparse
import itertools
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np                                                              # noqa: E402
import torch                                                                    # noqa: E402

LMO_OBJECTS = ("ape", "can", "cat", "driller", "duck", "eggbox", "glue", "holepuncher")     # ids 1,5,6,8,9,10,11,12
RESOLUTIONS = ((240, 320), (480, 640))
MLP_FLOP_PER_SAMPLE = 1_821_184
F16_PEAK_TFLOPS = 2500.0


def image_resolution(i: int):
    """Image i of an object's batch: even -> 240x320, odd -> 480x640."""
    return RESOLUTIONS[i % 2]


def build_object(obj: int, device, n_samples: int, precision: str = "f16x3", stratified: bool = True):
    """Graph + per-resolution options of object `obj`: its own network weights (seed 100 + obj), latents and box size."""
    from texpose_amd import synthetic
    from texpose_amd.graph import Graph
    from texpose_amd.options import default_options
    opts = {}
    for (H, W) in RESOLUTIONS:
        o = default_options(H=H, W=W, device=str(device))
        o.nerf.sample_intvs, o.nerf.sample_stratified, o.batch_size = n_samples, stratified, 1
        o.data.image_size = [H, W]
        opts[(H, W)] = o
    g = Graph(opts[RESOLUTIONS[0]]).to(device)
    params = synthetic.network_weights(100 + obj, bias_scale=0.0)
    g.nerf.load_state_dict({**g.nerf.state_dict(), **{k: v.to(device) for k, v in params.items()}})
    g.attach_latents(189, opts[RESOLUTIONS[0]])
    rs = np.random.RandomState(1000 + obj)
    with torch.no_grad():
        g.latent_vars_trans.weight.copy_(torch.from_numpy(rs.normal(size=(189, 16)).astype(np.float32)))
        g.latent_vars_light.weight.copy_(torch.from_numpy(rs.normal(size=(189, 48)).astype(np.float32)))
    g.nerf.precision = precision
    g.eval()
    return g, opts


def build_image(obj: int, i: int, device):
    """Pose / intrinsics / per-pixel depth bounds of image i of object obj (box half extent 0.35 .. 0.7 dm by object)."""
    from texpose_amd import synthetic
    H, W = image_resolution(i)
    sc = synthetic.eval_scene(H, W, B=1, seed=10_000 * (obj + 1) + i, half_extent=0.35 + 0.05 * obj)
    near, far = synthetic.scene_bounds(sc, H, W, device)
    return dict(H=H, W=W, pose=sc["pose"].to(device), intr=sc["intr"].to(device),
                depth_range=(near[:, :, None], far[:, :, None]), mask=torch.ones(1, H, W, device=device))


def render_image(graph, opts, im):
    with torch.no_grad():
        return graph.render_by_slices(opts[(im["H"], im["W"])], im["pose"], intr=im["intr"], depth_range=im["depth_range"],
                                      object_mask=im["mask"], sample_idx=None, mode="val")


class HipBackend:
    """What `measure` needs from the device: builders, the render call, synchronisation and a timer around ops.mlp_forward.
    tests/test_dist_gloo_cpu.py substitutes a CPU stand-in to exercise the sharding / aggregation logic under gloo."""

    def __init__(self, device, n_samples, precision):
        self.device, self.n_samples, self.precision = device, n_samples, precision
        self.events = []

    def build_object(self, o):
        return build_object(o, self.device, self.n_samples, self.precision)

    def build_image(self, o, i):
        return build_image(o, i, self.device)

    def render(self, obj, im):
        return render_image(*obj, im)

    def sync(self):
        torch.cuda.synchronize(self.device)

    def __enter__(self):
        from texpose_amd import ops
        self._ops, self._orig = ops, ops.mlp_forward

        def timed(*a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = self._orig(*a, **k)
            e1.record()
            self.events.append((e0, e1, out[1].numel() // 2))
            return out
        ops.mlp_forward = timed
        return self

    def __exit__(self, *exc):
        self._ops.mlp_forward = self._orig

    def kernel_totals(self):
        """(samples, ms) over all MLP launches of the timed region."""
        return sum(s for _, _, s in self.events), sum(a.elapsed_time(b) for a, b, _ in self.events)

    def finish(self, ret):
        assert torch.isfinite(ret.rgb).all()
        self._ops.check_mlp_status(self.device)


def objects_of_rank(n_objects: int, rank: int, world: int):
    """Objects shard contiguously over the ranks (one object per GPU at N = 8); no data-path collective."""
    from texpose_amd import dist as tdist
    return list(tdist.shard_batch(n_objects, rank, world))


def aggregate(device, world, n_objects, steps, mine, per_obj_s, samples, mlp_ms, dt):
    """The only exchange of the job: per-object times and kernel totals of every rank (SUM over disjoint slots) and the job's
    wall time (MAX).  Returns (seconds per step of the slowest rank, per-object seconds [n_objects], samples, kernel ms)."""
    stats = torch.zeros(n_objects + 2, dtype=torch.float64, device=device)
    for o in mine:
        stats[o] = per_obj_s[o] / steps
    stats[n_objects], stats[n_objects + 1] = samples, mlp_ms
    t = torch.tensor([dt], dtype=torch.float64, device=device)
    if world > 1:
        torch.distributed.all_reduce(stats, op=torch.distributed.ReduceOp.SUM)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    stats = stats.tolist()
    return float(t) / steps, stats[:n_objects], stats[n_objects], stats[n_objects + 1]


def measure(device, rank: int, world: int, n_objects: int = 8, images_per_object: int = 8, n_samples: int = 256,
            precision: str = "f16x3", warm: int = 1, steps: int = 1, keep_outputs: bool = False, backend=None):
    """One step = every object of this rank renders its whole image batch.  Returns the rank-0 line (dict) on every rank."""
    be = backend if backend is not None else HipBackend(device, n_samples, precision)
    mine = objects_of_rank(n_objects, rank, world)
    objs = {o: be.build_object(o) for o in mine}
    imgs = {o: [be.build_image(o, i) for i in range(images_per_object)] for o in mine}

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        be.sync()

    for _ in range(warm):
        for o in mine:
            be.render(objs[o], imgs[o][0])
            be.render(objs[o], imgs[o][1 % images_per_object])
    per_obj_s = {o: 0.0 for o in mine}
    outputs = {}
    ret = None
    with be:
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            for o in mine:
                be.sync()
                t1 = time.perf_counter()
                for im in imgs[o]:
                    ret = be.render(objs[o], im)
                    if keep_outputs:
                        outputs.setdefault(o, []).append(ret)
                be.sync()
                per_obj_s[o] += time.perf_counter() - t1
        barrier()
        dt = time.perf_counter() - t0
    if mine:
        be.finish(ret)
    # every object has the same image sizes: rays per object from the resolution sequence (a rank without objects knows it too)
    rays_obj = sum(h * w for h, w in (image_resolution(i) for i in range(images_per_object)))
    samples, mlp_ms = be.kernel_totals()
    dt, stats, tot_samples, tot_ms = aggregate(device, world, n_objects, steps, mine, per_obj_s, samples, mlp_ms, dt)
    achieved = MLP_FLOP_PER_SAMPLE * tot_samples / (tot_ms * 1e-3) / 1e12 if tot_ms else 0.0
    peak = F16_PEAK_TFLOPS if precision == "f16x3" else 157.3
    total_rays = n_objects * rays_obj
    from texpose_amd import dist as tdist
    line = dict(
        metric="rendered rays/sec, C5 (8 objects x 8 images, 240x320 / 480x640 alternating, 256 samples per ray)",
        value=total_rays / dt, unit="rays/s", n_gpus=world, ranks_seen=tdist.ranks_seen(device=device), steps=steps, warmup=warm, ms_per_step=dt * 1e3,
        higher_is_better=True, scaling="strong", vs_baseline=None,
        dtype="f32 carried as 2xf16 (f16x3 products, f32 accumulate)" if precision == "f16x3" else "f32", data="synthetic",
        config=dict(workload="C5: %d independent object models (one Graph each: own weights / latents / poses / box), %d "
                             "images per object alternating 240x320 and 480x640, %d samples per ray, forward render of every "
                             "pixel (render_by_slices mode='val'); objects sharded over the GPUs, no collective"
                             % (n_objects, images_per_object, n_samples),
                    objects=list(LMO_OBJECTS[:n_objects]) if n_objects <= len(LMO_OBJECTS) else n_objects,
                    global_batch=n_objects * images_per_object, samples_per_ray=n_samples, rays_per_object=rays_obj,
                    mlp_precision=precision, parallelism="%d object(s) per GPU on %d GPU(s)" % (-(-n_objects // world), world)),
        per_object=[dict(object=LMO_OBJECTS[o] if o < len(LMO_OBJECTS) else o, rays_per_s=rays_obj / stats[o] if stats[o] else None,
                         ms=stats[o] * 1e3) for o in range(n_objects)],
        roofline=dict(kernel="mlp_fwd_f16x3_kernel" if precision == "f16x3" else "mlp_fwd_exact_asm_kernel", bound="mfma",
                      achieved=achieved, peak=peak, unit="TFLOP/s", frac=achieved / peak, traffic=None,
                      kernel_ms_total_all_ranks=tot_ms / steps, samples_all_ranks=tot_samples / steps,
                      flop_per_sample=MLP_FLOP_PER_SAMPLE,
                      note="ALGORITHMIC FLOP of all MLP launches of the timed region / their summed HIP-event time (all ranks)"))
    return (line, outputs) if keep_outputs else line


class StubBackend:
    """CPU stand-in of HipBackend for the launcher / sharding self-test (`bench.py --config c5 --spawn-check`, gloo): a render
    "takes" 1 ms x (object + 1) and "runs" one MLP launch of H*W*n_samples samples in 0.5 ms."""

    def __init__(self, n_samples):
        self.n_samples, self.samples, self.ms, self.calls = n_samples, 0, 0.0, []

    def build_object(self, o):
        return o

    def build_image(self, o, i):
        h, w = image_resolution(i)
        return dict(H=h, W=w, obj=o, i=i)

    def render(self, obj, im):
        time.sleep(1e-3 * (obj + 1))
        if getattr(self, "_timing", False):
            self.samples += im["H"] * im["W"] * self.n_samples
            self.ms += 0.5
            self.calls.append((obj, im["i"]))
        return im

    def sync(self):
        pass

    def __enter__(self):
        self._timing = True
        return self

    def __exit__(self, *exc):
        self._timing = False

    def kernel_totals(self):
        return self.samples, self.ms

    def finish(self, ret):
        pass


def spawn_check(rank: int, world: int, n_objects: int = 8, images_per_object: int = 2, n_samples: int = 4):
    """`measure` with the stub renderer on CPU under gloo: the objects -> ranks sharding and the SUM / MAX aggregation of the real
    entry point, without a GPU.  Returns the line (identical on every rank)."""
    import torch.distributed as dist
    if world > 1 and not dist.is_initialized():
        dist.init_process_group("gloo")
    be = StubBackend(n_samples)
    line = measure(torch.device("cpu"), rank, world, n_objects, images_per_object, n_samples, "f16x3", warm=0, steps=1, backend=be)
    line["stub_calls_this_rank"] = be.calls
    return line


def run(argv=None, emit=None):
    """``emit``: how rank 0 hands over its result line (bench.py passes its stdout guard; default: print)."""
    ap = argparse.ArgumentParser()
    ap.add_argument("--objects", type=int, default=8)
    ap.add_argument("--images-per-object", type=int, default=8)
    ap.add_argument("--samples", type=int, default=256)
    ap.add_argument("--precision", choices=["f16x3", "fp32"], default="f16x3")
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=1)
    a = ap.parse_args(argv)
    from texpose_amd import dist as tdist
    rank, world, local = tdist.init_distributed()
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    line = measure(device, rank, world, a.objects, a.images_per_object, a.samples, a.precision, a.warmup, a.steps)
    if rank == 0:
        (emit or (lambda l: print(json.dumps(l))))(line)
    if world > 1:
        torch.distributed.destroy_process_group()
    return line


if __name__ == "__main__":
    run()
