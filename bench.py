#!/usr/bin/env python3
"""bench.py -- fwd+bwd views/sec of the textured Gaussian rasterizer operator on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N>1 launched under
torch.distributed.run, one rank per GPU.  Prints ONE JSON line on rank 0.

Workload (BASELINE.json configs[2], "C3"): 300k Gaussians + 6x1024^2x3 cubemap, 800x800, fwd+bwd, synthetic
seeded scene.  A "step" = `--views-per-step` (default 8) views per rank, each a full
operator forward + backward through the public GaussianRasterizer (the module render/uv_tex_render.py:40 builds),
gradients accumulated into one flat fp32 bucket; with N>1 the bucket is all-reduced (RCCL, SUM) once per step --
BASELINE configs[3]: 64-view batch over 8 GPUs = 8 views per rank per all-reduce.  value = views of all ranks / time.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "texture-gs_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

WORKLOADS = {
    # name: (N, R, W, H, mode)
    "c1": (1000, 64, 256, 256, "fwd+bwd"),
    "c2": (100_000, 512, 800, 800, "fwd"),
    "c3": (300_000, 1024, 800, 800, "fwd+bwd"),
    "c5": (1_000_000, 2048, 1600, 1200, "fwd+bwd"),
}
HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--views-per-step", type=int, default=8)
    ap.add_argument("--num-views", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-table", action="store_true",
                    help="skip the extra serial steps behind the per-kernel table (profiling runs: every launch is then a pipelined one)")
    ap.add_argument("--wire", default=os.environ.get("TEXGS_WIRE", "f32"), choices=["f32", "bf16-sh"],
                    help="N > 1: dtype of the gradient all-reduce on the wire.  bf16-sh = the view-dependent SH gradients (54 of the 150 MB "
                         "of a C3 bucket) are summed as bf16 -- GradBucket.all_reduce_async(wire_dtype=...) --, everything else stays f32")
    ap.add_argument("--tex-res", type=int, default=0,
                    help="experiment: override the workload's cubemap resolution (e.g. C3 geometry with R = 2048: a 302 MB texture, past the "
                         "256 MB Infinity Cache); the metric string then says so and is never the headline")
    ap.add_argument("--no-extra-legs", action="store_true",
                    help="skip the reference-call-pattern / reference-iteration / retexture legs after the timed region (A/B runs)")
    ap.add_argument("--streams", type=int, default=int(os.environ.get("TEXGS_BENCH_STREAMS", "3")),
                    help="HIP streams the views of a step are pipelined over (texgs.multiview.ViewPipeline); 1 = serial")
    ap.add_argument("--prefetch", type=int, default=int(os.environ.get("TEXGS_BENCH_PREFETCH", "1")), choices=[0, 1],
                    help="1 = ViewPipeline.run begins every view's forward (K1 + the instance-count readback) one view per stream ahead "
                         "(GaussianRasterizer.prefetch); 0 = every forward begins when it is called")
    ap.add_argument("--order", default=os.environ.get("TEXGS_BENCH_ORDER", "accumulate"), choices=["backward", "accumulate", "none"])
    ap.add_argument("--leg", default="operator", choices=["operator", "iteration"],
                    help="iteration = one training iteration of the reference's texture stage on this stack (UV map, two renders, the "
                         "loss front-end, one backward; models/texture_gaussian3d.py:315-418) -- scripts/bench_iteration.py; its own "
                         "metric, never the headline")
    ap.add_argument("--surface", default="textured", choices=["textured", "diff_gauss"],
                    help="diff_gauss = the untextured operator the reference's stages 1-2 call (render/render.py:75-84): same scene, "
                         "colours from SH with DC, no texture (SURVEY 8f-1); reported as its own metric, not the headline")
    return ap.parse_args()


def algorithmic_bytes(N, K, D, D_eff, P, T, R, n_vis, n_touched, texels_touched):
    """Per-launch algorithmic HBM bytes of every kernel (DESIGN.md section 5 states each term).  D_eff = sum over
    tiles of the last contributor's position (the replay never needs the rest of the list); texels_touched =
    distinct texels with a non-zero gradient in this view."""
    tex = 12 * texels_touched
    return {
        "preprocess_fwd": N * (92 + 12 * K) + n_vis * 96 + N * 20,
        # K2: partition of the N depth keys into bins (count: 4N read; scatter: 4N read + 8N of (key, index) pairs written) + one
        # sort per group of bins (8N read, 8N of rank-ordered key / index written, tiles_touched gathered 4N, offsets written 4N)
        "scan": N * 4 + N * 12 + N * 24,
        "duplicate": N * 20 + D * 8 + T * 8,
        # K4: 2 stable passes over the D 8-byte (tile | rank) elements; the last writes 12 B / element and gathers 8 B
        "sort": 2 * D * (8 + 8) + D * 8 + D * (12 + 8),
        "ranges": D * 8 + T * 8 + T * 12,
        "render_fwd": D_eff * 100 + tex + P * 40 + T * 8,
        "render_bwd": D_eff * 100 + P * 40 + tex + 2 * tex + n_touched * 256 + T * 8,
        "preprocess_bwd": N * (96 + 12 * K) + n_vis * 96 + N * (68 + 12 * K),
        "texgrad_reduce": 0,      # an artefact of the gradient scatter, no algorithmic traffic of its own
    }


# ---- cpu_baseline leg: the ONLY place bench.py touches oracle/ (a CPU port of the operator, used as the reported
# baseline; the reference has no CPU rasterizer).  Never on the measured GPU path.
def cpu_baseline(scene, cam, W, H, with_bwd, budget_s=20.0):
    from oracle import texgs_ref as CR
    from oracle import texgs_torch as O
    st = O.Settings(cam.image_height, cam.image_width, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5),
                    torch.zeros(3), 1.0, cam.world_view_transform, cam.full_proj_transform, 3, cam.camera_center,
                    False, False)
    run = CR.RefRun(scene, st)
    g = np.random.RandomState(1234)
    dout = (g.randn(8, H, W) / (H * W)).astype(np.float32)
    dout[3] = 0.0                                   # the bench's upstream grads: image, norm, alpha
    views, t0 = 0, time.perf_counter()
    while True:
        run.forward()
        if with_bwd:
            run.backward(dout)
        views += 1
        el = time.perf_counter() - t0
        if el > budget_s or views >= 8:
            break
    return {"value": round(views / el, 4), "unit": "views/s", "cores": run.threads, "kind": "port",
            "sample": f"{views} whole view(s) fwd{'+bwd' if with_bwd else ''} of the same scene/camera by oracle/texgs_ref.c "
                      f"(gcc -O2 -fopenmp, fp32, {run.threads} OpenMP threads) in {el:.1f} s; D={run.D}",
            "seconds_measured": round(el, 2)}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    # one rank per GPU.  TEXGS_DIST_BACKEND=gloo is the single-GPU rehearsal of the N>1 path (tests): the ranks then share the
    # visible device(s) round-robin and the bucket is reduced through host memory (RCCL refuses two ranks on one device).
    backend = os.environ.get("TEXGS_DIST_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    force_dist = os.environ.get("TEXGS_FORCE_DIST") == "1" and "RANK" in os.environ   # 1-rank RCCL smoke of the N>1 code path
    if world > 1 or force_dist:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)

    from texgs import synth, _lib
    from texgs.rasterizer import GaussianRasterizationSettings, GaussianRasterizer, forward_raw, backward_raw
    from texgs.multiview import GradBucket, ViewPipeline, shard_views

    N, R, W, H, mode = WORKLOADS[args.workload]
    if args.tex_res:
        R = args.tex_res
    if args.leg == "iteration":
        sys.path.insert(0, os.path.join(ROOT, "scripts"))
        import bench_iteration
        if rank == 0:
            res = bench_iteration.run(N, R, W, H, iters=max(args.steps, 4), warm=max(args.warmup, 2), dev_index=dev_index,
                                      precision=os.environ.get("TEXGS_UV_PRECISION", "mixed"))
            print(json.dumps({"metric": f"reference training iteration (texture stage), ms per iteration ({args.workload})",
                              "value": res["uv_once_ms_per_iteration"], "unit": "ms/iteration", "higher_is_better": False, "n_gpus": 1,
                              "dtype": "f32", "data": "synthetic", **res}), flush=True)
        return
    with_bwd = mode == "fwd+bwd"
    K = 15
    scene = synth.make_scene(N, R, seed=0)
    cams = synth.fibonacci_cameras(args.num_views, W, H)
    my_views = shard_views(args.num_views, rank, world)          # (replaced below by cost-balanced sharding when world > 1)
    bg = torch.zeros(3, device=dev)

    untextured = args.surface == "diff_gauss"
    names = ["means3D", "shs", "opacities", "scales", "rotations"] + ([] if untextured else ["uvs", "texture"])
    leaves = {n: getattr(scene, n).to(dev).requires_grad_(with_bwd) for n in names}
    if untextured:          # SH with the DC band first, as models/gaussian3d.py keeps them
        leaves["shs"] = torch.cat([torch.zeros(N, 1, 3), scene.shs], 1).to(dev).requires_grad_(with_bwd)
    juv = scene.gradient_uvs.to(dev)
    means2D = torch.zeros(N, 3, device=dev, requires_grad=with_bwd)
    # the texture last, the SH coefficients before it: the flat gradient bucket is then three contiguous runs,
    # [small per-Gaussian | shs | texture] -- all-reduced as [small + shs], [texture] (f32 wire) or [small], [shs as bf16], [texture]
    small = [leaves[n] for n in names if n not in ("texture", "shs")] + [means2D]
    params = small + [leaves["shs"]] + ([leaves["texture"]] if "texture" in leaves else [])
    bucket = GradBucket(params) if with_bwd else None
    seg_tex = bucket.segment_of([leaves["texture"]]) if (with_bwd and "texture" in leaves) else None
    seg_gauss = bucket.segment_of([p_ for p_ in params if p_ is not leaves.get("texture")]) if with_bwd else None
    seg_small = bucket.segment_of(small) if with_bwd else None
    seg_shs = bucket.segment_of([leaves["shs"]]) if with_bwd else None

    def settings(cam):
        return GaussianRasterizationSettings(
            image_height=cam.image_height, image_width=cam.image_width,
            tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5), bg=bg, scale_modifier=1.0,
            viewmatrix=cam.world_view_transform.to(dev), projmatrix=cam.full_proj_transform.to(dev), sh_degree=3,
            campos=cam.camera_center.to(dev), prefiltered=False, debug=False)
    # fused gradient accumulation: the kernels add into the bucket slices (texgs.multiview), no AccumulateGrad pass
    if untextured:
        import diff_gauss as dg
        rasters = {v: dg.GaussianRasterizer(settings(cams[v]), grad_sink=bucket) for v in my_views}
    else:
        rasters = {v: GaussianRasterizer(settings(cams[v]), grad_sink=bucket) for v in my_views}

    # fixed upstream gradients of the synthetic loss' shape: image, alpha, norm
    g = torch.Generator().manual_seed(1234)
    P = W * H
    g_img = ((torch.rand(3, H, W, generator=g) > 0.5).float() * 2 - 1).to(dev) / (3 * P)
    g_alpha = ((torch.rand(1, H, W, generator=g) > 0.5).float() * 2 - 1).to(dev) / P
    nh = torch.randn(3, H, W, generator=g)
    g_norm = (-0.1 * nh / nh.norm(dim=0, keepdim=True)).to(dev) / P

    def view_fwd(v):
        if untextured:
            return rasters[v](means3D=leaves["means3D"], means2D=means2D, shs=leaves["shs"], opacities=leaves["opacities"],
                              scales=leaves["scales"], rotations=leaves["rotations"])
        return rasters[v](means3D=leaves["means3D"], means2D=means2D, shs=leaves["shs"],
                          opacities=leaves["opacities"], scales=leaves["scales"], rotations=leaves["rotations"],
                          uvs=leaves["uvs"], gradient_uvs=juv, texture=leaves["texture"], extra_attrs=None)

    def view_prefetch(v):        # K1 + the instance-count readback of view v, begun ahead of the stream's pending backward
        return rasters[v].prefetch(means3D=leaves["means3D"], means2D=means2D, shs=leaves["shs"],
                                   opacities=leaves["opacities"], scales=leaves["scales"], rotations=leaves["rotations"],
                                   uvs=leaves["uvs"], gradient_uvs=juv, texture=leaves["texture"], extra_attrs=None)

    def view_bwd(out):
        torch.autograd.backward([out[0], out[3], out[2]], [g_img, g_alpha, g_norm])

    if world > 1:
        # views dealt by estimated cost instead of round-robin: one forward-only pass over every view gives its instance count D
        # (what K6 / K7 time is proportional to); deterministic, so every rank computes the same partition
        from texgs.multiview import lpt_shard_views
        costs = []
        with torch.no_grad():
            for v in range(args.num_views):
                st_v = settings(cams[v])
                if untextured:
                    _, s_v = forward_raw(st_v, leaves["means3D"].detach(), None, leaves["opacities"].detach(), leaves["scales"].detach(),
                                         leaves["rotations"].detach(), None, None, None, for_backward=False)
                else:
                    _, s_v = forward_raw(st_v, leaves["means3D"].detach(), None, leaves["opacities"].detach(), leaves["scales"].detach(),
                                         leaves["rotations"].detach(), leaves["uvs"].detach(), juv, leaves["texture"].detach(),
                                         for_backward=False)
                costs.append(s_v.D)
        my_views = lpt_shard_views(costs, rank, world)
        if untextured:
            import diff_gauss as dg
            rasters = {v: dg.GaussianRasterizer(settings(cams[v]), grad_sink=bucket) for v in my_views}
        else:
            rasters = {v: GaussianRasterizer(settings(cams[v]), grad_sink=bucket) for v in my_views}
    pipe = ViewPipeline(dev, depth=args.streams)
    pipe_serial = ViewPipeline(dev, depth=1)
    cursor = [0]

    step_recs = []          # N > 1: (start, render done, all-reduce waited for) events of the timed steps, this rank

    def step(p=pipe, rec=None):
        if rec is not None:
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            ev[0].record()
        if with_bwd:
            bucket.zero()
        batch = [my_views[(cursor[0] + i) % len(my_views)] for i in range(args.views_per_step)]
        cursor[0] += args.views_per_step
        two = with_bwd and dist is not None and args.order == "accumulate" and seg_tex is not None
        # the texture half of the bucket is all-reduced on a side stream as soon as the last view's texture-gradient reduce has been
        # issued (it overlaps that view's K8 and the host's end-of-step work); the per-Gaussian half after the last K8
        p.run(batch, view_fwd, view_bwd if with_bwd else None, sink=bucket, order=args.order,
              texture_ready=(lambda evs: bucket.all_reduce_async(dist, seg_tex, after=evs, timing=True)) if two else None,
              prefetch_fn=view_prefetch if (args.prefetch and p.streams and not untextured and with_bwd) else None,   # (forward-only: nothing to get ahead of)
              prefetch_ahead=int(os.environ.get("TEXGS_BENCH_PREFETCH_AHEAD", "1")))
        if rec is not None:
            ev[1].record()
        if with_bwd and dist is not None:
            if two and args.wire == "bf16-sh":
                bucket.all_reduce_async(dist, seg_small, timing=True)
                bucket.all_reduce_async(dist, seg_shs, timing=True, wire_dtype=torch.bfloat16)
                bucket.wait()
            elif two:
                bucket.all_reduce_async(dist, seg_gauss, timing=True)
                bucket.wait()
            else:
                bucket.all_reduce(dist)
        if rec is not None:
            ev[2].record()
            rec.append(ev)

    def fence():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    if os.environ.get("TEXGS_BENCH_GC_FREEZE", "1") == "1":
        # everything allocated so far (torch, the scene, the modules) is long-lived: keep the cyclic collector from walking it when its
        # oldest generation comes due in the middle of the step loop (measured: ONE 40-60 ms stall, 10-40 ms after the loop starts --
        # 25 % of a C2 timed region, and able to land in C3's)
        import gc
        gc.collect()
        gc.freeze()
    for _ in range(args.warmup):
        step()
    fence()
    # HIP events inside the timed region bracket ONLY the dominant kernel (an event pair costs ~5 us of stream time; all
    # nine kernel groups bracketed cost ~90 us / view).  The full per-kernel table comes from two extra steps afterwards.
    DOMINANT = "render_bwd" if with_bwd else "render_fwd"
    _lib.profile_enable(True, only=[DOMINANT])
    _lib.profile_read()
    step_ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    step_ev[0].record()
    for k in range(args.steps):
        step(rec=step_recs if dist is not None else None)
        step_ev[k + 1].record()
    fence()
    t1 = time.perf_counter()
    kern_timed = _lib.profile_read()
    comm = bucket.comm_timings() if (with_bwd and dist is not None) else []
    # the same step loop again for >= 2 s (the mandated K steps are a fraction of a second -- too short for an outside
    # sampler to see the GPU busy); reported beside `value`, never instead of it
    _lib.profile_enable(False)
    el_max = t1 - t0
    if dist is not None:        # (every rank must run the SAME number of steps: they contain collectives)
        te = torch.tensor([el_max], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        el_max = float(te.item())
    long_steps = min(4096, max(args.steps, int(math.ceil(2.2 / max(el_max / args.steps, 1e-6)))))
    fence()
    tl0 = time.perf_counter()
    for _ in range(long_steps):
        step()
    fence()
    long_elapsed = time.perf_counter() - tl0
    if dist is not None:
        tl = torch.tensor([long_elapsed], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tl, op=dist.ReduceOp.MAX)
        long_elapsed = float(tl.item())
    value_long = long_steps * args.views_per_step * world / long_elapsed
    if with_bwd and dist is not None:
        bucket.comm_timings()           # (drop the long loop's entries: `comm` above is the timed region's)
    # the per-kernel table: two extra steps with every kernel group bracketed, views one after the other on one stream, so
    # that each duration is the kernel's own (in the pipelined region a kernel shares the GPU with other views' kernels)
    kern = dict(kern_timed)
    kern_solo_dom = (0.0, 0)
    if not args.no_kernel_table:
        _lib.profile_enable(False)
        for _ in range(2):      # the serial path runs on another stream: its per-stream scratch (texture-bin capacity) adapts first
            step(pipe_serial)
        fence()
        _lib.profile_enable(True, only=[DOMINANT])
        step(pipe_serial)
        fence()
        kern_solo_dom = _lib.profile_read()[DOMINANT]      # the dominant kernel alone on the GPU, only it bracketed
        _lib.profile_enable(True)
        for _ in range(2):
            step(pipe_serial)
        fence()
        kern = _lib.profile_read()
    _lib.profile_enable(False)
    kern[DOMINANT] = kern_timed[DOMINANT]
    # N > 1: what every rank did in the timed region, gathered to rank 0 (the first real multi-GPU run must be
    # diagnosable from its one JSON line): its views and their instance counts, GPU time of the rendering part of a step, the part
    # of the all-reduce the compute stream had to wait for, the collectives' own durations on the comm stream (call order)
    rank_table = None
    if dist is not None:
        ncoll = (len(comm) // max(len(step_recs), 1)) if step_recs else 0
        mine = {"rank": rank, "device": torch.cuda.get_device_name(dev), "views": len(my_views),
                "D_sum": int(sum(costs[v] for v in my_views)) if world > 1 else None,
                "render_ms_per_step": round(sum(e[0].elapsed_time(e[1]) for e in step_recs) / max(len(step_recs), 1), 4),
                "allreduce_wait_ms_per_step": round(sum(e[1].elapsed_time(e[2]) for e in step_recs) / max(len(step_recs), 1), 4),
                "allreduce_ms_by_collective": [round(sum(m for _, m in comm[k::ncoll]) / max(len(comm[k::ncoll]), 1), 4) for k in range(ncoll)],
                "allreduce_wire_bytes_by_collective": [int(comm[k][0]) for k in range(ncoll)]}
        gathered = [None] * dist.get_world_size()
        dist.all_gather_object(gathered, mine)
        rank_table = {"backend": dist.get_backend(), "ranks_seen": dist.get_world_size(), "wire": args.wire, "ranks": gathered}
    step_raw = [step_ev[k].elapsed_time(step_ev[k + 1]) for k in range(args.steps)]
    step_ms = sorted(step_raw)
    if os.environ.get("TEXGS_BENCH_STEP_TRACE") and rank == 0:
        print("step_ms", [round(x, 2) for x in step_raw], file=sys.stderr)
    pct = lambda q: step_ms[min(len(step_ms) - 1, int(q * len(step_ms)))]
    elapsed = t1 - t0
    if dist is not None:
        tt = torch.tensor([elapsed], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    total_views = args.steps * args.views_per_step * world
    value = total_views / elapsed

    # ---- units of one representative launch (view my_views[0]), outside the timed region
    st0 = settings(cams[my_views[0]])
    with torch.no_grad():
        if untextured:
            outs, s = forward_raw(st0, leaves["means3D"].detach(), leaves["shs"].detach()[:, 1:, :].contiguous(), leaves["opacities"].detach(),
                                  leaves["scales"].detach(), leaves["rotations"].detach(), None, None, None,
                                  color_offset=(0.28209479177387814 * leaves["shs"].detach()[:, 0, :]).contiguous())
        else:
            outs, s = forward_raw(st0, leaves["means3D"].detach(), leaves["shs"].detach(), leaves["opacities"].detach(),
                                  leaves["scales"].detach(), leaves["rotations"].detach(), leaves["uvs"].detach(), juv,
                                  leaves["texture"].detach())
        T = s.tensors["ranges"].shape[0]
        nc = s.tensors["n_contrib"].to(torch.int64)
        tx, ty = (W + 15) // 16, (H + 15) // 16
        pad = torch.zeros(ty * 16, tx * 16, dtype=torch.int64, device=dev)
        pad[:H, :W] = nc
        D_eff = int(pad.reshape(ty, 16, tx, 16).amax(dim=(1, 3)).sum())
        n_vis = int((outs[4] > 0).sum())
        n_touched, texels = n_vis, 0
        if with_bwd:
            res = backward_raw(s, g_img, None, g_norm, g_alpha)
            n_touched = int((res[3].reshape(N, -1).abs().sum(1) > 0).sum())     # Gaussians that received a gradient
            texels = 0 if untextured else int((res[7].abs().sum(-1) > 0).sum())
        else:
            texels = 0
    ab = algorithmic_bytes(N, K, s.D, D_eff, P, T, R, n_vis, n_touched, texels)
    kinfo = {}
    for name, (ms, cnt) in kern.items():
        if cnt:
            kinfo[name] = {"avg_us": 1e3 * ms / cnt, "launches": cnt, "alg_MB": ab.get(name, 0) / 1e6,
                           "GBps": ab.get(name, 0) / (ms / cnt * 1e-3) / 1e9}
    dom = max(kinfo, key=lambda k: kinfo[k]["avg_us"]) if kinfo else None        # every group launches once per view
    roofline = None
    traffic = None
    traffic_all = None
    traffic_source = None
    # PMC pass of the same command (scripts/prof.sh + scripts/make_traffic.py), used only if it was measured on THESE kernel sources
    tfiles = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_traffic.json")) if os.path.isdir(os.path.join(ROOT, "profiles")) else []
    if dom and args.workload == "c3" and not args.tex_res and not untextured and tfiles:
        try:
            sys.path.insert(0, os.path.join(ROOT, "scripts"))
            from make_traffic import source_hash
            tj = json.load(open(os.path.join(ROOT, "profiles", tfiles[-1])))
            if tj.get("_kernel_source_hash") == source_hash():
                traffic = tj.get(dom, {}).get("traffic_bytes")
                traffic_all = {k: v for k, v in tj.items() if not k.startswith("_")}
                traffic_source = (f"profiles/{tfiles[-1]} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this workload with --streams 1 on "
                                  f"these kernel sources, hash {tj['_kernel_source_hash']}; {tj.get('_calibration', '')})")
            else:
                traffic_source = f"profiles/{tfiles[-1]} was measured on other kernel sources (hash mismatch): not used"
        except Exception as e:          # noqa: BLE001
            traffic_source = f"no usable traffic file ({type(e).__name__})"
    if dom:
        ach = kinfo[dom]["GBps"]
        roofline = {"bound": "hbm", "kernel": dom, "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_source,
                    "alg_bytes_per_launch": ab[dom], "avg_launch_us": round(kinfo[dom]["avg_us"], 2)}
        # the same figure on SURVEY.md section 8(d)'s terms for the two blend kernels (K7: 52 P + 116 D + 228 N + 216 R^2, K6:
        # 116 D + 72 R^2 + 40 P + 8 T -- whole lists and the whole texture, where this file's own definition counts the list up to
        # the last contributor and the texels actually touched: the smaller, stricter number stays `achieved`)
        sv = {"render_bwd": 52 * P + 116 * s.D + 228 * N + 216 * R * R, "render_fwd": 116 * s.D + 72 * R * R + 40 * P + 8 * T}.get(dom)
        if sv and not untextured:
            roofline["alg_bytes_per_launch_survey_8d"] = sv
            roofline["frac_survey_8d"] = round(sv / (kinfo[dom]["avg_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 5)
        if traffic_all:
            # every kernel group's measured past-L2 traffic per view beside its algorithmic bytes (VERDICT r5 #3 iii): the ratio is
            # what the gather / scatter access patterns cost on top of the compulsory bytes
            tb = {k: {"traffic_MB": round(v["traffic_bytes"] / 1e6, 1), "alg_MB": round(ab.get(k, 0) / 1e6, 1),
                      "traffic_over_alg": (round(v["traffic_bytes"] / ab[k], 2) if ab.get(k) else None)} for k, v in traffic_all.items()}
            ttot = sum(v["traffic_bytes"] for v in traffic_all.values())
            roofline["traffic_by_kernel_per_view"] = tb
            roofline["traffic_total_per_view_MB"] = round(ttot / 1e6, 1)
            roofline["traffic_total_over_alg"] = round(ttot / max(sum(ab.values()), 1), 2)
            roofline["traffic_rate_at_value_GBps"] = round(ttot * value / world / 1e9, 1)
        if dom == DOMINANT and kern_solo_dom[1]:
            solo_us = 1e3 * kern_solo_dom[0] / kern_solo_dom[1]
            roofline["solo_launch_us"] = round(solo_us, 2)
            roofline["solo_frac"] = round(ab[dom] / (solo_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 5)
            if args.streams > 1:
                roofline["note"] = (f"achieved / frac use the launch duration inside the timed region, where {args.streams} views are in "
                                    "flight on separate HIP streams and the kernel shares the GPU with other views' kernels; solo_* is "
                                    "the same kernel with the views run one after the other (extra steps after the timed region)")
    view_bytes = sum(ab[k] for k in ab if (with_bwd or k not in ("render_bwd", "preprocess_bwd")))

    # same-run measured HBM ceiling: device-to-device copy of 1 GiB, read + write bytes
    hbm_measured = None
    if rank == 0:
        try:
            src = torch.empty(1 << 28, dtype=torch.float32, device=dev)
            dst = torch.empty_like(src)
            for _ in range(2):
                dst.copy_(src)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                dst.copy_(src)
            e1.record()
            torch.cuda.synchronize(dev)
            hbm_measured = round(5 * 2 * src.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1)
            del src, dst
        except Exception:
            hbm_measured = None

    # ---- the same workload through the reference-compatible call pattern: activation OUTPUTS as operator
    # inputs, a fresh non-leaf means2D per view, plain autograd (no grad_sink) -- what render/uv_tex_render.py does.
    compat = None
    ref_iter = None
    if rank == 0 and with_bwd and not args.no_kernel_table and not args.no_extra_legs and not untextured:
        raw = {n: leaves[n].detach().clone().requires_grad_(True) for n in names}
        raw["scales"] = leaves["scales"].detach().log().requires_grad_(True)
        op = leaves["opacities"].detach().clamp(1e-6, 1 - 1e-6)
        raw["opacities"] = torch.log(op / (1 - op)).requires_grad_(True)

        compat_settings = {v: settings(cams[v]) for v in my_views[:max(4, min(16, len(my_views)))]}     # the reference's cameras
        # live on the GPU (utils/cameras.py:62-65); the rasterizer module itself is built per call (render/uv_tex_render.py:40)

        def compat_view(v):
            m2 = torch.zeros_like(raw["means3D"], requires_grad=True) + 0
            m2.retain_grad()
            out = GaussianRasterizer(compat_settings[v])(
                means3D=raw["means3D"], means2D=m2, shs=raw["shs"], opacities=torch.sigmoid(raw["opacities"]),
                scales=torch.exp(raw["scales"]), rotations=torch.nn.functional.normalize(raw["rotations"]),
                uvs=raw["uvs"], gradient_uvs=juv, texture=raw["texture"], extra_attrs=None)
            torch.autograd.backward([out[0], out[3], out[2]], [g_img, g_alpha, g_norm])
            for p_ in raw.values():          # optimizer.zero_grad(set_to_none=True) after every iteration (models/texture_gaussian3d.py:442-444)
                p_.grad = None
        nv = min(16, len(my_views))
        for v in my_views[:4]:
            compat_view(v)
        torch.cuda.synchronize(dev)
        c0 = time.perf_counter()
        for v in my_views[:nv]:
            compat_view(v)
        torch.cuda.synchronize(dev)
        compat = {"views_per_s": round(nv / (time.perf_counter() - c0), 2), "views": nv,
                  "note": "one view per call through autograd: sigmoid/exp/normalize activations + their backward, fresh "
                          "means2D, gradients delivered by autograd (no fused sink) and dropped with set_to_none after every view as "
                          "models/texture_gaussian3d.py:442-444 does (until round 5 they were left to accumulate: an extra "
                          "read-modify-write of 150 MB per view the reference does not pay); measured after the timed region"}

        # ---- the reference's ITERATION after iteration 10 000 (models/texture_gaussian3d.py:318, 375-389, 410): render at the active
        # degree, render AGAIN at sh_degree 0 (lambda_no_sh: same camera, same Gaussians, activations recomputed by the getters),
        # one backward of both losses.  With the shared-geometry path the second render is K1 + K6 on the first one's lists.
        from texgs import rasterizer as RZ
        st_pairs = {v: (compat_settings[v], compat_settings[v]._replace(sh_degree=0)) for v in compat_settings}

        def ref_iteration(v):
            outs = []
            for st_ in st_pairs[v]:
                m2 = torch.zeros_like(raw["means3D"], requires_grad=True) + 0
                m2.retain_grad()
                outs.append(GaussianRasterizer(st_)(
                    means3D=raw["means3D"], means2D=m2, shs=raw["shs"], opacities=torch.sigmoid(raw["opacities"]),
                    scales=torch.exp(raw["scales"]), rotations=torch.nn.functional.normalize(raw["rotations"]),
                    uvs=raw["uvs"], gradient_uvs=juv, texture=raw["texture"], extra_attrs=None))
            torch.autograd.backward([outs[0][0], outs[0][3], outs[0][2], outs[1][0]], [g_img, g_alpha, g_norm, 2.0 * g_img])
            for p_ in raw.values():
                p_.grad = None
        ref_iter = {}
        saved_gc = RZ.GEOM_CACHE
        for label, on in (("separate_geometry", False), ("shared_geometry", True)):
            RZ.GEOM_CACHE = on
            RZ.release_scratch(dev)
            vs = list(compat_settings)[:nv]
            for v in vs[:3]:
                ref_iteration(v)
            torch.cuda.synchronize(dev)
            c0 = time.perf_counter()
            for v in vs:
                ref_iteration(v)
            torch.cuda.synchronize(dev)
            ref_iter[label + "_ms_per_iteration"] = round(1e3 * (time.perf_counter() - c0) / len(vs), 4)
        RZ.GEOM_CACHE = saved_gc
        ref_iter["iterations"] = nv
        ref_iter["note"] = ("2 renders (sh_degree 3, then 0) + 1 backward per iteration through plain autograd, one view per iteration; "
                            "shared = the second render re-uses the first one's tile / survivor lists (K1 geometry fingerprint)")
        del raw

    # ---- forward-only callers that build a graph (retexture.py:27, visual_step: parameters require grad, backward never runs)
    retex = None
    if rank == 0 and not with_bwd and not args.no_kernel_table and not args.no_extra_legs and not untextured:
        from texgs import rasterizer as RZ
        saved_gc = RZ.GEOM_CACHE
        RZ.GEOM_CACHE = False               # (every frame here is a different view anyway)
        vs = my_views[:min(32, len(my_views))]
        res = {}
        for label, rg in (("no_grad_inputs", False), ("inputs_require_grad", True)):
            lv = {n: leaves[n].detach().clone().requires_grad_(rg) for n in names}

            def frame_(v):
                out = GaussianRasterizer(settings(cams[v]))(means3D=lv["means3D"], means2D=None, shs=lv["shs"], opacities=lv["opacities"],
                                                            scales=lv["scales"], rotations=lv["rotations"], uvs=lv["uvs"],
                                                            gradient_uvs=juv, texture=lv["texture"], extra_attrs=None)
                return out[0]
            for v in vs[:4]:
                frame_(v)
            torch.cuda.synchronize(dev)
            c0 = time.perf_counter()
            for v in vs:
                frame_(v)
            torch.cuda.synchronize(dev)
            res[label + "_views_per_s"] = round(len(vs) / (time.perf_counter() - c0), 1)
        RZ.GEOM_CACHE = saved_gc
        res["ratio"] = round(res["inputs_require_grad_views_per_s"] / res["no_grad_inputs_views_per_s"], 4)
        res["note"] = ("views one after the other on one stream, a new rasterizer module per frame as render/uv_tex_render.py:40 builds; "
                       "with inputs that require grad the forward leaves K6's hand-off to a backward that never comes (LAZY_HANDOFF)")
        retex = res

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(scene, cams[my_views[0]], W, H, with_bwd)

    if rank == 0:
        line = {
            "metric": ("fwd+bwd views/sec @800x800, 300k Gaussians + 1024^2 texture; HBM GB/s vs roofline"
                       if (args.workload == "c3" and not args.tex_res) else f"{mode} views/sec ({args.workload}" + (f", R={R}" if args.tex_res else "") + ")") if not untextured
                      else f"{mode} views/sec ({args.workload}, untextured diff_gauss surface)",
            "value": round(value, 3), "unit": "views/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "value_long": round(value_long, 3),
            "value_long_note": f"the same step loop run again for {long_elapsed:.2f} s ({long_steps} steps) after the timed region",
            "build_id": _lib.BUILD_ID,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: N={N} Gaussians, cubemap 6x{R}x{R}x3 f32, {W}x{H}, {mode}, sh_degree 3",
                       "views_per_step_per_gpu": args.views_per_step, "global_views_per_step": args.views_per_step * world,
                       "num_rendered_D": s.D, "D_eff": D_eff, "parallelism": f"views sharded dp{world}",
                       "view_pipeline": (f"{args.streams} HIP streams, order={args.order}"
                                         + (", forwards begun one view per stream ahead" if (args.prefetch and not untextured and with_bwd) else ""))
                       if args.streams > 1 else "serial",
                       "view_sharding": "LPT by per-view instance count D" if world > 1 else "all views on the one GPU",
                       "grad_allreduce": (("RCCL" if backend == "nccl" else backend + " (host-staged rehearsal)")
                                          + " SUM of one flat f32 bucket per step, as two segments on a side stream "
                                            "(texture after the last reduce kernel, per-Gaussian after the last K8)"
                                          + ("; the SH-coefficient run summed as bf16 on the wire" if args.wire == "bf16-sh" else "")) if (world > 1 or force_dist) else "none (1 GPU)",
                       "grad_allreduce_measured": ({"collectives_timed": len(comm), "steps_timed": args.steps,
                                                    "bytes_per_step": int(sum(b for b, _ in comm) / max(args.steps, 1)),
                                                    "ms_per_step_on_comm_stream": round(sum(m for _, m in comm) / max(args.steps, 1), 4),
                                                    "world": world} if comm else None)},
            "ms_per_view": round(1e3 * elapsed / (args.steps * args.views_per_step), 4),
            "ms_per_step_percentiles": {"p10": round(pct(0.1), 4), "median": round(pct(0.5), 4), "p90": round(pct(0.9), 4), "max": round(pct(1.0), 4), "slowest_step": int(np.argmax(step_raw)),
                                        "source": "torch.cuda.Event per step on the op's stream, this rank"},
            "per_rank": rank_table,
            "reference_call_pattern": compat,
            "reference_iteration": ref_iter,
            "retexture_pattern": retex,
            "alg_bytes_per_view": view_bytes,
            "pipeline_GBps": round(view_bytes * value / world / 1e9, 2),
            "pipeline_frac_of_hbm_peak": round(view_bytes * value / world / 1e9 / HBM_PEAK_GBS, 5),
            "hbm_copy_ceiling_GBps_measured": hbm_measured,
            "roofline": roofline,
            "kernels": {k: {kk: (round(vv, 2) if isinstance(vv, float) else vv) for kk, vv in v.items()}
                        for k, v in kinfo.items()},
            "kernels_note": f"{DOMINANT}: HIP events inside the (pipelined) timed region; the others: two extra fully bracketed "
                            "steps after it, views one after the other on one stream",
            "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
