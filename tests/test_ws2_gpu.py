"""C4 rehearsal on ONE GPU: the rasterizer under world_size 2.

Two processes share cuda:0 under the gloo backend (RCCL refuses two ranks on one device; GradBucket.all_reduce stages the
bucket through pinned host memory for any backend that is not nccl).  Each rank renders its shard of 16 views at C3 size
through ViewPipeline + the fused gradient sink, the buckets are all-reduced, and the result must equal the bucket one
process accumulates over the same 16 views.  Also: `bench.py --gpus 2` runs and parses under the driver's launcher."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NUM_VIEWS = 16


def _free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _accumulate(views, dev, depth):
    """Bucket of fwd+bwd over `views` of the C3 scene (300k Gaussians, 6x1024^2x3 cubemap, 800x800)."""
    import math
    from texgs import synth
    from texgs.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
    from texgs.multiview import GradBucket, ViewPipeline
    N, R, W, H = 300_000, 1024, 800, 800
    scene = synth.make_scene(N, R, seed=0)
    cams = synth.fibonacci_cameras(NUM_VIEWS, W, H)
    names = ["means3D", "shs", "opacities", "scales", "rotations", "uvs", "texture"]
    leaves = {n: getattr(scene, n).to(dev).requires_grad_(True) for n in names}
    juv = scene.gradient_uvs.to(dev)
    m2 = torch.zeros(N, 3, device=dev, requires_grad=True)
    bucket = GradBucket([leaves[n] for n in names] + [m2])
    bg = torch.zeros(3, device=dev)
    g = torch.Generator().manual_seed(1234)
    P = W * H
    g_img = ((torch.rand(3, H, W, generator=g) > 0.5).float() * 2 - 1).to(dev) / (3 * P)
    g_alpha = ((torch.rand(1, H, W, generator=g) > 0.5).float() * 2 - 1).to(dev) / P

    def fwd(v):
        cam = cams[v]
        st = GaussianRasterizationSettings(
            image_height=H, image_width=W, tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5), bg=bg,
            scale_modifier=1.0, viewmatrix=cam.world_view_transform.to(dev), projmatrix=cam.full_proj_transform.to(dev),
            sh_degree=3, campos=cam.camera_center.to(dev), prefiltered=False, debug=False)
        return GaussianRasterizer(st, grad_sink=bucket)(
            means3D=leaves["means3D"], means2D=m2, shs=leaves["shs"], opacities=leaves["opacities"], scales=leaves["scales"],
            rotations=leaves["rotations"], uvs=leaves["uvs"], gradient_uvs=juv, texture=leaves["texture"], extra_attrs=None)

    def bwd(out):
        torch.autograd.backward([out[0], out[3]], [g_img, g_alpha])
    bucket.zero()
    pipe = ViewPipeline(dev, depth=depth)
    pipe.run(list(views), fwd, bwd, sink=bucket, order="accumulate")
    torch.cuda.synchronize(dev)
    pipe.close()
    return bucket


def _rank_worker(rank, world, port, q):
    try:
        sys.path[:0] = [ROOT, os.path.join(ROOT, "texture-gs_amd")]
        import torch.distributed as dist
        from texgs.multiview import shard_views
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        bucket = _accumulate(shard_views(NUM_VIEWS, rank, world), dev, depth=3)
        bucket.all_reduce(dist)
        torch.cuda.synchronize(dev)
        out = None
        if rank == 0:
            reduced = bucket.flat.double().cpu()
            whole = _accumulate(range(NUM_VIEWS), dev, depth=1).flat.double().cpu()
            rel = float((reduced - whole).norm() / whole.norm())
            out = dict(rel=rel, norm=float(whole.norm()), numel=int(whole.numel()))
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, out))
    except Exception as e:          # surface the failure in the parent instead of a silent timeout
        import traceback
        q.put((rank, dict(error=f"{e!r}\n{traceback.format_exc()}")))


@pytest.mark.gpu
def test_c4_two_ranks_on_one_gpu_equal_single_process(lib_built):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=900) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
        if p.is_alive():
            p.kill()
    for r in (0, 1):
        assert not (res[r] and "error" in res[r]), res[r]["error"]
    out = res[0]
    import helpers as Hh
    Hh.report("c4/ws2_one_gpu_gloo/C3_16_views", **out)
    assert out["numel"] > 3e7 and out["norm"] > 0
    assert out["rel"] < 1e-5, out          # 8-rank grads == single-rank sum (1e-5 rel; atomics order differs)


@pytest.mark.gpu
def test_bench_gpus2_under_driver_launcher_parses(lib_built):
    """`python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2` (the driver's launch line) with the two ranks
    rehearsed on one device: one JSON line, n_gpus 2, value = views of both ranks / max-over-ranks time."""
    env = dict(os.environ, TEXGS_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline", "--no-kernel-table"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 2 and j["unit"] == "views/s" and j["scaling"] == "weak"
    assert j["config"]["global_views_per_step"] == 16 and j["value"] > 0
    assert "gloo" in j["config"]["grad_allreduce"]
    # the per-rank table of an N > 1 line: both ranks report their views, instance counts and step split
    pr = j["per_rank"]
    assert pr["ranks_seen"] == 2 and sorted(r["rank"] for r in pr["ranks"]) == [0, 1]
    assert all(r["views"] == 32 and r["D_sum"] > 0 and r["render_ms_per_step"] > 0 for r in pr["ranks"])
    assert j["value_long"] > 0
    import helpers as Hh
    Hh.report("c4/bench_gpus2_one_gpu_gloo", views_per_s=j["value"], ms_per_step=j["ms_per_step"])


@pytest.mark.gpu
def test_bench_gpus2_bf16_sh_wire(lib_built):
    """The wire-size lever (`--wire bf16-sh`: the SH-coefficient run of the bucket summed as bf16, three collectives per step instead
    of two) through the driver's launch line, two ranks rehearsed on one device."""
    env = dict(os.environ, TEXGS_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "c1", "--steps", "2",
           "--warmup", "1", "--no-cpu-baseline", "--no-kernel-table", "--wire", "bf16-sh"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["per_rank"]["wire"] == "bf16-sh" and "bf16" in j["config"]["grad_allreduce"]


@pytest.mark.gpu
def test_bench_gpus8_code_path_on_one_gpu(lib_built):
    """BASELINE configs[3] is 8 ranks; no 8-GPU node has been available to any round.  This runs `bench.py --gpus 8` exactly as the
    driver launches it -- 8 processes under torch.distributed.run -- with the ranks rehearsed on the one device (gloo, the bucket
    staged through host memory) on the plumbing-sized workload C1: the LPT quotas (8 views per rank of 64), the two-segment
    collective order on 8 ranks and the max-over-ranks timing all execute; one well-formed JSON line comes back."""
    env = dict(os.environ, TEXGS_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "8", "--workload", "c1", "--steps", "2",
           "--warmup", "1", "--no-cpu-baseline", "--no-kernel-table"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 8 and j["steps"] == 2 and j["scaling"] == "weak" and j["value"] > 0
    assert j["config"]["global_views_per_step"] == 64 and "LPT" in j["config"]["view_sharding"]
    assert "gloo" in j["config"]["grad_allreduce"] and "two segments" in j["config"]["grad_allreduce"]
    import helpers as Hh
    Hh.report("c4/bench_gpus8_one_gpu_gloo_c1", views_per_s=j["value"], ms_per_step=j["ms_per_step"])
