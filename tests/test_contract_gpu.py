"""-m gpu: the operator used the way its callers use it, and the configurations of BASELINE.json that the plain parity
files do not reach.

* a4  -- the reference glue's call pattern (tests/glue_contract.py): non-leaf zero `means2D` with retain_grad, [N,1]
         sigmoid opacities, exp scales, normalised rotations, uvs from a differentiable UV map, detached [n,3i+j]
         Jacobian, dict packaging, `radii > 0`; gradients reach the RAW parameters and `viewspace_points.grad[:, :2]`.
* ADVICE r1 -- zero_grad(set_to_none=True) between steps with a GradBucket; in-place modification between forward and
         backward; second backward.
* C4  -- 8 views split over two "virtual ranks" (each with its own bucket + grad_sink) sum to the single-bucket run; the
         same through a 1-rank RCCL group.
* ViewPipeline -- the views of a step over 2-3 HIP streams (all three ordering modes) against the serial loop.
* C5  -- 1M Gaussians / 2048^2 cubemap / 1600x1200 against the C oracle (integer stages bit-exact, fwd, bwd).
* wave_ops.h self-test on hardware.
"""
import math
import os

import numpy as np
import pytest
import torch

from texgs import synth
from oracle import texgs_torch as O
import helpers as Hh
import glue_contract as GC

pytestmark = pytest.mark.gpu


class _OracleModule:
    """The torch float64 oracle behind the same two names the glue imports."""
    GaussianRasterizationSettings = O.Settings

    class GaussianRasterizer:
        def __init__(self, raster_settings):
            self.st = raster_settings

        def __call__(self, means3D, means2D, shs, opacities, scales, rotations, uvs, gradient_uvs, texture, extra_attrs):
            return O.rasterize(means3D, means2D, shs, opacities, scales, rotations, uvs, gradient_uvs, texture, self.st)


def test_reference_glue_call_pattern(lib_built):
    import diff_gauss_uv_tex as M
    dev = torch.device("cuda:0")
    scene = synth.make_scene(1200, 64, seed=31, scale_mean=0.035, random_jacobian=True)
    cam32 = synth.fibonacci_cameras(4, 224, 160)[2]
    bg = torch.tensor([0.1, 0.3, 0.2])
    target, nhat = synth.make_targets(160, 224, seed=8)

    def run(module, device, dtype):
        cam = cam32._replace(world_view_transform=cam32.world_view_transform.to(device),
                             full_proj_transform=cam32.full_proj_transform.to(device),
                             camera_center=cam32.camera_center.to(device))
        model = GC.ToyTexturedGaussians(scene, device, dtype, sh_degree=2)
        pkg = GC.render_like_reference_glue(module, cam, model, bg.to(device))
        assert set(pkg) == {"render", "depth", "norm", "alpha", "viewspace_points", "visibility_filter", "extra", "radii"}
        loss = synth.synthetic_loss(pkg["render"], pkg["alpha"], pkg["norm"], target.to(device, dtype), nhat.to(device, dtype)) \
            + 0.05 * pkg["depth"].mean()
        loss.backward()
        return pkg, model

    pkg, model = run(M, dev, torch.float32)
    ref, rmodel = run(_OracleModule, torch.device("cpu"), torch.float64)
    assert pkg["extra"] is None and pkg["radii"].shape == (1200,)
    assert pkg["visibility_filter"].dtype == torch.bool
    vis = pkg["visibility_filter"].cpu()
    assert int((vis != ref["visibility_filter"]).sum()) <= 1
    assert float(torch.max(torch.zeros(1200, device=dev), pkg["radii"]).max()) > 0       # models/gaussian3d.py:431 usage
    # the non-leaf grad carrier got its gradient (densification statistic, models/gaussian3d.py:334-336)
    vg = pkg["viewspace_points"].grad
    assert vg is not None and vg.shape == (1200, 3)
    ok, msg = Hh.grad_close(vg[vis, :2].cpu(), ref["viewspace_points"].grad[vis, :2], label="glue/viewspace_points.grad[vis,:2]")
    assert ok, msg
    assert float(vg[:, 2].abs().max()) == 0.0
    for (name, got), exp in zip(model.leaves().items(), rmodel.leaves().values()):
        assert got.grad is not None, name
        ok, msg = Hh.grad_close(got.grad.cpu(), exp.grad, label=f"glue/raw_param/{name}")
        assert ok, (name, msg)


def _leaves(scene, dev):
    names = ["means3D", "shs", "opacities", "scales", "rotations", "uvs", "texture"]
    return names, [getattr(scene, n).clone().to(dev).requires_grad_(True) for n in names]


def _view(rast_cls, st, leaves, m2, juv, sink=None):
    m3, shs, op, sc, rot, uv, tex = leaves
    return rast_cls(st, grad_sink=sink)(means3D=m3, means2D=m2, shs=shs, opacities=op, scales=sc, rotations=rot, uvs=uv,
                                        gradient_uvs=juv, texture=tex, extra_attrs=None)


def test_zero_grad_set_to_none_between_steps(lib_built):
    """after optimizer.zero_grad(set_to_none=True) the fused sink must not accumulate into a buffer
    the optimizer no longer sees."""
    from texgs.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
    from texgs.multiview import GradBucket
    dev = torch.device("cuda:0")
    scene = synth.make_scene(800, 32, seed=2, scale_mean=0.04)
    cam = synth.fibonacci_cameras(4, 128, 96)[1]
    st = Hh.settings_for(cam, 2, torch.zeros(3), device=dev, cls=GaussianRasterizationSettings)
    target, nhat = synth.make_targets(96, 128, seed=1)
    juv = scene.gradient_uvs.to(dev)

    def step_grads(fused, none_between):
        names, leaves = _leaves(scene, dev)
        m2 = torch.zeros(800, 3, device=dev, requires_grad=True)
        params = leaves + [m2]
        bucket = GradBucket(params) if fused else None
        opt = torch.optim.SGD(params, lr=0.0)
        for it in range(2):
            out = _view(GaussianRasterizer, st, leaves, m2, juv, sink=bucket)
            synth.synthetic_loss(out[0], out[3], out[2], target.to(dev), nhat.to(dev)).backward()
            if it == 0:
                opt.step()
                if none_between:
                    opt.zero_grad(set_to_none=True)
                elif fused:
                    bucket.zero()
                else:
                    opt.zero_grad(set_to_none=False)
        assert all(p.grad is not None for p in params), "a parameter lost its gradient"
        return [p.grad.detach().clone().cpu() for p in params]
    ref = step_grads(False, False)
    for fused, none_between in [(True, True), (True, False), (False, True)]:
        got = step_grads(fused, none_between)
        for a, b in zip(got, ref):
            assert Hh.rel_err(a, b) < 1e-4, (fused, none_between)


def test_inplace_update_and_second_backward_are_errors(lib_built):
    from texgs.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
    dev = torch.device("cuda:0")
    scene = synth.make_scene(300, 16, seed=4, scale_mean=0.06)
    cam = synth.fibonacci_cameras(4, 64, 48)[0]
    st = Hh.settings_for(cam, 1, torch.zeros(3), device=dev, cls=GaussianRasterizationSettings)
    juv = scene.gradient_uvs.to(dev)
    names, leaves = _leaves(scene, dev)
    m2 = torch.zeros(300, 3, device=dev, requires_grad=True)
    out = _view(GaussianRasterizer, st, leaves, m2, juv)
    with torch.no_grad():
        leaves[6].mul_(0.5)                                   # change_texture-style in-place update before backward
    with pytest.raises(RuntimeError, match="modified in place"):
        out[0].sum().backward()
    out = _view(GaussianRasterizer, st, leaves, m2, juv)
    out[0].sum().backward(retain_graph=True)
    with pytest.raises(RuntimeError, match="second time"):
        out[0].sum().backward()


def test_c4_semantics_two_virtual_ranks_equal_single_bucket(lib_built):
    """BASELINE configs[3] semantics on one GPU: views sharded over ranks, per-rank fused accumulation, SUM == the
    single-rank accumulation over all views (1e-5 relative; atomics order differs)."""
    from texgs.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
    from texgs.multiview import GradBucket, shard_views
    dev = torch.device("cuda:0")
    scene = synth.make_scene(20000, 256, seed=0, scale_mean=0.012)
    cams = synth.fibonacci_cameras(8, 400, 304)
    juv = scene.gradient_uvs.to(dev)
    target, nhat = synth.make_targets(304, 400, seed=3)

    def accumulate(views):
        names, leaves = _leaves(scene, dev)
        m2 = torch.zeros(20000, 3, device=dev, requires_grad=True)
        bucket = GradBucket(leaves + [m2])
        bucket.zero()
        for v in views:
            st = Hh.settings_for(cams[v], 3, torch.zeros(3), device=dev, cls=GaussianRasterizationSettings)
            out = _view(GaussianRasterizer, st, leaves, m2, juv, sink=bucket)
            synth.synthetic_loss(out[0], out[3], out[2], target.to(dev), nhat.to(dev)).backward()
        return bucket
    whole = accumulate(range(8)).flat.double()
    parts = sum(accumulate(shard_views(8, r, 2)).flat.double() for r in range(2))
    rel = float((whole - parts).norm() / whole.norm())
    Hh.report("c4_semantics/2_virtual_ranks_vs_single", rel_l2=rel, max_abs=float((whole - parts).abs().max()),
              grad_max=float(whole.abs().max()))
    assert rel < 1e-5, rel


@pytest.mark.parametrize("depth,order,prefetch", [(2, "backward", False), (2, "accumulate", False), (2, "none", False), (3, "none", False),
                                                  (3, "accumulate", False), (3, "accumulate", True), (2, "backward", True)])
def test_view_pipeline_streams_equal_serial(lib_built, depth, order, prefetch):
    """texgs.multiview.ViewPipeline: the views of a step pipelined over HIP streams accumulate the same bucket as the
    serial loop (per-Gaussian sums: same K8 order in "backward"/"accumulate", a different association in "none"; the
    texture gradient and K7's moment sums go through atomics either way)."""
    from texgs.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
    from texgs.multiview import GradBucket, ViewPipeline
    dev = torch.device("cuda:0")
    scene = synth.make_scene(20000, 256, seed=5, scale_mean=0.012)
    cams = synth.fibonacci_cameras(7, 400, 304)
    juv = scene.gradient_uvs.to(dev)
    target, nhat = synth.make_targets(304, 400, seed=3)
    target, nhat = target.to(dev), nhat.to(dev)
    sts = [Hh.settings_for(c, 3, torch.zeros(3), device=dev, cls=GaussianRasterizationSettings) for c in cams]

    from texgs import rasterizer as RZ

    def accumulate(pipe_depth, pipe_order, steps=2, pre=False):
        names, leaves = _leaves(scene, dev)
        m2 = torch.zeros(20000, 3, device=dev, requires_grad=True)
        bucket = GradBucket(leaves + [m2])
        pipe = ViewPipeline(dev, depth=pipe_depth)
        images = []

        def fwd(v):
            out = _view(GaussianRasterizer, sts[v], leaves, m2, juv, sink=bucket)
            return out, synth.synthetic_loss(out[0], out[3], out[2], target, nhat)

        def bwd(obj):
            obj[1].backward()

        def pre_fn(v):          # (the forwards of a step begun one view per stream ahead: GaussianRasterizer.prefetch)
            m3, shs, op, sc, rot, uv, tex = leaves
            GaussianRasterizer(sts[v], grad_sink=bucket).prefetch(means3D=m3, means2D=m2, shs=shs, opacities=op, scales=sc, rotations=rot,
                                                                  uvs=uv, gradient_uvs=juv, texture=tex)
        for _ in range(steps):                        # two steps: the replicas / scratch must come back clean
            bucket.zero()
            res = pipe.run(range(7), fwd, bwd, sink=bucket, order=pipe_order, prefetch_fn=pre_fn if pre else None)
            torch.cuda.synchronize()
            images = [r[0][0].detach().clone() for r in res]
            assert not any(RZ._PREFETCH.values()), "a begun forward was not picked up by its view"
        assert bucket.before_accumulate is None and bucket.active == 0
        return bucket.flat.double().clone(), images
    whole, img_s = accumulate(1, "backward")
    got, img_p = accumulate(depth, order, pre=prefetch)
    for a, b in zip(img_s, img_p):
        assert torch.equal(a, b), "forward images differ between the serial and the pipelined run"
    rel = float((whole - got).norm() / whole.norm())
    Hh.report(f"view_pipeline/depth{depth}_{order}{'_prefetch' if prefetch else ''}_vs_serial", rel_l2=rel, max_abs=float((whole - got).abs().max()),
              grad_max=float(whole.abs().max()))
    assert rel < 1e-5, rel


def test_forward_prefetch_is_the_same_forward(lib_built):
    """GaussianRasterizer.prefetch begins a forward (K1 + the instance-count readback + K2); the same call through forward() finishes
    it: bit-identical outputs and the same gradients as a forward that was never begun early, with other work (another view's forward
    and backward) queued on the stream in between.  A begun forward whose inputs changed in between is not picked up."""
    from texgs.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
    from texgs import rasterizer as RZ
    dev = torch.device("cuda:0")
    scene = synth.make_scene(20000, 256, seed=7, scale_mean=0.012)
    cams = synth.fibonacci_cameras(3, 400, 304)
    juv = scene.gradient_uvs.to(dev)
    sts = [Hh.settings_for(c, 3, torch.zeros(3), device=dev, cls=GaussianRasterizationSettings) for c in cams]
    target, nhat = synth.make_targets(304, 400, seed=3)
    target, nhat = target.to(dev), nhat.to(dev)

    def run(pre):
        names, leaves = _leaves(scene, dev)
        m2 = torch.zeros(20000, 3, device=dev, requires_grad=True)
        m3, shs, op, sc, rot, uv, tex = leaves
        kw = dict(means3D=m3, means2D=m2, shs=shs, opacities=op, scales=sc, rotations=rot, uvs=uv, gradient_uvs=juv, texture=tex)
        if pre:
            GaussianRasterizer(sts[1]).prefetch(**kw)
            assert sum(len(v) for v in RZ._PREFETCH.values()) == 1
        out0 = _view(GaussianRasterizer, sts[0], leaves, m2, juv)               # another view in between, forward and backward
        synth.synthetic_loss(out0[0], out0[3], out0[2], target, nhat).backward()
        g0 = [l.grad.clone() for l in leaves]
        for l in leaves:
            l.grad = None
        out1 = _view(GaussianRasterizer, sts[1], leaves, m2, juv)
        assert not any(RZ._PREFETCH.values())
        synth.synthetic_loss(out1[0], out1[3], out1[2], target, nhat).backward()
        torch.cuda.synchronize()
        return [o.detach().clone() for o in out1[:5]], [l.grad.clone() for l in leaves], g0
    outs_a, grads_a, g0_a = run(False)
    outs_b, grads_b, g0_b = run(True)
    for a, b in zip(outs_a, outs_b):
        assert torch.equal(a, b)
    for n, a, b in zip(["means3D", "shs", "opacities", "scales", "rotations", "uvs", "texture"], grads_a, grads_b):
        rel = float((a.double() - b.double()).norm() / a.double().norm())
        assert rel < 1e-6, (n, rel)                      # (atomics order in K7's moment sums / the overflow path)
    # inputs modified after the begin: the pending forward must not be used
    names, leaves = _leaves(scene, dev)
    m2 = torch.zeros(20000, 3, device=dev, requires_grad=True)
    m3, shs, op, sc, rot, uv, tex = leaves
    kw = dict(means3D=m3, means2D=m2, shs=shs, opacities=op, scales=sc, rotations=rot, uvs=uv, gradient_uvs=juv, texture=tex)
    GaussianRasterizer(sts[2]).prefetch(**kw)
    with torch.no_grad():
        m3.mul_(1.01)
    out = _view(GaussianRasterizer, sts[2], leaves, m2, juv)
    assert sum(len(v) for v in RZ._PREFETCH.values()) == 1, "a stale begun forward was consumed"
    with torch.no_grad():
        ref = _view(GaussianRasterizer, sts[2], [l.detach() for l in leaves], m2.detach(), juv)
    assert torch.equal(out[0], ref[0]) and torch.equal(out[4], ref[4])
    RZ.release_scratch()
    assert not RZ._PREFETCH


def test_c4_one_rank_rccl_group_allreduce(lib_built):
    """The N>1 code path of bench.py (RCCL all-reduce of the flat bucket, mean over views) under a 1-rank RCCL group."""
    import torch.distributed as dist
    from texgs.multiview import GradBucket
    dev = torch.device("cuda:0")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29631")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        p = [torch.randn(1000, 3, device=dev, requires_grad=True), torch.randn(6, 8, 8, 3, device=dev, requires_grad=True)]
        b = GradBucket(p)
        b.zero()
        sum((x * x).sum() for x in p).backward()
        before = b.flat.clone()
        b.all_reduce(dist, average_over=4)
        torch.cuda.synchronize()
        assert torch.allclose(b.flat, before / 4.0)
        assert p[0].grad.data_ptr() == b.flat.data_ptr()
        # the asynchronous segments on the side stream under RCCL itself (one rank: a sum over one term): f32 segments come back bit for
        # bit, a segment sent as bf16 (the wire-size lever, texgs.multiview) comes back as its bf16 rounding; the timed collectives report
        # their WIRE bytes
        ref = torch.randn_like(b.flat)
        b.flat.copy_(ref)
        seg0, seg1 = b.segment_of([p[0]]), b.segment_of([p[1]])
        b.all_reduce_async(dist, seg1, timing=True)
        b.all_reduce_async(dist, seg0, timing=True, wire_dtype=torch.bfloat16)
        out = b.wait().clone()
        torch.cuda.synchronize()
        assert torch.equal(out[seg1[0]:seg1[0] + seg1[1]], ref[seg1[0]:seg1[0] + seg1[1]])
        assert torch.equal(out[seg0[0]:seg0[0] + seg0[1]], ref[seg0[0]:seg0[0] + seg0[1]].to(torch.bfloat16).float())
        tm = b.comm_timings()
        assert [t[0] for t in tm] == [seg1[1] * 4, seg0[1] * 2] and all(t[1] >= 0.0 for t in tm)
    finally:
        dist.destroy_process_group()


def test_c5_full_size_vs_c_oracle(lib_built):
    """BASELINE configs[4]: 1M Gaussians, 6x2048^2x3 cubemap, 1600x1200 (7500 tiles, 45-bit sort keys) -- integer
    stages bit-exact, forward and backward against the fp32 C oracle, one view."""
    from oracle import texgs_ref as CR
    from texgs.rasterizer import backward_raw
    scene = synth.make_scene(1_000_000, 2048, seed=0)
    cam = synth.fibonacci_cameras(64, 1600, 1200)[7]
    bg = torch.zeros(3)
    ref = CR.RefRun(scene, Hh.settings_for(cam, 3, bg))
    ref.forward()
    outs, s = Hh.hip_debug_state(scene, cam, 3, bg)
    N, D, t = ref.N, ref.D, s.tensors
    assert s.D == D and D > 2_000_000
    assert np.array_equal(outs[4].cpu().numpy(), ref.radii[:N])
    assert np.array_equal(t["tiles_touched"][:N].cpu().numpy().astype(np.uint32), ref.tiles[:N])
    assert np.array_equal(t["keys_sorted"][:D].cpu().numpy().view(np.uint64), ref.keys_sorted[:D])
    assert np.array_equal(t["point_list"][:D].cpu().numpy().astype(np.uint32), ref.point_list[:D])
    assert np.array_equal(t["ranges"].cpu().numpy().astype(np.uint32), ref.ranges)
    margin, gflag, tflag = ref.ambiguity(tau_fwd=Hh.TAU_FWD, tau_cell=Hh.tau_cell(2048), tau_relu=Hh.tau_relu(2048))
    got = torch.cat([outs[0], outs[1], outs[2], outs[3]], 0)
    nc = t["n_contrib"].cpu().numpy().astype(np.uint32)
    Hh.forward_attributed("hip_vs_c32/c5/fwd", got, ref, margin, n_contrib=nc)      # 1e-4 on every unambiguous pixel
    H, W = 1200, 1600
    g = torch.Generator().manual_seed(55)
    dout = torch.randn(8, H, W, generator=g) / (H * W)
    dev = outs[0].device
    res = backward_raw(s, dout[0:3].to(dev).contiguous(), dout[3:4].to(dev).contiguous(), dout[4:7].to(dev).contiguous(),
                       dout[7:8].to(dev).contiguous())
    gref = ref.backward(dout.numpy())
    # needle-shaped splats whose scale gradient is an ill-conditioned function of the per-Gaussian sums (fp32 atomics in arbitrary
    # order here, fp64 in the oracle): found by perturbing the oracle's own sums, ~1e-4 of the Gaussians
    sens = ref.accumulation_sensitive()
    Hh.report("hip_vs_c32/c5/bwd/accumulation_sensitive_rows", rows=int(sens.sum()), frac=float(sens.mean()))
    assert sens.mean() < 1e-3
    gflag = gflag | sens
    for name_, got_g in zip(["means3D", "means2D", "shs", "opacities", "scales", "rotations", "uvs", "texture"], res[:8]):
        Hh.grad_attributed(f"hip_vs_c32/c5/bwd/{name_}", got_g.cpu(), torch.tensor(gref[name_]),
                           tflag if name_ == "texture" else gflag)
    # at PAIR level: every row checked (54 % of them are excused above for having one near-edge pair among hundreds)
    Hh.pair_level_gradient_check("hip_vs_c32/c5/bwd_pair_level", ref, res, dout, 2048, sens)


def test_c5_band_limited_texture_literal_tolerance(lib_built):
    """C5 (1M Gaussians, 2048^2 cubemap, 1600x1200) with a band-limited texture: RGB at the LITERAL 1e-4 (no R / 1024 scaling), no
    cell-edge flags, flagged gradient rows < 5 %, zero unexplained pixels / rows."""
    scene = synth.make_scene(1_000_000, 2048, seed=0)
    cam = synth.fibonacci_cameras(64, 1600, 1200)[7]
    Hh.band_limited_parity("hip_vs_c32/c5_band_limited", scene, cam, torch.zeros(3))


def test_texture_gradient_bins_full_and_disabled_paths_agree(lib_built):
    """The binned two-pass texture gradient (K6 counts -> offsets -> K7 records -> per-bin LDS reduce) against its own
    fallbacks: a record buffer of 2000 slots (most footprints overflow to atomics) and bins disabled (every footprint
    through atomics).  Same sums; the counted list sizes are exactly what K7 appends; a second call needs no clean-up."""
    from texgs import rasterizer as RZ
    dev = torch.device("cuda:0")
    scene = synth.make_scene(3000, 96, seed=12, scale_mean=0.03)       # R = 96: 3x3 bins per face
    scene2 = synth.make_scene(3000, 80, seed=12, scale_mean=0.03)      # R = 80: partial bins at the right / bottom edge
    cam = synth.fibonacci_cameras(4, 240, 176)[3]
    target, nhat = synth.make_targets(176, 240, seed=4)
    saved = (RZ.USE_TEX_BINS, RZ.TEX_REC_CAP)
    for sc in (scene, scene2):
        res = {}
        for mode, (use, cap) in dict(bins=(True, 0), tiny=(True, 2000), off=(False, 0)).items():
            RZ.USE_TEX_BINS, RZ.TEX_REC_CAP = use, cap
            RZ.release_scratch()
            try:
                _, g = Hh.hip_run(sc, cam, 2, torch.zeros(3), with_grad=True, target=target, nhat=nhat)
                _, g2 = Hh.hip_run(sc, cam, 2, torch.zeros(3), with_grad=True, target=target, nhat=nhat)   # cursors were left clean
                if use:
                    torch.cuda.synchronize()
                    (sc_,) = RZ._SCRATCH.values()
                    nb = sc_.bins.nbins
                    # every list's overflow part was filled exactly to the end of the list (absolute cursors)
                    assert torch.equal(sc_.bins.cursor[:nb], sc_.bins.base[1:nb + 1]), mode
                    wanted = int(sc_.bins.cursor[nb])
                    assert wanted == int(sc_.bins.base[nb]) > 2000          # list sizes of the last call: what K6 counted
                    # the reduce kernel's launch order (k_bin_offsets): every bin once, longest list first (up to the 1/32-octave
                    # length classes of the counting sort)
                    order = sc_.bins.base[nb + 1:2 * nb + 1].long().cpu()
                    assert torch.equal(torch.sort(order).values, torch.arange(nb)), mode
                    lens = (sc_.bins.base[1:nb + 1] - sc_.bins.base[:nb]).long().cpu()[order]
                    assert bool((lens[1:].float() <= lens[:-1].float() * (1 + 1 / 32) + 1).all()), mode
            finally:
                RZ.USE_TEX_BINS, RZ.TEX_REC_CAP = saved
                RZ.release_scratch()
            assert Hh.rel_err(g2["texture"], g["texture"]) < 1e-5, mode
            res[mode] = g
        for mode in ("tiny", "off"):
            for name in ("texture", "uvs", "means3D"):
                r = Hh.rel_err(res[mode][name], res["bins"][name])
                Hh.report(f"texture_bins/{mode}_vs_bins/R{sc.texture.shape[1]}/{name}", rel_l2=r)
                assert r < 1e-5, (mode, name, r)


def test_texture_gradient_counts_and_no_count_path(lib_built):
    """K6's per-bin footprint counts are deterministic; a forward-only call keeps neither them nor the survivor lists (its
    backward is refused); a backward without the counts (every footprint through atomics) equals the binned one."""
    from texgs import rasterizer as RZ
    from texgs.rasterizer import GaussianRasterizationSettings, forward_raw, backward_raw
    dev = torch.device("cuda:0")
    scene = synth.make_scene(4000, 128, seed=41, scale_mean=0.03)
    cam = synth.fibonacci_cameras(4, 256, 192)[1]
    st = Hh.settings_for(cam, 3, torch.zeros(3), device=dev, cls=GaussianRasterizationSettings)
    t = lambda x: x.to(dev)
    args = [t(scene.means3D), t(scene.shs), t(scene.opacities), t(scene.scales), t(scene.rotations), t(scene.uvs),
            t(scene.gradient_uvs), t(scene.texture)]
    _, s1 = forward_raw(st, *args)
    _, s2 = forward_raw(st, *args)
    _, s0 = forward_raw(st, *args, for_backward=False)
    torch.cuda.synchronize()
    c1, c2 = s1.tensors["tex_bin_count"], s2.tensors["tex_bin_count"]
    assert torch.equal(c1, c2) and int(c1.sum()) > 10000
    assert s0.tensors["tex_bin_count"] is None and s0.tensors["survivors"] is None
    g = torch.Generator().manual_seed(2)
    dimg = (torch.randn(3, 192, 256, generator=g) * 1e-4).to(dev)
    with pytest.raises(RuntimeError, match="for_backward=False"):
        backward_raw(s0, dimg, None, None, None)
    RZ.release_scratch()
    a = backward_raw(s1, dimg, None, None, None)[7]
    torch.cuda.synchronize()
    (sc_,) = RZ._SCRATCH.values()
    assert int(sc_.bins.cursor[sc_.bins.nbins]) == int(c1.sum())       # the lists of that call were exactly the counts
    s2.tensors["tex_bin_count"] = None                                   # no counts: every footprint through atomics
    s2.img.tex_bin_count = None
    b = backward_raw(s2, dimg, None, None, None)[7]
    torch.cuda.synchronize()
    r = Hh.rel_err(a, b)
    Hh.report("texture_bins/counted_vs_atomics", rel_l2=r, records=int(c1.sum()))
    assert r < 1e-5


def _check_reservations(resv, counts):
    """Invariants of K6's per-block reservation tables (TexGSImage.tex_bin_resv: 64 direct-mapped entries {bin, offset, count})
    against the per-bin totals (first half of tex_bin_count): the ranges the blocks took of every bin's list tile [0, total[bin])
    exactly -- no gap, no overlap; a block lists a bin at most once, at the entry its (x, y) low bits select."""
    resv = (resv.cpu().long() & 0xFFFFFFFF).view(-1, 3, 64)
    counts = counts.cpu().long() & 0xFFFFFFFF
    nb = counts.numel() // 2
    bins, offs, cnts = resv[:, 0], resv[:, 1], resv[:, 2]
    used = bins != 0xFFFFFFFF
    assert bool((cnts[~used] == 0).all())
    b, o, c = bins[used], offs[used], cnts[used]
    assert bool((c > 0).all()) and bool((b < nb).all())
    key = torch.arange(bins.shape[0])[:, None].expand_as(bins)[used] * (1 << 32) + b
    assert key.unique().numel() == key.numel()                     # a bin at most once per block
    tot = torch.zeros(nb, dtype=torch.long).index_add_(0, b, c)
    assert torch.equal(tot, counts[:nb])
    order = torch.argsort(b * (1 << 32) + o)
    b, o, c = b[order], o[order], c[order]
    first = torch.ones_like(b, dtype=torch.bool)
    first[1:] = b[1:] != b[:-1]
    assert bool((o[first] == 0).all())
    assert bool((o[1:][~first[1:]] == (o + c)[:-1][~first[1:]]).all())
    return int(used.sum()), int(used.sum(1).max()), int(counts[nb:].sum())


def test_block_reservations_tile_the_record_lists(lib_built):
    """v12: K6 reserves, per 8x8 pixel block and texture bin, the block's range of the bin's record list (a 64-entry direct-mapped
    table per block); K7 fills exactly those ranges, no cursor.  The tables partition the reserved part of every list; the binned
    gradient equals the one through atomics -- also when most footprints miss the table (a coarse image of a fine texture: a
    block sees hundreds of bins; the surplus is counted and appended through the overflow cursor)."""
    from texgs import rasterizer as RZ
    from texgs.rasterizer import GaussianRasterizationSettings, forward_raw, backward_raw
    dev = torch.device("cuda:0")
    t = lambda x: x.to(dev)
    for name, (n, R, W, H, scale) in dict(usual=(4000, 128, 256, 192, 0.03), coarse=(6000, 1024, 48, 32, 0.03)).items():
        scene = synth.make_scene(n, R, seed=43, scale_mean=scale)
        cam = synth.fibonacci_cameras(4, W, H)[1]
        st = Hh.settings_for(cam, 3, torch.zeros(3), device=dev, cls=GaussianRasterizationSettings)
        args = [t(scene.means3D), t(scene.shs), t(scene.opacities), t(scene.scales), t(scene.rotations), t(scene.uvs),
                t(scene.gradient_uvs), t(scene.texture)]
        RZ.release_scratch()
        _, s1 = forward_raw(st, *args)
        torch.cuda.synchronize()
        entries, widest, overflow = _check_reservations(s1.tensors["tex_bin_resv"], s1.tensors["tex_bin_count"])
        records = int(s1.tensors["tex_bin_count"].sum())
        g = torch.Generator().manual_seed(2)
        dimg = (torch.randn(3, H, W, generator=g) * 1e-4).to(dev)
        a = backward_raw(s1, dimg, None, None, None)
        torch.cuda.synchronize()
        (sc_,) = RZ._SCRATCH.values()
        nb = sc_.bins.nbins
        assert int(sc_.bins.base[nb]) == records                          # the lists are exactly the counts ...
        assert torch.equal(sc_.bins.cursor[:nb], sc_.bins.base[1:nb + 1])  # ... and K7 filled every overflow part to its end
        RZ.release_scratch()
        _, s2 = forward_raw(st, *args)
        s2.tensors["tex_bin_count"] = None                                   # no counts: every footprint through atomics
        s2.img.tex_bin_count = None
        b = backward_raw(s2, dimg, None, None, None)
        torch.cuda.synchronize()
        r = Hh.rel_err(a[7], b[7])
        Hh.report(f"texture_bins/reservations/{name}", rel_l2=r, entries=entries, widest_table=widest, records=records,
                  overflow_records=overflow)
        assert r < 1e-5, (name, r)
        assert Hh.rel_err(a[0], b[0]) < 1e-5 and Hh.rel_err(a[6], b[6]) < 1e-5
        if name == "coarse":
            assert overflow > records // 10          # the overflow path really ran
    RZ.release_scratch()


def test_reservation_mismatch_between_k6_and_k7_stays_exact(lib_built):
    """the reduce sums the whole RESERVED range of every list, so the texture gradient is exact only while K7 fills exactly
    the slots K6 reserved.  Three safety nets cover a disagreement -- the end-of-block zero-fill of reservations K7 did not use up, the
    `pos < tend` test that sends surplus footprints to dL_dtexture directly, and the overflow path for bins the block's table does
    not hold -- and none of them runs while K6 and K7 agree.  Force a disagreement: between forward and backward the UV anchor (phi)
    of a third of the Gaussians is changed in K1's shading records, so K7's footprints land in other cells / bins than K6 counted;
    the record buffer still holds the non-zero records of an earlier backward.  The binned gradient must equal the one through
    atomics for the same (changed) records."""
    from texgs import rasterizer as RZ
    from texgs.rasterizer import GaussianRasterizationSettings, forward_raw, backward_raw
    dev = torch.device("cuda:0")
    t = lambda x: x.to(dev)
    scene = synth.make_scene(4000, 128, seed=44, scale_mean=0.03)
    cam = synth.fibonacci_cameras(4, 256, 192)[2]
    st = Hh.settings_for(cam, 3, torch.zeros(3), device=dev, cls=GaussianRasterizationSettings)
    args = [t(scene.means3D), t(scene.shs), t(scene.opacities), t(scene.scales), t(scene.rotations), t(scene.uvs),
            t(scene.gradient_uvs), t(scene.texture)]
    g = torch.Generator().manual_seed(3)
    dimg = (torch.randn(3, 192, 256, generator=g) * 1e-4).to(dev)
    pick = (torch.rand(4000, generator=g) < 0.33).to(dev)
    saved = RZ.GEOM_CACHE
    try:
        RZ.GEOM_CACHE = False
        RZ.release_scratch()
        _, s0 = forward_raw(st, *args)
        backward_raw(s0, dimg * 50.0, None, None, None)            # leaves non-zero records of ANOTHER magnitude in the stream's buffer
        res = {}
        for mode in ("binned", "atomics"):
            _, s = forward_raw(st, *args)
            rs = s.tensors["rec_shade"]
            phi = rs[:, 8:11].clone()
            rs[:, 8:11] = torch.where(pick[:, None], phi.roll(1, dims=1) * torch.tensor([1.0, -1.0, 1.0], device=dev), phi)
            if mode == "atomics":
                s.tensors["tex_bin_count"] = None
                s.img.tex_bin_count = None
            res[mode] = backward_raw(s, dimg, None, None, None)
            torch.cuda.synchronize()
        r = Hh.rel_err(res["binned"][7], res["atomics"][7])
        Hh.report("texture_bins/forced_k6_k7_mismatch", rel_l2=r, changed_gaussians=int(pick.sum()))
        assert float(res["atomics"][7].abs().sum()) > 0
        assert r < 1e-5, r
        # and the unperturbed gradient differs from it (the perturbation really moved footprints)
        _, s = forward_raw(st, *args)
        plain = backward_raw(s, dimg, None, None, None)
        assert Hh.rel_err(plain[7], res["atomics"][7]) > 1e-2
    finally:
        RZ.GEOM_CACHE = saved
        RZ.release_scratch()


def test_texture_gradient_scale_is_per_call(lib_built):
    """The reduce kernel's fixed-point scale is the max |dL/dpixel| of THE CALL (the backward resets a scratch word, K7 raises
    it): a backward with 1e6 x larger upstream gradients on the same stream / scratch must not coarsen the next one."""
    from texgs import rasterizer as RZ
    from texgs.rasterizer import GaussianRasterizationSettings, forward_raw, backward_raw
    dev = torch.device("cuda:0")
    scene = synth.make_scene(2500, 64, seed=31, scale_mean=0.03)
    cam = synth.fibonacci_cameras(4, 208, 160)[2]
    st = Hh.settings_for(cam, 2, torch.zeros(3), device=dev, cls=GaussianRasterizationSettings)
    t = lambda x: x.to(dev)
    g = torch.Generator().manual_seed(5)
    dimg = (torch.randn(3, 160, 208, generator=g) * 1e-5).to(dev)

    def tex_grad(scale):
        _, s = forward_raw(st, t(scene.means3D), t(scene.shs), t(scene.opacities), t(scene.scales), t(scene.rotations),
                           t(scene.uvs), t(scene.gradient_uvs), t(scene.texture))
        return backward_raw(s, dimg * scale, None, None, None)[7]
    RZ.release_scratch()
    ref = tex_grad(1.0).clone()
    big = tex_grad(1e6)
    again = tex_grad(1.0)
    torch.cuda.synchronize()
    assert Hh.rel_err(big, ref * 1e6) < 1e-5
    # 2^-42 of the image-wide bound per record: with the scale of the 1e6 x call left behind, this would be off by ~1e-7 * 1e6
    r = Hh.rel_err(again, ref)
    Hh.report("texture_bins/scale_per_call", rel_l2_after_big_call=r)
    assert r < 1e-6, r


def test_retexture_path_from_cross_image(lib_built):
    """retexture.py's path with texgs.texture_io: cross image (fixture: a 12-px-per-face sample of the reference's
    assets/textures/mosaic.png) -> resize -> change_texture -> forward-only renders at full SH degree and degree 0
    (models/texture_gaussian3d.py:499-511), against the oracle rendering the same texture."""
    from texgs import texture_io as TIO
    G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "texture_io.npz"))
    R = 48
    cross = torch.tensor(TIO.resize_bilinear_u8(G["cross_u8"], 3 * R, 4 * R).astype(np.float32) / 255.0)
    scene = synth.make_scene(2500, R, seed=17, scale_mean=0.03)
    tex = TIO.change_texture(scene.texture, cross, mode=0)
    assert tuple(tex.shape) == (6, R, R, 3)
    scene = scene._replace(texture=tex.contiguous())
    cam = synth.fibonacci_cameras(4, 208, 160)[0]
    bg = torch.zeros(3)
    for deg in (3, 0):
        ref, dbg, _ = Hh.oracle_run(scene, cam, deg, bg)
        out, _ = Hh.hip_run(scene, cam, deg, bg)
        amb = dbg["ambiguity"] < 1e-4
        res = Hh.forward_errors(f"retexture/mosaic/sh{deg}", out, ref, amb)
        assert res["image"][0] < 1e-4 and res["alpha"][0] < 1e-4
    # and back out: the cross image of the new texture is the (clamped) input modulated by mode 0
    back = TIO.cross_to_cube(TIO.texture_to_cross(tex))
    assert torch.allclose(back, TIO.sh02rgb(tex))


def test_wave_ops_primitives_on_hardware(lib_built):
    """csrc/wave_ops.h (DPP row_shl/shr exchanges, permlane16/32 swaps, transposing butterflies incl. the inline-asm
    bank-first one): every primitive equals the __shfl_xor formulation, exactly, for several random seeds."""
    from texgs import _lib
    lib = _lib.load()
    dev = torch.device("cuda:0")
    for seed in range(4):
        g = torch.Generator().manual_seed(seed)
        sd = (torch.randn(128, generator=g) * 3).to(dev)
        out = torch.full((576,), float("nan"), device=dev)
        _lib.check(lib.texgs_selftest_waveops(sd.data_ptr(), out.data_ptr(), torch.cuda.current_stream().cuda_stream),
                   "texgs_selftest_waveops")
        torch.cuda.synchronize()
        o = out.cpu().reshape(9, 64)
        for name, row in zip(["lane_xor4", "lane_xor8", "sum_xor16", "sum_xor32", "wave_sum", "reduce_transposed16",
                              "reduce_transposed32", "reduce32_bankfirst", "wave_max_i"], o):
            assert float(row.abs().max()) == 0.0, (name, seed, row)


def _stream_case():
    scene = synth.make_scene(4000, 96, seed=31, scale_mean=0.03, random_jacobian=True)
    cam = synth.fibonacci_cameras(6, 272, 208)[2]
    target, nhat = synth.make_targets(208, 272, seed=6)
    return scene, cam, target, nhat


@pytest.mark.parametrize("surface", ["textured", "frozen_texture"])
def test_item_stream_backward_equals_survivor_replay(lib_built, surface):
    """The opt-in K6 -> K7 item stream (texgs.h v15: K6 leaves {T, alpha_raw, Gaussian | pixel} per contributing pair, K7 walks the
    items instead of re-testing the survivor lists): forward images bit-identical, every gradient equal to the survivor-replay
    backward's up to fp32 summation order and the replay's recomputed transmittance (T /= 1 - alpha vs the forward's own T); the
    stream really ran (no overflow flag, every block's item count consistent with its pixels' contributor counts)."""
    from texgs import rasterizer as RZ
    scene, cam, target, nhat = _stream_case()
    saved = (RZ.USE_ITEMS, RZ.GEOM_CACHE)
    res, outs = {}, {}
    try:
        RZ.GEOM_CACHE = False      # (every call here renders the same view: a later forward must not re-use an earlier one's stream)
        RZ.USE_ITEMS = True        # the first view of a new size sizes its page buffer by guess; the next ones from what that one needed
        Hh.hip_run(scene, cam, 2, torch.zeros(3), with_grad=True, target=target, nhat=nhat)
        torch.cuda.synchronize()
        for mode in (False, True):
            RZ.USE_ITEMS = mode
            if surface == "textured":
                outs[mode], res[mode] = Hh.hip_run(scene, cam, 2, torch.tensor([0.2, 0.1, 0.0]), with_grad=True, target=target, nhat=nhat,
                                                   depth_weight=0.05)
            else:
                dev = torch.device("cuda:0")
                st = Hh.settings_for(cam, 2, torch.zeros(3), device=dev, cls=RZ.GaussianRasterizationSettings)
                lv = {n: getattr(scene, n).clone().to(dev).requires_grad_(n != "texture") for n in
                      ("means3D", "shs", "opacities", "scales", "rotations", "uvs", "texture")}
                out = RZ.GaussianRasterizer(st)(means3D=lv["means3D"], means2D=None, shs=lv["shs"], opacities=lv["opacities"],
                                                scales=lv["scales"], rotations=lv["rotations"], uvs=lv["uvs"],
                                                gradient_uvs=scene.gradient_uvs.to(dev), texture=lv["texture"], extra_attrs=None)
                synth.synthetic_loss(out[0], out[3], out[2], target.to(dev), nhat.to(dev)).backward()
                outs[mode] = [o.detach() for o in out[:4]]
                res[mode] = {n: t.grad for n, t in lv.items() if t.grad is not None}
        # the stream's own bookkeeping, through the raw path
        RZ.USE_ITEMS = True
        st = Hh.settings_for(cam, 2, torch.zeros(3), device=torch.device("cuda:0"), cls=RZ.GaussianRasterizationSettings)
        dv = {n: getattr(scene, n).to("cuda:0") for n in ("means3D", "shs", "opacities", "scales", "rotations", "uvs", "gradient_uvs", "texture")}
        _, s = RZ.forward_raw(st, dv["means3D"], dv["shs"], dv["opacities"].reshape(-1), dv["scales"], dv["rotations"], dv["uvs"],
                              dv["gradient_uvs"], dv["texture"])
        torch.cuda.synchronize()
        ctl, tail = s.tensors["item_ctl"], s.tensors["item_tail"]
        assert ctl is not None and int(ctl[_lib_const("ITEM_CTL_FLAG")]) == 0
        items = tail[:, 1].long()
        nc = s.tensors["n_contrib"]
        assert int(items.sum()) > 0
        # a block with a contributing pixel has items, a block without has none; no block has more items than pixels x list length
        H, W = nc.shape
        ty, tx = (H + 15) // 16, (W + 15) // 16
        pad = torch.zeros(ty * 16, tx * 16, dtype=torch.int64, device=nc.device)
        pad[:H, :W] = nc.long()
        blk = pad.reshape(ty, 2, 8, tx, 2, 8).permute(0, 3, 1, 4, 2, 5).reshape(ty * tx, 4, 64)      # [tile][block = 2 * row half + col half][pixel]
        any_contrib = (blk > 0).any(-1).reshape(-1)
        assert torch.equal(items > 0, any_contrib)
        assert bool((items <= blk.sum(-1).reshape(-1)).all())
    finally:
        RZ.USE_ITEMS, RZ.GEOM_CACHE = saved
        RZ.release_scratch()
    for a, b in zip(outs[False][:4], outs[True][:4]):
        assert torch.equal(a, b)
    for name in res[False]:
        r = Hh.rel_err(res[True][name], res[False][name])
        Hh.report(f"item_stream/{surface}/stream_vs_replay/{name}", rel_l2=r)
        assert r < 2e-5, (name, r)


def _lib_const(name):
    from texgs import _lib
    return getattr(_lib, name)


def test_item_stream_page_overflow_falls_back_to_the_replay(lib_built):
    """A page buffer that is too small (here: two pages per sub-pool) is not an error: K6 raises the overflow flag, the stream kernel
    does nothing and the survivor-replay kernel produces the gradients -- the same numbers as with the stream off."""
    from texgs import rasterizer as RZ
    scene, cam, target, nhat = _stream_case()
    saved = (RZ.USE_ITEMS, RZ.ITEM_PAGES_FIXED, RZ.GEOM_CACHE)
    res = {}
    try:
        RZ.GEOM_CACHE = False
        for mode, (use, pages) in dict(replay=(False, 0), starved=(True, 2)).items():
            RZ.USE_ITEMS, RZ.ITEM_PAGES_FIXED = use, pages
            RZ.release_scratch()
            _, res[mode] = Hh.hip_run(scene, cam, 2, torch.zeros(3), with_grad=True, target=target, nhat=nhat)
        RZ.USE_ITEMS, RZ.ITEM_PAGES_FIXED = True, 2
        st = Hh.settings_for(cam, 2, torch.zeros(3), device=torch.device("cuda:0"), cls=RZ.GaussianRasterizationSettings)
        dv = {n: getattr(scene, n).to("cuda:0") for n in ("means3D", "shs", "opacities", "scales", "rotations", "uvs", "gradient_uvs", "texture")}
        _, s = RZ.forward_raw(st, dv["means3D"], dv["shs"], dv["opacities"].reshape(-1), dv["scales"], dv["rotations"], dv["uvs"],
                              dv["gradient_uvs"], dv["texture"])
        torch.cuda.synchronize()
        ctl = s.tensors["item_ctl"]
        assert int(ctl[_lib_const("ITEM_CTL_FLAG")]) == 1
        pools = int(s.img.item_sub_pools)
        assert int(ctl[0:16 * pools:16].max()) * pools > int(s.img.item_page_cap)       # the cursors kept counting: what the view needs
    finally:
        RZ.USE_ITEMS, RZ.ITEM_PAGES_FIXED, RZ.GEOM_CACHE = saved
        RZ.release_scratch()
    for name in res["replay"]:
        r = Hh.rel_err(res["starved"][name], res["replay"][name])
        assert r < 2e-5, (name, r)        # (the same kernel twice: the run-to-run floor of its fp32 atomics)
