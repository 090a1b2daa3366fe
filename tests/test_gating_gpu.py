"""What the backward computes follows what the caller asks for (TexGSGrads.want, texgs.h): the four flavours of K7 against the
full one, autograd semantics of the default gradient path, and non-finite upstream gradients through the binned texture
gradient."""
import math

import pytest
import torch

import helpers as Hh
from texgs import synth

pytestmark = pytest.mark.gpu


def _setup(N=3000, R=96, W=240, H=176, seed=12, deg=2):
    from texgs.rasterizer import GaussianRasterizationSettings
    dev = torch.device("cuda:0")
    scene = synth.make_scene(N, R, seed=seed, scale_mean=0.03, random_jacobian=True)
    cam = synth.fibonacci_cameras(4, W, H)[3]
    st = Hh.settings_for(cam, deg, torch.tensor([0.1, 0.2, 0.0]), device=dev, cls=GaussianRasterizationSettings)
    target, nhat = synth.make_targets(H, W, seed=4)
    return dev, scene, st, target.to(dev), nhat.to(dev)


NAMES = ["means3D", "shs", "opacities", "scales", "rotations", "uvs", "texture"]


def _run(dev, scene, st, target, nhat, train):
    """Forward + backward through the public module with only the inputs in `train` requiring a gradient."""
    from texgs.rasterizer import GaussianRasterizer
    leaves = {n: getattr(scene, n).clone().to(dev).requires_grad_(n in train) for n in NAMES}
    m2 = torch.zeros(scene.means3D.shape[0], 3, device=dev, requires_grad="means2D" in train)
    out = GaussianRasterizer(st)(means3D=leaves["means3D"], means2D=m2, shs=leaves["shs"], opacities=leaves["opacities"],
                                 scales=leaves["scales"], rotations=leaves["rotations"], uvs=leaves["uvs"],
                                 gradient_uvs=scene.gradient_uvs.to(dev), texture=leaves["texture"], extra_attrs=None)
    synth.synthetic_loss(out[0], out[3], out[2], target, nhat).backward()
    g = {n: (None if leaves[n].grad is None else leaves[n].grad.clone()) for n in NAMES}
    g["means2D"] = None if m2.grad is None else m2.grad.clone()
    return [o.detach().clone() for o in out[:4]], g


def test_backward_flavours_equal_the_full_backward(lib_built):
    """Texture-only step (every Gaussian input frozen: models/texture_gaussian3d.py:439-440 steps only optimizer_tex before
    iteration 10 000), frozen-texture step, and the full step: the gradients that are computed are the same numbers, the ones
    nobody asked for are None, and the forward images are bit-identical (K6 only drops the footprint counting)."""
    dev, scene, st, target, nhat = _setup()
    full_out, full = _run(dev, scene, st, target, nhat, set(NAMES) | {"means2D"})
    tex_out, tex = _run(dev, scene, st, target, nhat, {"texture"})
    geo_out, geo = _run(dev, scene, st, target, nhat, set(NAMES) - {"texture"} | {"means2D"})
    for a, b, c in zip(full_out, tex_out, geo_out):
        assert torch.equal(a, b) and torch.equal(a, c)
    assert all(tex[n] is None for n in tex if n != "texture")
    assert geo["texture"] is None
    r = Hh.rel_err(tex["texture"], full["texture"])
    Hh.report("gating/texture_only_vs_full/texture", rel_l2=r)
    assert r < 3e-6           # (two differently shaped kernels: k7_lds vs k7_occ -- same numbers up to the order of their fp32 sums)
    for n in geo:
        if n == "texture":
            continue
        r = Hh.rel_err(geo[n], full[n])
        Hh.report(f"gating/frozen_texture_vs_full/{n}", rel_l2=r)
        assert r < 1e-5, (n, r)          # (same arithmetic; the accumulator-row atomics add in another order: rotations 2e-6)


def test_partial_gaussian_gradients_only(lib_built):
    """Only the opacity is trained: every other input comes back without a gradient, the opacity gradient equals the full run's."""
    dev, scene, st, target, nhat = _setup(seed=5)
    _, full = _run(dev, scene, st, target, nhat, set(NAMES) | {"means2D"})
    _, part = _run(dev, scene, st, target, nhat, {"opacities"})
    assert all(part[n] is None for n in part if n != "opacities")
    assert Hh.rel_err(part["opacities"], full["opacities"]) < 2e-6


def test_autograd_grad_leaves_dot_grad_untouched(lib_built):
    """Default gradient path = plain autograd: torch.autograd.grad returns the gradients and does not touch .grad; a second
    backward() accumulates into an existing .grad exactly as AccumulateGrad does; tensor hooks fire."""
    from texgs import rasterizer as RZ
    from texgs.rasterizer import GaussianRasterizer
    assert RZ.DIRECT_LEAF_GRADS is False
    dev, scene, st, target, nhat = _setup(seed=9)
    leaves = {n: getattr(scene, n).clone().to(dev).requires_grad_(True) for n in NAMES}
    juv = scene.gradient_uvs.to(dev)

    def loss():
        out = GaussianRasterizer(st)(means3D=leaves["means3D"], means2D=None, shs=leaves["shs"], opacities=leaves["opacities"],
                                     scales=leaves["scales"], rotations=leaves["rotations"], uvs=leaves["uvs"],
                                     gradient_uvs=juv, texture=leaves["texture"], extra_attrs=None)
        return synth.synthetic_loss(out[0], out[3], out[2], target, nhat)
    loss().backward()
    first = {n: leaves[n].grad.clone() for n in NAMES}
    seen = []
    h = leaves["texture"].register_hook(lambda g: seen.append(float(g.abs().sum())))
    got = torch.autograd.grad(loss(), [leaves[n] for n in NAMES])
    h.remove()
    assert len(seen) == 1 and seen[0] > 0
    for n, g in zip(NAMES, got):
        assert g is not None and Hh.rel_err(g, first[n]) < 1e-5, n
        assert torch.equal(leaves[n].grad, first[n]), f"autograd.grad changed {n}.grad"
    loss().backward()                                     # accumulates: .grad = 2 x first
    for n in NAMES:
        assert Hh.rel_err(leaves[n].grad, 2 * first[n]) < 1e-5, n


def test_opt_in_direct_leaf_grads_skips_hooked_leaves(lib_built):
    """TEXGS_DIRECT_LEAF_GRADS=1 (opt-in): kernels add into an existing .grad of hook-free leaves; a leaf with a hook still gets
    its gradient through autograd (the hook fires) and the sums are the same."""
    from texgs import rasterizer as RZ
    from texgs.rasterizer import GaussianRasterizer
    dev, scene, st, target, nhat = _setup(seed=10)
    juv = scene.gradient_uvs.to(dev)

    def two_backwards(direct):
        RZ.DIRECT_LEAF_GRADS = direct
        try:
            leaves = {n: getattr(scene, n).clone().to(dev).requires_grad_(True) for n in NAMES}
            fired = []
            leaves["texture"].register_hook(lambda g: fired.append(1))
            for _ in range(2):
                out = GaussianRasterizer(st)(means3D=leaves["means3D"], means2D=None, shs=leaves["shs"],
                                             opacities=leaves["opacities"], scales=leaves["scales"], rotations=leaves["rotations"],
                                             uvs=leaves["uvs"], gradient_uvs=juv, texture=leaves["texture"], extra_attrs=None)
                synth.synthetic_loss(out[0], out[3], out[2], target, nhat).backward()
            return {n: leaves[n].grad.clone() for n in NAMES}, len(fired)
        finally:
            RZ.DIRECT_LEAF_GRADS = False
    a, fa = two_backwards(False)
    b, fb = two_backwards(True)
    assert fa == 2 and fb == 2
    for n in NAMES:
        assert Hh.rel_err(b[n], a[n]) < 1e-5, n


def test_nonfinite_upstream_gradient_reaches_the_texture(lib_built):
    """An inf / NaN in dL/dimage (a GradScaler overflow step) must come out of the binned texture gradient as non-finite values on
    exactly the texels the plain atomic path marks, with every other texel unchanged -- not as finite garbage."""
    from texgs import rasterizer as RZ
    from texgs.rasterizer import forward_raw, backward_raw
    dev, scene, st, _, _ = _setup(N=2500, R=64, W=208, H=160, seed=31)
    t = lambda x: x.to(dev)
    args = [t(scene.means3D), t(scene.shs), t(scene.opacities), t(scene.scales), t(scene.rotations), t(scene.uvs),
            t(scene.gradient_uvs), t(scene.texture)]
    g = torch.Generator().manual_seed(5)
    dimg = (torch.randn(3, 160, 208, generator=g) * 1e-4).to(dev)
    _, s = forward_raw(st, *args)
    clean = backward_raw(s, dimg, None, None, None)[7].clone()
    alpha = forward_raw(st, *args, for_backward=False)[0][3][0]
    ys, xs = torch.nonzero(alpha > 0.5, as_tuple=True)
    y, x = int(ys[len(ys) // 2]), int(xs[len(xs) // 2])
    saved = RZ.USE_TEX_BINS
    for bad in (float("inf"), float("nan")):
        d2 = dimg.clone()
        d2[1, y, x] = bad
        res = {}
        for mode, use in (("bins", True), ("atomics", False)):
            RZ.USE_TEX_BINS = use
            RZ.release_scratch()
            try:
                _, s = forward_raw(st, *args)
                res[mode] = backward_raw(s, d2, None, None, None)[7].clone()
            finally:
                RZ.USE_TEX_BINS = saved
                RZ.release_scratch()
        nf_b, nf_a = ~torch.isfinite(res["bins"]), ~torch.isfinite(res["atomics"])
        assert int(nf_a.sum()) > 0 and torch.equal(nf_b, nf_a), (bad, int(nf_b.sum()), int(nf_a.sum()))
        ok = ~nf_a
        assert Hh.rel_err(res["bins"][ok], res["atomics"][ok]) < 1e-5
        # a clean call right after (same scratch, same stream) is not affected
        _, s = forward_raw(st, *args)
        again = backward_raw(s, dimg, None, None, None)[7]
        assert bool(torch.isfinite(again).all()) and Hh.rel_err(again, clean) < 1e-6


def _reference_iteration(dev, scene, cam_tensors, deg, target, nhat, recompute_activations):
    """models/texture_gaussian3d.py:318,375-389,410: render at the active degree, render again at degree 0 (same camera, same
    Gaussians), ONE backward of the summed losses.  recompute_activations: sigmoid / exp / normalize are evaluated again for the
    second render, as the reference's property getters do."""
    from texgs.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
    H, W, tfx, tfy, bg, vm, pm, cp = cam_tensors
    raw = dict(means3D=scene.means3D.clone(), shs=scene.shs.clone(), uvs=scene.uvs.clone(), texture=scene.texture.clone(),
               rotations=scene.rotations.clone(), scales=scene.scales.log(),
               opacities=torch.log(scene.opacities.clamp(1e-6, 1 - 1e-6) / (1 - scene.opacities.clamp(1e-6, 1 - 1e-6))))
    raw = {k: v.to(dev).requires_grad_(True) for k, v in raw.items()}
    juv = scene.gradient_uvs.to(dev)

    def acts():
        return dict(opacities=torch.sigmoid(raw["opacities"]), scales=torch.exp(raw["scales"]),
                    rotations=torch.nn.functional.normalize(raw["rotations"]))
    a1 = acts()
    a2 = acts() if recompute_activations else a1
    outs = []
    loss = 0.0
    for d, a in ((deg, a1), (0, a2)):
        st = GaussianRasterizationSettings(H, W, tfx, tfy, bg, 1.0, vm, pm, d, cp, False, False)
        m2 = torch.zeros_like(raw["means3D"], requires_grad=True) + 0
        out = GaussianRasterizer(st)(means3D=raw["means3D"], means2D=m2, shs=raw["shs"], opacities=a["opacities"],
                                     scales=a["scales"], rotations=a["rotations"], uvs=raw["uvs"], gradient_uvs=juv,
                                     texture=raw["texture"], extra_attrs=None)
        outs.append([o.detach().clone() for o in out[:5]])
        loss = loss + (2.0 if d == 0 else 1.0) * synth.synthetic_loss(out[0], out[3], out[2], target, nhat)
    loss.backward()
    return outs, {k: v.grad.clone() for k, v in raw.items()}


@pytest.mark.parametrize("recompute", [False, True])
def test_second_forward_shares_geometry_bit_identically(lib_built, recompute):
    """The reference renders every training view twice per iteration; the second render (sh_degree 0) finds K1's geometry
    fingerprint unchanged and re-uses the first one's tile lists / survivor lists / footprint counts (K1 + K6 only).  Images,
    radii and every gradient of the two-forwards-one-backward iteration are BIT-IDENTICAL to the same iteration with the cache
    off -- also when the activations are recomputed for the second render (new tensors, same values: the reference's getters)."""
    from texgs import rasterizer as RZ
    import math
    dev = torch.device("cuda:0")
    scene = synth.make_scene(6000, 128, seed=77, scale_mean=0.02, random_jacobian=True)
    cam = synth.fibonacci_cameras(6, 320, 240)[4]
    cam_t = (240, 320, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), torch.tensor([0.1, 0.2, 0.3], device=dev),
             cam.world_view_transform.to(dev), cam.full_proj_transform.to(dev), cam.camera_center.to(dev))
    target, nhat = synth.make_targets(240, 320, seed=2)
    target, nhat = target.to(dev), nhat.to(dev)
    saved = RZ.GEOM_CACHE
    try:
        RZ.GEOM_CACHE = False
        RZ.release_scratch()
        outs0, g0 = _reference_iteration(dev, scene, cam_t, 3, target, nhat, recompute)
        # run-to-run floor of the fp32 atomics: the largest of four repeats (heavy-tailed where terms cancel: rotations 1e-6 .. 2e-5)
        g0b = [_reference_iteration(dev, scene, cam_t, 3, target, nhat, recompute)[1] for _ in range(4)]
        RZ.GEOM_CACHE = True
        RZ.release_scratch()
        before = RZ.geometry_cache_stats()
        outs1, g1 = _reference_iteration(dev, scene, cam_t, 3, target, nhat, recompute)
        after = RZ.geometry_cache_stats()
    finally:
        RZ.GEOM_CACHE = saved
        RZ.release_scratch()
    assert after["hits"] - before["hits"] == 1 and after["misses"] - before["misses"] == 1
    for a, b in zip(outs0, outs1):
        for x, y in zip(a, b):
            assert torch.equal(x, y)
    assert not torch.equal(outs1[0][0], outs1[1][0])               # (the two renders do differ: sh_degree 3 vs 0)
    for k in g0:
        # gradients go through fp32 atomics (accumulator rows, border footprints): order-dependent last bits, amplified where terms
        # cancel (rotations).  The shared-geometry run must sit at the run-to-run floor of the cache-off iteration itself.
        floor, r = max(Hh.rel_err(gb[k], g0[k]) for gb in g0b), Hh.rel_err(g1[k], g0[k])
        Hh.report(f"shared_geometry/recompute{int(recompute)}/{k}", rel_l2_shared_vs_separate=r, rel_l2_run_to_run=floor)
        assert r <= 5.0 * floor + (2e-5 if k == "rotations" else 1e-6), (k, r, floor)


def test_changed_geometry_is_not_shared(lib_built):
    """Same camera tensors, one Gaussian moved by one ulp-sized step / a different opacity: the fingerprint differs, lists are rebuilt."""
    from texgs import rasterizer as RZ
    from texgs.rasterizer import forward_raw
    dev, scene, st, _, _ = _setup(seed=3)
    t = lambda x: x.to(dev)
    args = [t(scene.means3D), t(scene.shs), t(scene.opacities), t(scene.scales), t(scene.rotations), t(scene.uvs),
            t(scene.gradient_uvs), t(scene.texture)]
    RZ.release_scratch()
    s0 = RZ.geometry_cache_stats()
    o1, _ = forward_raw(st, *args)
    o2, st2 = forward_raw(st, *args)
    assert st2.shared_geometry and torch.equal(o1[0], o2[0])
    args[2] = args[2].clone(); args[2][17] *= 0.5                  # one opacity
    o3, st3 = forward_raw(st, *args)
    assert not st3.shared_geometry
    args[0] = args[0].clone(); args[0][5, 0] += 1e-3               # one centre
    o4, st4 = forward_raw(st, *args)
    assert not st4.shared_geometry
    s1 = RZ.geometry_cache_stats()
    assert s1["hits"] - s0["hits"] == 1 and s1["misses"] - s0["misses"] == 3
    RZ.GEOM_CACHE = False
    try:
        o5, st5 = forward_raw(st, *args)
    finally:
        RZ.GEOM_CACHE = True
    assert torch.equal(o4[0], o5[0]) and torch.equal(o4[4], o5[4])
    RZ.release_scratch()


def test_forward_only_callers_stop_paying_for_the_handoff(lib_built):
    """retexture.py:27 / visual_step render with parameters that require grad and never call backward: after two such graphs were
    dropped, the forward leaves K6's hand-off (survivor lists, footprint counts) to the backward; a backward that does arrive
    then builds it itself, the gradients are those of an ordinary forward + backward, and the forwards switch back."""
    from texgs import rasterizer as RZ
    from texgs.rasterizer import GaussianRasterizer
    dev, scene, st, target, nhat = _setup(seed=21)
    juv = scene.gradient_uvs.to(dev)

    def fwd(leaves):
        return GaussianRasterizer(st)(means3D=leaves["means3D"], means2D=None, shs=leaves["shs"], opacities=leaves["opacities"],
                                      scales=leaves["scales"], rotations=leaves["rotations"], uvs=leaves["uvs"],
                                      gradient_uvs=juv, texture=leaves["texture"], extra_attrs=None)

    def run_with_backward():
        leaves = {n: getattr(scene, n).clone().to(dev).requires_grad_(True) for n in NAMES}
        out = fwd(leaves)
        synth.synthetic_loss(out[0], out[3], out[2], target, nhat).backward()
        return [o.detach().clone() for o in out[:4]], {n: leaves[n].grad.clone() for n in NAMES}
    saved = RZ.GEOM_CACHE
    RZ.GEOM_CACHE = False                  # (every call here is the same view: keep the two mechanisms apart)
    try:
        import gc
        gc.collect()
        RZ.reset_handoff_predictor()
        ref_out, ref_g = run_with_backward()
        leaves = {n: getattr(scene, n).clone().to(dev).requires_grad_(True) for n in NAMES}
        for _ in range(3):                 # three graphs built and dropped
            out = fwd(leaves)
            del out
        assert RZ.unused_streak(dev.index) >= 2
        h0 = RZ.geometry_cache_stats()["late_handoffs"]
        got_out, got_g = run_with_backward()           # forward in lazy mode, backward builds the hand-off
        assert RZ.geometry_cache_stats()["late_handoffs"] == h0 + 1
        assert RZ.unused_streak(dev.index) == 0
        for a, b in zip(ref_out, got_out):
            assert torch.equal(a, b)
        for n in NAMES:
            assert Hh.rel_err(got_g[n], ref_g[n]) < 2e-5, n
        _, again_g = run_with_backward()               # back to the ordinary path
        assert RZ.geometry_cache_stats()["late_handoffs"] == h0 + 1
        for n in NAMES:
            assert Hh.rel_err(again_g[n], ref_g[n]) < 2e-5, n
    finally:
        RZ.GEOM_CACHE = saved
        RZ.reset_handoff_predictor()
        RZ.release_scratch()


def test_train_eval_interleave_costs_one_late_handoff_per_switch_and_holds_no_more_memory(lib_built):
    """A training loop with a periodic visual_step (models/texture_gaussian3d.py:499-511: two renders with parameters that require
    grad, no backward): ten differentiated iterations over changing views, two dropped forwards, three times over.  Each return
    to training pays at most ONE late hand-off (the first backward after the drops builds K6's lists itself, then the forwards
    switch back), and the module's caches -- backward scratch and the shared-geometry entry -- do not grow from cycle to cycle."""
    import gc
    from texgs import rasterizer as RZ
    from texgs.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
    dev = torch.device("cuda:0")
    scene = synth.make_scene(3000, 64, seed=5, scale_mean=0.03)
    cams = synth.fibonacci_cameras(8, 192, 144)
    sts = [Hh.settings_for(c, 2, torch.zeros(3), device=dev, cls=GaussianRasterizationSettings) for c in cams]
    target, nhat = synth.make_targets(144, 192, seed=3)
    target, nhat = target.to(dev), nhat.to(dev)
    juv = scene.gradient_uvs.to(dev)
    leaves = {n: getattr(scene, n).clone().to(dev).requires_grad_(True) for n in NAMES}

    def fwd(v):
        return GaussianRasterizer(sts[v])(means3D=leaves["means3D"], means2D=None, shs=leaves["shs"], opacities=leaves["opacities"],
                                          scales=leaves["scales"], rotations=leaves["rotations"], uvs=leaves["uvs"],
                                          gradient_uvs=juv, texture=leaves["texture"], extra_attrs=None)
    gc.collect()
    RZ.reset_handoff_predictor()
    RZ.release_scratch()
    held, late = [], []
    try:
        h0 = RZ.geometry_cache_stats()["late_handoffs"]
        for cycle in range(3):
            for it in range(10):
                out = fwd((cycle * 10 + it) % 8)
                synth.synthetic_loss(out[0], out[3], out[2], target, nhat).backward()
                for t in leaves.values():
                    t.grad = None
                del out
            for k in range(2):                      # visual_step: graphs built and dropped
                out = fwd(k)
                del out
            torch.cuda.synchronize()
            gc.collect()
            held.append(RZ.scratch_bytes())
            late.append(RZ.geometry_cache_stats()["late_handoffs"] - h0)
        assert late[-1] <= 2, late                  # one per return to training (cycles 2 and 3), none inside a training run
        assert held[2] <= held[1] * 1.001 + 4096, held
        assert len(RZ._GEOM) <= 1 and len(RZ._SCRATCH) <= 1
    finally:
        RZ.reset_handoff_predictor()
        RZ.release_scratch()
