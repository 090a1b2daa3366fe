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
    assert r < 1e-6
    for n in geo:
        if n == "texture":
            continue
        r = Hh.rel_err(geo[n], full[n])
        Hh.report(f"gating/frozen_texture_vs_full/{n}", rel_l2=r)
        assert r < 2e-6, (n, r)


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
