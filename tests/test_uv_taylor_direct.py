"""-m "not gpu": the UV Taylor step of the oracle (and therefore of the kernels, which are compared with it) is checked
against the DIRECT statement of the algorithm -- no shared algebra.

Every restatement in this repository (oracle/texgs_torch.py, oracle/texgs_ref.c, csrc/preprocess.hip + render.hip)
evaluates the per-(pixel, Gaussian) texture coordinate in the pre-folded form  uv(p) = phi + G dp / (1 + g.dp),
dp = pixel - xy  (DESIGN.md section 3.4).  An algebra slip in that fold would be common to all of them.  This file
computes the same quantity literally as SURVEY.md Appendix A.4 / A.5.1 (Texture-GS paper, Sec. 3) state it,

    ray through the pixel centre:   o = campos,  d = R_c2w [ndc_x tanfovx, ndc_y tanfovy, 1]
    intersection with the splat plane (mu, n):   x = o + d (n.(mu - o)) / (n.d)
    first-order Taylor expansion of the UV map:  uv = phi(mu) + J_phi (x - mu),   J[3i+j] = d uv_i / d x_j
                                                 (layout of models/texture_gaussian3d.py:216-227)

in float64 with plain torch ops, and compares VALUES and AUTOGRAD GRADIENTS (w.r.t. mu, rotation, phi, and the
constant J as a layout probe) with `oracle.preprocess`'s G, g, phi, xy.

The only non-algebraic difference between the two is the lineage's `1 / (w + 1e-7)` in the homogeneous divide that
defines xy: the fold is centred on that slightly shifted xy.  It is accounted for exactly (the fold at pixel p equals
the direct formula at p - xy_oracle + xy_exact, |shift| < 1e-4 px is asserted), so the value tolerance is 1e-9.
"""
import math

import torch

from texgs import synth
from oracle import texgs_torch as O
import helpers as Hh

D = torch.float64


def _direct_uv(means, rots, scales, phi, J9, st, pix):
    """uv[N, M, 3] for pixel positions pix[N, M, 2] (float pixel coordinates), literal ray/plane/Taylor form."""
    H, W = int(st.image_height), int(st.image_width)
    V = st.viewmatrix.to(D)
    cam = st.campos.to(D)
    Rc2w = V[:3, :3]                               # row-vector view matrix: its 3x3 block maps view -> world columns
    ndc = torch.stack([(2.0 * pix[..., 0] + 1.0) / W - 1.0, (2.0 * pix[..., 1] + 1.0) / H - 1.0], -1)
    dview = torch.stack([ndc[..., 0] * st.tanfovx, ndc[..., 1] * st.tanfovy, torch.ones_like(ndc[..., 0])], -1)
    d = dview @ Rc2w.t()                           # world-space ray direction (not normalised; the formula is scale-free)
    R = O.build_rotation(rots)
    kmin = torch.argmin(scales, dim=1)
    n = R[torch.arange(R.shape[0]), :, kmin]       # shortest axis; the sign cancels in x
    s = (n * (means - cam[None])).sum(-1)          # n.(mu - o)
    q = (n[:, None, :] * d).sum(-1)                # n.d
    x = cam[None, None, :] + d * (s[:, None] / q)[..., None]
    Jm = J9.reshape(-1, 3, 3)
    return phi[:, None, :] + torch.einsum("nij,nmj->nmi", Jm, x - means[:, None, :])


def _case(random_jacobian, seed):
    N = 400
    scene = synth.make_scene(N, 8, seed=seed, scale_mean=0.03, random_jacobian=random_jacobian)
    cam = synth.fibonacci_cameras(4, 200, 136)[seed % 4]
    # a float64-CONSISTENT camera: the float32 tensors of synth are each rounded separately (projmatrix != viewmatrix @ P
    # campos != -R^T T and R R^T != I beyond 1e-7), which would mask an algebra error of that size
    V = cam.world_view_transform.to(D).clone()
    U, _, Vh = torch.linalg.svd(V[:3, :3])
    V[:3, :3] = U @ Vh                                                     # exactly orthonormal in float64
    proj = synth.projection(0.01, 100.0, cam.FoVx, cam.FoVy).to(D).t()
    proj[0, 0] = 1.0 / math.tan(cam.FoVx * 0.5)
    proj[1, 1] = 1.0 / math.tan(cam.FoVy * 0.5)
    cam = cam._replace(world_view_transform=V, full_proj_transform=V @ proj, camera_center=torch.linalg.inv(V)[3, :3])
    st = Hh.settings_for(cam, 0, torch.zeros(3))
    leaves = dict(means=scene.means3D.to(D).requires_grad_(True), rots=scene.rotations.to(D).requires_grad_(True),
                  phi=scene.uvs.to(D).requires_grad_(True), J=scene.gradient_uvs.to(D).requires_grad_(True))
    scales = scene.scales.to(D)
    pre = O.preprocess(leaves["means"], None, None, scene.opacities.to(D), scales, leaves["rots"], leaves["phi"],
                       leaves["J"], st, D)
    g = torch.Generator().manual_seed(seed)
    M = 6
    dp = (torch.rand(N, M, 2, generator=g, dtype=D) - 0.5) * 24.0          # pixels within +-12 px of the centre
    pix = (pre["xy"].detach()[:, None, :] + dp).round()                      # integer pixel centres, as the blend uses
    return scene, st, leaves, scales, pre, pix


def _fold_uv(pre, pix):
    dpx = pix - pre["xy"][:, None, :]
    den = 1.0 + (pre["g"][:, None, :] * dpx).sum(-1)
    num = torch.einsum("nic,nmc->nmi", pre["G"], dpx)
    return pre["phi"][:, None, :] + num / den[..., None], den


def _exact_xy(means, st):
    H, W = int(st.image_height), int(st.image_width)
    hom = torch.cat([means, torch.ones(means.shape[0], 1, dtype=D)], 1) @ st.projmatrix.to(D)
    ndc = hom[:, :2] / hom[:, 3:4]
    return torch.stack([((ndc[:, 0] + 1.0) * W - 1.0) * 0.5, ((ndc[:, 1] + 1.0) * H - 1.0) * 0.5], 1)


def test_prefolded_uv_equals_direct_ray_plane_taylor_values_and_grads():
    worst_v, worst_g = 0.0, 0.0
    for random_jacobian, seed in [(False, 3), (True, 4), (True, 5)]:
        scene, st, lv, scales, pre, pix = _case(random_jacobian, seed)
        uv_fold, den = _fold_uv(pre, pix)
        # the fold is centred on the oracle's xy (lineage 1e-7 in the homogeneous divide): same point in exact coordinates
        shift = (pre["xy"] - _exact_xy(lv["means"], st)).detach()
        assert float(shift.abs().max()) < 1e-4
        uv_dir = _direct_uv(lv["means"], lv["rots"], scales, lv["phi"], lv["J"], st, pix - shift[:, None, :])
        keep = pre["valid"][:, None] & (den.detach() >= O.DEN_MIN) & (pre["g"].detach().abs().sum(-1) > 0)[:, None]
        assert int(keep.sum()) > 1000                              # outside the two guards
        err = ((uv_fold - uv_dir).abs().amax(-1))[keep]
        worst_v = max(worst_v, float(err.max().detach()))
        assert float(err.max().detach()) < 1e-9, float(err.max().detach())
        # gradients of a random linear functional of uv, by autograd on both sides
        gen = torch.Generator().manual_seed(100 + seed)
        c = torch.randn(uv_fold.shape, generator=gen, dtype=D) * keep[..., None]
        ga = torch.autograd.grad((uv_fold * c).sum(), list(lv.values()), retain_graph=True)
        gb = torch.autograd.grad((uv_dir * c).sum(), list(lv.values()))
        for name, a, b in zip(lv.keys(), ga, gb):
            rel = float((a - b).norm() / b.norm().clamp_min(1e-30))
            worst_g = max(worst_g, rel)
            # the shift is detached (its mu-derivative is O(1e-7) relative): that bounds the agreement
            assert rel < 2e-6, (name, random_jacobian, rel)
    print(f"uv fold vs direct formula: worst |uv| error {worst_v:.2e}, worst relative gradient error {worst_g:.2e}")


def test_jacobian_layout_is_row_major_duv_i_dx_j():
    """A transposed J must NOT reproduce the direct formula when J is not symmetric (and must when it is)."""
    for random_jacobian, expect_equal in [(False, True), (True, False)]:
        scene, st, lv, scales, pre, pix = _case(random_jacobian, 9)
        Jt = lv["J"].detach().reshape(-1, 3, 3).transpose(1, 2).reshape(-1, 9)
        uv_fold, den = _fold_uv(pre, pix)
        shift = (pre["xy"] - _exact_xy(lv["means"], st)).detach()
        uv_t = _direct_uv(lv["means"], lv["rots"], scales, lv["phi"], Jt, st, pix - shift[:, None, :])
        keep = pre["valid"][:, None] & (den.detach() >= O.DEN_MIN) & (pre["g"].detach().abs().sum(-1) > 0)[:, None]
        err = float(((uv_fold - uv_t).detach().abs().amax(-1))[keep].max())
        assert (err < 1e-9) == expect_equal, (random_jacobian, err)


def test_guards_fall_back_to_phi():
    """|n.t| <= PLANE_EPS |t| -> G = g = 0 (uv = phi for every pixel); den < DEN_MIN -> uv = phi for that pixel."""
    scene, st, lv, scales, pre, pix = _case(True, 11)
    t = (torch.cat([lv["means"], torch.ones(lv["means"].shape[0], 1, dtype=D)], 1) @ st.viewmatrix.to(D))[:, :3]
    R = O.build_rotation(lv["rots"])
    n = R[torch.arange(R.shape[0]), :, torch.argmin(scales, dim=1)]
    nv = n @ st.viewmatrix.to(D)[:3, :3]
    degen = ((nv * t).sum(1).abs() <= O.PLANE_EPS * t.norm(dim=1)).detach()
    assert bool((pre["G"][degen] == 0).all()) and bool((pre["g"][degen] == 0).all())
    assert bool((pre["g"][~degen & pre["valid"]].abs().sum(-1) > 0).all())
