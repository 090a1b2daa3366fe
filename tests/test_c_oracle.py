"""-m "not gpu": the plain-C restatement (oracle/texgs_ref.c, fp32) against the torch float64 restatement.
Forward within fp32 rounding; the hand-written C backward against torch AUTOGRAD (independent derivation)."""
import numpy as np
import torch

from texgs import synth
from oracle import texgs_ref as CR
import helpers as Hh

CASES = [(600, 32, 96, 80, 0.05, 3, 1, (0.1, 0.2, 0.3)), (300, 16, 64, 48, 0.08, 1, 3, (0.0, 0.5, 0.0)),
         (500, 32, 96, 80, 0.05, 2, 2, (0.0, 0.1, 0.2), True)]       # last: non-symmetric UV Jacobians


def _scene(case):
    N, R, W, H, sm, deg, view, bg = case[:8]
    return synth.make_scene(N, R, seed=N + R, scale_mean=sm, random_jacobian=len(case) > 8 and case[8]), synth.fibonacci_cameras(4, W, H)[view], deg, torch.tensor(bg)


def test_c_forward_matches_torch_oracle():
    for case in CASES:
        scene, cam, deg, bg = _scene(case)
        ref, dbg, _ = Hh.oracle_run(scene, cam, deg, bg)
        run = CR.RefRun(scene, Hh.settings_for(cam, deg, bg))
        out = torch.tensor(run.forward()).double()
        amb = dbg["ambiguity"] < 1e-4
        full = torch.cat([ref[0], ref[1], ref[2], ref[3]], 0).double()
        err = (out - full).abs()
        assert float(err[:3][:, ~amb].max()) < 1e-4
        assert float(err[3:4][:, ~amb].max()) < 4e-4
        assert float(err[4:][:, ~amb].max()) < 1e-4
        assert np.array_equal(run.radii[:run.N].astype(np.int64), ref[4].numpy().astype(np.int64))
        assert run.D == dbg["binning"]["D"]
        assert np.array_equal(run.ranges.astype(np.int64), dbg["binning"]["ranges"].numpy())
        assert float((torch.tensor(run.n_contrib.astype(np.int64)) == dbg["n_contrib"]).float().mean()) > 0.995


def test_c_backward_matches_torch_autograd():
    for case in CASES:
        scene, cam, deg, bg = _scene(case)
        target, nhat = synth.make_targets(cam.image_height, cam.image_width, seed=5)
        ref, dbg, gref = Hh.oracle_run(scene, cam, deg, bg, with_grad=True, target=target, nhat=nhat, depth_weight=0.05)
        run = CR.RefRun(scene, Hh.settings_for(cam, deg, bg))
        out = torch.tensor(run.forward(), requires_grad=True)
        L = synth.synthetic_loss(out[0:3], out[7:8], out[4:7], target, nhat) + 0.05 * out[3:4].mean()
        L.backward()
        g = run.backward(out.grad.numpy())
        for name, exp in gref.items():
            ok, msg = Hh.grad_close(torch.tensor(g[name]), exp)
            assert ok, (name, msg)


def test_c_oracle_sort_is_stable_and_sorted():
    scene, cam, deg, bg = _scene(CASES[0])
    run = CR.RefRun(scene, Hh.settings_for(cam, deg, bg))
    run.forward()
    ks = run.keys_sorted[:run.D]
    assert np.all(ks[1:] >= ks[:-1])
    same = ks[1:] == ks[:-1]
    assert np.all(run.point_list[1:run.D][same] > run.point_list[:run.D - 1][same])   # ties keep Gaussian-index order
    # multiset preserved
    assert np.array_equal(np.sort(run.keys_unsorted[:run.D]), ks)


def test_ambiguity_attribution_explains_a_differently_rounded_build():
    """The attribution the full-size GPU tests rely on (helpers.forward_attributed / grad_attributed on oracle/texgs_ref.c's
    texgs_ref_ambiguity), exercised on the CPU: the same blend loops built with another exp, another reciprocal and fused
    multiply-adds (oracle/Makefile: libtexgs_ref_variant.so) are a second fp32 implementation of the contract.  Every pixel
    and every gradient row the oracle does not flag must agree -- zero unexplained differences -- and the margin map must agree
    with the float64 oracle's own ambiguity map about which pixels are safe."""
    N, R, W, H = 40_000, 256, 320, 320
    scene = synth.make_scene(N, R, seed=3, scale_mean=0.012, random_jacobian=True)
    cam = synth.fibonacci_cameras(8, W, H)[5]
    run = CR.RefRun(scene, Hh.settings_for(cam, 3, torch.tensor([0.1, 0.0, 0.2])))
    run.forward()
    g = torch.Generator().manual_seed(7)
    dout = (torch.randn(8, H, W, generator=g) / (H * W)).numpy()
    gref = run.backward(dout)
    out_v, nc_v, g_v = run.variant_render(dout)
    assert not np.array_equal(out_v, run.out)                      # it IS a different rounding
    margin, gflag, tflag = run.ambiguity(tau_fwd=Hh.TAU_FWD, tau_cell=Hh.tau_cell(R), tau_relu=Hh.tau_relu(R))
    Hh.forward_attributed("cpu/c32_vs_variant/fwd", torch.tensor(out_v), run, margin, n_contrib=nc_v, amb_frac_max=2e-3)
    for name in ["means3D", "means2D", "shs", "opacities", "scales", "rotations", "uvs", "texture"]:
        Hh.grad_attributed(f"cpu/c32_vs_variant/bwd/{name}", torch.tensor(g_v[name]), torch.tensor(gref[name]),
                           tflag if name == "texture" else gflag)
    # pair level (VERDICT r4 #3b): EVERY row, tolerance widened by what its own near-edge pairs can contribute; only rows behind a
    # 1/255 / T-stop / clamp decision are excused
    mg, hard, _ = run.ambiguity(tau_fwd=Hh.TAU_FWD, tau_cell=0.0, tau_relu=0.0, own_only=True)
    run.backward(dout, tau_cell=Hh.tau_cell(R), cell_weight=1.0, margin=mg, tau_fwd=Hh.TAU_FWD, tau_relu=Hh.tau_relu(R))
    dev = run.cell_edge_deviation()
    for name in ["means3D", "means2D", "shs", "opacities", "scales", "rotations", "uvs"]:
        Hh.grad_mass_attributed(f"cpu/c32_vs_variant/bwd_pair_level/{name}", torch.tensor(g_v[name]), torch.tensor(gref[name]),
                                hard, dev[name])
    assert hard.mean() < 0.05
    # flags are not a blanket: most rows are unflagged
    assert gflag.mean() < 0.25 and tflag.mean() < 0.05
    # accumulation-sensitive rows (needle-shaped splats): few, and found deterministically
    sens = run.accumulation_sensitive()
    assert sens.mean() < 1e-3 and np.array_equal(sens, run.accumulation_sensitive())


def test_conditioning_mass_explains_the_stress_scene_between_two_fp32_builds():
    """VERDICT r5 #5 (second half): the stress scene (screen-filling splats: the falloff exponent is a difference of terms of size
    10..100, alpha is known to 1e-5..1e-4 relative in ANY fp32 implementation) had its gradients checked against row budgets.  The
    oracle's backward now books a CONDITIONING mass per pair (texgs_ref.c: the pair's terms under the rounding of its own exponent,
    of the transmittance in front of it and of the blend behind it) and pushes it through the last stage like the cell-edge masses.
    Exercised on the CPU against the differently-rounded build: EVERY row within plain tolerance + 1.5 x its own bound -- zero
    unexplained rows, no outlier budget --, only rows behind a decision inside its own rounding error may differ freely."""
    scene = Hh.stress_scene()
    cam = synth.look_at_camera((0.3, -0.2, -3.0), 640, 360, fovx=1.2)
    run = CR.RefRun(scene, Hh.settings_for(cam, 3, torch.tensor([0.05, 0.1, 0.15])))
    run.forward()
    R, H, W = scene.texture.shape[1], cam.image_height, cam.image_width
    g = torch.Generator().manual_seed(11)
    dout = (torch.randn(8, H, W, generator=g) / (H * W)).numpy()
    out_v, nc_v, g_v = run.variant_render(dout)
    assert not np.array_equal(out_v, run.out)
    Hh.stress_gradient_check("cpu/stress/c32_vs_variant", run, [torch.tensor(g_v[n]) for n in Hh.GRAD_NAMES], dout, R)
    assert float((run.cond > 1e-5).mean()) > 0.05          # the scene really is ill-conditioned (benchmark scenes: cond ~ 1e-7)


def test_band_limited_texture_needs_no_cell_edge_flags():
    """The premise of helpers.band_limited_parity, on the CPU: with synth.band_limited_texture a bilinear cell chosen differently
    moves a pair's dL/duv by ~1 %, so two differently-rounded fp32 builds agree on EVERY gradient row that no 1/255, T-stop or
    colour-clamp decision flags (tau_cell = 0), RGB within the literal 1e-4 at R = 2048; the flagged rows are a few per cent."""
    N, R, W, H = 40_000, 2048, 320, 320
    scene = synth.make_scene(N, 8, seed=3, scale_mean=0.012, random_jacobian=True)
    scene = scene._replace(texture=synth.band_limited_texture(R, seed=9, period=64, amplitude=0.5))
    cam = synth.fibonacci_cameras(8, W, H)[5]
    run = CR.RefRun(scene, Hh.settings_for(cam, 3, torch.tensor([0.1, 0.0, 0.2])))
    run.forward()
    g = torch.Generator().manual_seed(7)
    dout = (torch.randn(8, H, W, generator=g) / (H * W)).numpy()
    gref = run.backward(dout)
    out_v, nc_v, g_v = run.variant_render(dout)
    margin, gflag, tflag = run.ambiguity(tau_fwd=Hh.TAU_FWD, tau_cell=0.0, tau_relu=1e-5)
    Hh.forward_attributed("cpu/band_limited/c32_vs_variant/fwd", torch.tensor(out_v), run, margin, n_contrib=nc_v, amb_frac_max=2e-3,
                          rgb_tol=1e-4)
    _, hard, _ = run.ambiguity(tau_fwd=Hh.TAU_FWD, tau_cell=0.0, tau_relu=0.0, own_only=True)
    run.backward(dout, tau_cell=Hh.tau_cell(R), cell_weight=Hh.BAND_LIMITED_CELL_WEIGHT, margin=margin, tau_fwd=Hh.TAU_FWD, tau_relu=1e-5)
    dev = run.cell_edge_deviation()
    for name in ["means3D", "means2D", "shs", "opacities", "scales", "rotations", "uvs"]:
        r = Hh.grad_mass_attributed(f"cpu/band_limited/c32_vs_variant/bwd/{name}", torch.tensor(g_v[name]), torch.tensor(gref[name]),
                                    hard, dev[name])
        assert r["rows_with_tolerance_more_than_doubled_frac"] < 0.05, (name, r)
    Hh.grad_attributed("cpu/band_limited/c32_vs_variant/bwd/texture", torch.tensor(g_v["texture"]), torch.tensor(gref["texture"]), tflag,
                       flagged_frac_max=0.05)


def test_c_untextured_surface_matches_torch_oracle():
    """The untextured surface (`diff_gauss`, reference render/render.py:52-53,66-68,75-84) in the C restatement -- texture NULL,
    colour offset, view-dependent SH rows, cov3D_precomp with the smallest-eigenvector normal -- against the float64 torch
    restatement: forward to fp32 rounding, the hand-written backward (incl. dL/dcolor_offset, dL/dcov3D) against torch autograd."""
    base, cam, deg, bg = _scene((500, 16, 96, 80, 0.05, 2, 1, (0.1, 0.2, 0.3)))
    target, nhat = synth.make_targets(cam.image_height, cam.image_width, seed=6)
    for mode in ("shs", "precomp", "cov"):
        us, _ = Hh.untextured_from(base, mode, seed=11)
        d = deg if mode == "shs" else 0
        ref, dbg, gref = Hh.oracle_run_untextured(us, cam, d, bg, with_grad=True, target=target, nhat=nhat, depth_weight=0.05)
        run = CR.RefRun(us, Hh.settings_for(cam, d, bg))
        out = torch.tensor(run.forward(), requires_grad=True)
        amb = dbg["ambiguity"] < 1e-4
        full = torch.cat([ref[0], ref[1], ref[2], ref[3]], 0).double()
        err = (out.detach().double() - full).abs()
        assert float(err[:3][:, ~amb].max()) < 1e-4, mode
        assert float(err[3:4][:, ~amb].max()) < 4e-4, mode
        assert float(err[4:][:, ~amb].max()) < 1e-4, mode
        assert np.array_equal(run.radii[:run.N].astype(np.int64), ref[4].numpy().astype(np.int64)), mode
        assert run.D == dbg["binning"]["D"], mode
        L = synth.synthetic_loss(out[0:3], out[7:8], out[4:7], target, nhat) + 0.05 * out[3:4].mean()
        L.backward()
        g = run.backward(out.grad.numpy())
        assert g["texture"] is None and g["uvs"] is None
        for name, exp in gref.items():
            assert g[name] is not None, (mode, name)
            ok, msg = Hh.grad_close(torch.tensor(g[name]), exp)
            assert ok, (mode, name, msg)
        margin, gflag, _ = run.ambiguity()
        assert margin.shape == (cam.image_height, cam.image_width) and gflag.shape == (run.N,)
