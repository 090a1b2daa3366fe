"""-m "not gpu": the oracle and the host-side restatements against the golden vectors generated from the importable
pieces of the reference (tests/golden/make_golden.py).  The operator itself has no reference vector (parity
unpinned, see oracle/texgs_torch.py); op_small.npz pins the oracle against regressions of itself."""
import math
import os

import numpy as np
import torch

from texgs import synth
from oracle import texgs_torch as O
import helpers as Hh

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_camera_matrices_match_reference():
    d = np.load(os.path.join(G, "cameras.npz"))
    for i in range(4):
        R, T = d[f"R{i}"], d[f"T{i}"]
        fovx, fovy = d[f"fov{i}"]
        wvt = torch.tensor(synth.world2view(R, T)).transpose(0, 1)
        proj = synth.projection(0.01, 100.0, float(fovx), float(fovy)).transpose(0, 1)
        full = (wvt.unsqueeze(0).bmm(proj.unsqueeze(0))).squeeze(0)
        assert np.allclose(wvt.numpy(), d[f"wvt{i}"], atol=1e-6)
        assert np.allclose(proj.numpy(), d[f"proj{i}"], atol=1e-6)
        assert np.allclose(full.numpy(), d[f"full{i}"], atol=1e-5)
        assert np.allclose(wvt.inverse()[3, :3].numpy(), d[f"center{i}"], atol=1e-5)
        # convention the kernels rely on: clip w == view z (utils/graphics.py:68)
        p = torch.tensor([[0.3, -0.2, 0.5, 1.0]])
        assert abs(float((p @ full)[0, 3] - (p @ wvt)[0, 2])) < 1e-5


def test_look_at_camera_consistency():
    cam = synth.fibonacci_cameras(5, 80, 60)[2]
    o = cam.camera_center.double()
    hom = torch.cat([o, torch.ones(1, dtype=torch.float64)])
    v = hom @ cam.world_view_transform.double()
    assert float(v[:3].abs().max()) < 1e-5                      # camera centre maps to the view origin
    origin = torch.tensor([0.0, 0.0, 0.0, 1.0], dtype=torch.float64) @ cam.world_view_transform.double()
    assert abs(float(origin[0])) < 1e-5 and abs(float(origin[1])) < 1e-5 and float(origin[2]) > 3.0


def test_sh_matches_reference_eval_sh():
    d = np.load(os.path.join(G, "sh.npz"))
    dirs = torch.tensor(d["dirs"])
    coef = torch.tensor(d["coef"])                   # [n, 3, 16]
    shs_rest = coef[:, :, 1:].permute(0, 2, 1)       # operator layout [n, 15, 3], coefficient 1.. first
    for deg in range(4):
        want = torch.tensor(d[f"deg{deg}"])
        got = O.SH_C0 * coef[:, :, 0] + O.sh_view_dependent(deg, shs_rest, dirs)
        assert torch.allclose(got, want, atol=1e-12), deg
    assert abs(float(d["C0"]) - O.SH_C0) < 1e-15


def test_cubemap_convention_matches_cube_to_dir():
    d = np.load(os.path.join(G, "cube.npz"))
    dirs = torch.tensor(d["dirs"])                   # [6,R,R,3] at texel centres
    Rr = int(d["R"])
    tex = torch.arange(6 * Rr * Rr * 3, dtype=torch.float64).reshape(6, Rr, Rr, 3)
    got, margin = O.cubemap_fetch(dirs * 2.5, tex)   # scale-invariant
    assert torch.allclose(got, tex, atol=1e-9)
    # bilinear half-way between two texel centres of face 4 (+z), clamp-to-edge at the border
    u = torch.tensor([[(0.5 + 0.5) / Rr * 2 - 1 + 1.0 / Rr, -((0.5) / Rr * 2 - 1), 1.0]], dtype=torch.float64)
    got, _ = O.cubemap_fetch(u, tex)
    assert torch.allclose(got[0], 0.5 * (tex[4, 0, 0] + tex[4, 0, 1]) + 0.5 * (tex[4, 0, 1] - tex[4, 0, 0]), atol=1e-9)


def test_reference_losses_shape_the_upstream_grads():
    d = np.load(os.path.join(G, "losses.npz"))
    a, b = torch.tensor(d["a"]), torch.tensor(d["b"])
    assert abs(float((a - b).abs().mean()) - float(d["l1"])) < 1e-6      # losses/pixelwise_loss.py l1_loss
    assert 0.0 < float(d["ssim"]) < 1.0


def test_oracle_regression_op_small():
    d = np.load(os.path.join(G, "op_small.npz"))
    scene = synth.make_scene(200, 16, seed=11, scale_mean=0.06)
    cam = synth.fibonacci_cameras(4, 64, 48)[1]
    bg = torch.tensor([0.2, 0.1, 0.3])
    target, nhat = synth.make_targets(48, 64, seed=2)
    ref, dbg, grads = Hh.oracle_run(scene, cam, 2, bg, with_grad=True, target=target, nhat=nhat, depth_weight=0.05)
    for k, name in enumerate(["image", "depth", "norm", "alpha"]):
        assert np.allclose(ref[k].detach().numpy(), d[name], atol=1e-10), name
    assert np.array_equal(ref[4].numpy(), d["radii"])
    assert np.array_equal(dbg["n_contrib"].numpy(), d["n_contrib"])
    assert np.array_equal(dbg["binning"]["point_list"].numpy(), d["point_list"])
    assert np.array_equal(dbg["binning"]["ranges"].numpy(), d["ranges"])
    for k, v in grads.items():
        assert np.allclose(v.numpy(), d["grad_" + k], atol=1e-10, rtol=1e-8), k


def test_oracle_properties():
    """Domain properties: alpha = 1 - T_final, image = colour + T*bg (linearity in bg), culled Gaussians change
    nothing, gradient of a culled Gaussian is zero."""
    scene = synth.make_scene(150, 8, seed=4, scale_mean=0.07)
    cam = synth.fibonacci_cameras(3, 48, 32)[0]
    bg0, bg1 = torch.zeros(3), torch.tensor([0.5, 0.25, 1.0])
    r0, dbg0, _ = Hh.oracle_run(scene, cam, 1, bg0)
    r1, _, _ = Hh.oracle_run(scene, cam, 1, bg1)
    T = dbg0["final_T"]
    assert torch.allclose(r0[3][0], 1.0 - T, atol=1e-9)
    assert torch.allclose(r1[0] - r0[0], T[None] * bg1.double()[:, None, None], atol=1e-12)
    behind = scene._replace(means3D=torch.cat([scene.means3D, cam.camera_center[None] * 1.5]),
                            scales=torch.cat([scene.scales, scene.scales[:1]]),
                            rotations=torch.cat([scene.rotations, scene.rotations[:1]]),
                            opacities=torch.cat([scene.opacities, scene.opacities[:1]]),
                            shs=torch.cat([scene.shs, scene.shs[:1]]), uvs=torch.cat([scene.uvs, scene.uvs[:1]]),
                            gradient_uvs=torch.cat([scene.gradient_uvs, scene.gradient_uvs[:1]]))
    r2, _, _ = Hh.oracle_run(behind, cam, 1, bg0)
    assert torch.allclose(r2[0], r0[0]) and int(r2[4][-1]) == 0
