"""Fused loss front-end.  PINNED parity: tests/golden/loss_frontend.npz was produced by the reference's
own losses/ functions and their autograd.  CPU: the torch restatement vs the golden.  GPU: the HIP kernels vs the golden
(values within 2e-6, gradients within 1e-7 absolute ~ 1e-3 of their scale) and vs the restatement at 800x800."""
import os

import numpy as np
import pytest
import torch

from oracle import losses_torch as LO

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "loss_frontend.npz")


def _load(tag):
    d = np.load(G)
    t = lambda k: torch.tensor(d[f"{tag}_{k}"])
    return d, t("img"), t("gt"), t("alpha"), t("gta")


@pytest.mark.parametrize("tag", ["s", "m"])
def test_restatement_matches_reference_losses(tag):
    d, img, gt, alpha, gta = _load(tag)
    i64 = img.double().requires_grad_(True)
    a64 = alpha.double().requires_grad_(True)
    loss = LO.rgb_alpha_loss(i64, gt.double(), a64, gta.double(), float(d["lam"]), float(d["la"]))
    loss.backward()
    assert abs(float(loss) - float(d[f"{tag}_loss"])) < 2e-6
    assert abs(float(LO.ssim(img.double(), gt.double())) - float(d[f"{tag}_ssim"])) < 2e-6
    assert np.allclose(i64.grad.numpy(), d[f"{tag}_dimg"], atol=2e-8, rtol=1e-4)
    assert np.allclose(a64.grad.numpy(), d[f"{tag}_dalpha"], atol=1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["s", "m"])
def test_hip_loss_matches_reference_golden(lib_built, tag):
    from texgs.losses import rgb_alpha_loss
    d, img, gt, alpha, gta = _load(tag)
    dev = torch.device("cuda:0")
    i = img.to(dev).requires_grad_(True)
    a = alpha.to(dev).requires_grad_(True)
    loss = rgb_alpha_loss(i, gt.to(dev), a, gta.to(dev), float(d["lam"]), float(d["la"]))
    loss.backward()
    assert abs(float(loss) - float(d[f"{tag}_loss"])) < 3e-6
    gi = d[f"{tag}_dimg"]
    assert float(np.abs(i.grad.cpu().numpy() - gi).max()) < 1e-3 * float(np.abs(gi).max())
    assert np.allclose(a.grad.cpu().numpy(), d[f"{tag}_dalpha"], atol=1e-9)


@pytest.mark.gpu
def test_hip_loss_full_size_vs_restatement(lib_built):
    from texgs.losses import rgb_alpha_loss
    g = torch.Generator().manual_seed(5)
    H = W = 800
    img = torch.rand(3, H, W, generator=g)
    gt = torch.rand(3, H, W, generator=g)
    dev = torch.device("cuda:0")
    i = img.to(dev).requires_grad_(True)
    loss = rgb_alpha_loss(i, gt.to(dev), None, None, 0.2, 0.0)
    (2.0 * loss).backward()
    i64 = img.double().requires_grad_(True)
    ref = LO.rgb_alpha_loss(i64, gt.double(), None, None, 0.2, 0.0)
    (2.0 * ref).backward()
    assert abs(float(loss) - float(ref)) < 5e-6
    assert float((i.grad.cpu().double() - i64.grad).abs().max()) < 1e-3 * float(i64.grad.abs().max())


# ---- geometric regularisers (normal loss, bilateral normal smoothness, depth L1): models/texture_gaussian3d.py:347-368
GG = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "geom_losses.npz")


def _gload(tag):
    d = np.load(GG)
    t = lambda k: torch.tensor(d[f"{tag}_{k}"])
    return d, t("norm"), t("gtn"), t("gti"), t("mask"), t("depth"), t("gtd")


@pytest.mark.parametrize("tag", ["s", "m"])
def test_geom_restatement_matches_reference_losses(tag):
    d, norm, gtn, gti, mask, depth, gtd = _gload(tag)
    n64 = norm.double().requires_grad_(True)
    d64 = depth.double().requires_grad_(True)
    loss = LO.geom_losses(n64, gtn.double(), gti.double(), mask.double(), d64, gtd.double(), float(d["ln"]), float(d["ls"]),
                          float(d["ld"]), float(d["gamma"]))
    loss.backward()
    assert abs(float(loss) - float(d[f"{tag}_loss"])) < 2e-6
    assert np.allclose(n64.grad.numpy(), d[f"{tag}_dnorm"], atol=1e-8, rtol=2e-4)
    assert np.allclose(d64.grad.numpy(), d[f"{tag}_ddepth"], atol=1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["s", "m"])
def test_hip_geom_losses_match_reference_golden(lib_built, tag):
    from texgs.losses import geom_losses
    d, norm, gtn, gti, mask, depth, gtd = _gload(tag)
    dev = torch.device("cuda:0")
    n = norm.to(dev).requires_grad_(True)
    z = depth.to(dev).requires_grad_(True)
    loss = geom_losses(n, gtn.to(dev), gti.to(dev), mask.to(dev), z, gtd.to(dev), float(d["ln"]), float(d["ls"]),
                       float(d["gamma"]), float(d["ld"]))
    loss.backward()
    assert abs(float(loss) - float(d[f"{tag}_loss"])) < 3e-6
    gn = d[f"{tag}_dnorm"]
    assert float(np.abs(n.grad.cpu().numpy() - gn).max()) < 1e-3 * float(np.abs(gn).max())
    assert np.allclose(z.grad.cpu().numpy(), d[f"{tag}_ddepth"], atol=1e-9)
    # each term alone (a zero lambda skips the term and its pointers)
    only = geom_losses(norm=n.detach(), gt_image=gti.to(dev), mask=mask.to(dev), lambda_smooth=1.0, gamma=float(d["gamma"]))
    assert abs(float(only) - float(d[f"{tag}_Lnsm"])) < 3e-6
    only = geom_losses(norm=n.detach(), gt_norm=gtn.to(dev), mask=mask.to(dev), lambda_norm=1.0)
    assert abs(float(only) - float(d[f"{tag}_Lnorm"])) < 3e-6


@pytest.mark.gpu
def test_hip_geom_losses_full_size_into_rasterizer_backward(lib_built):
    """800x800: HIP vs the float64 restatement, and the produced dL/dnorm flows into the rasterizer's backward."""
    from texgs.losses import geom_losses
    g = torch.Generator().manual_seed(6)
    H = W = 800
    norm = torch.randn(3, H, W, generator=g) * 0.5
    gtn = torch.nn.functional.normalize(torch.randn(3, H, W, generator=g), dim=0)
    gti = torch.rand(3, H, W, generator=g)
    mask = (torch.rand(1, H, W, generator=g) > 0.2).float()
    dev = torch.device("cuda:0")
    n = norm.to(dev).requires_grad_(True)
    loss = geom_losses(norm=n, gt_norm=gtn.to(dev), gt_image=gti.to(dev), mask=mask.to(dev), lambda_norm=0.1, lambda_smooth=0.5)
    loss.backward()
    n64 = norm.double().requires_grad_(True)
    ref = LO.geom_losses(n64, gtn.double(), gti.double(), mask.double(), None, None, 0.1, 0.5, 0.0)
    ref.backward()
    assert abs(float(loss) - float(ref)) < 5e-6
    assert float((n.grad.cpu().double() - n64.grad).abs().max()) < 1e-3 * float(n64.grad.abs().max())
