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
