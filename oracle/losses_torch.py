"""TEST INFRASTRUCTURE ONLY -- torch (float64) restatement of the reference's l1_loss (losses/pixelwise_loss.py) and
ssim_loss (losses/ssim_loss.py:6-54: 11x11 Gaussian window sigma 1.5, zero padding, per-channel, C1=0.01^2, C2=0.03^2)
and of the always-on loss terms of models/texture_gaussian3d.py:333-345.  PINNED: tests/golden/loss_frontend.npz holds
values and autograd gradients produced by the reference's own functions (tests/golden/make_golden.py)."""
import math

import torch
import torch.nn.functional as F


def window(size=11, sigma=1.5, dtype=torch.float64):
    g = torch.tensor([math.exp(-(x - size // 2) ** 2 / (2.0 * sigma ** 2)) for x in range(size)], dtype=dtype)
    g = g / g.sum()
    return (g[:, None] @ g[None, :])


def ssim(img1, img2):
    C = img1.shape[0]
    w = window(dtype=img1.dtype)[None, None].expand(C, 1, 11, 11).contiguous().to(img1.device)
    x, y = img1[None], img2[None]
    conv = lambda t: F.conv2d(t, w, padding=5, groups=C)
    mu1, mu2 = conv(x), conv(y)
    s1 = conv(x * x) - mu1 * mu1
    s2 = conv(y * y) - mu2 * mu2
    s12 = conv(x * y) - mu1 * mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    return (((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1 * mu1 + mu2 * mu2 + C1) * (s1 + s2 + C2))).mean()


def rgb_alpha_loss(image, gt_image, alpha, gt_alpha, lam, la):
    loss = (1.0 - lam) * (image - gt_image).abs().mean() + lam * (1.0 - ssim(image, gt_image))
    if alpha is not None:
        loss = loss + la * (alpha - gt_alpha).abs().mean()
    return loss


def norm_loss(pred, gt, mask=None):
    """losses/norm_reg_loss.py:66-71."""
    if mask is None:
        return torch.mean(1.0 - torch.sum(pred * gt, dim=0))
    return torch.sum((1.0 - torch.sum(pred * gt, dim=0, keepdim=True)) * mask) / (mask.sum() + 1e-6)


def smooth_loss(rgb, value, mask, gamma=0.1):
    """losses/smooth_loss.py:4-27: bilateral first-order smoothness over right / down / down-right / anti-diagonal
    neighbours, weight exp(-|d rgb|_1 / gamma) * mask_a * mask_b, each direction normalised by its weight sum."""
    pairs = [(lambda t: t[:, :, :-1], lambda t: t[:, :, 1:]), (lambda t: t[:, :-1, :], lambda t: t[:, 1:, :]),
             (lambda t: t[:, :-1, :-1], lambda t: t[:, 1:, 1:]), (lambda t: t[:, 1:, :-1], lambda t: t[:, :-1, 1:])]
    total = 0.0
    for a, b in pairs:
        w = torch.exp(-(a(rgb) - b(rgb)).abs().sum(0, keepdim=True) / gamma) * a(mask) * b(mask)
        total = total + (w * (a(value) - b(value))).abs().sum() / (w.sum() + 1e-6)
    return total / 4


def geom_losses(norm, gt_norm, gt_image, mask, depth, gt_depth, ln, ls, ld, gamma=0.1):
    loss = ln * norm_loss(norm, gt_norm, mask) + ls * smooth_loss(gt_image, norm, mask, gamma)
    if depth is not None:
        loss = loss + ld * (depth - gt_depth).abs().mean()
    return loss
