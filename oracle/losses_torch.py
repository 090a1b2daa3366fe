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


def norm_from_depth(depth, world_view_transform, tanfovx, tanfovy, threshold=1e-2):
    """losses/norm_reg_loss.py:16-63 restated with slices instead of conv2d filters: back-projection of every pixel with
    ndc = (2 p + 1) / S - 1, one-sided differences with replicate border, normal = normalise(cross(grad_y, grad_x), eps 1e-6),
    mask = all four one-sided differences shorter than `threshold`.  Works in the dtype of `depth`."""
    _, H, W = depth.shape
    dt = depth.dtype
    px = torch.arange(W, dtype=dt).reshape(1, 1, W).expand(1, H, W)
    py = torch.arange(H, dtype=dt).reshape(1, H, 1).expand(1, H, W)
    nx, ny = (2.0 * px + 1.0) / W - 1.0, (2.0 * py + 1.0) / H - 1.0
    cc = torch.cat([nx * tanfovx * depth, ny * tanfovy * depth, depth, torch.ones_like(depth)], dim=0)
    c2w = torch.linalg.inv(world_view_transform.to(dt).transpose(0, 1))
    xyz = (c2w @ cc.reshape(4, H * W)).reshape(4, H, W)[:3]
    pad = F.pad(xyz.unsqueeze(0), (1, 1, 1, 1), mode="replicate").squeeze(0)
    c = pad[:, 1:-1, 1:-1]
    gl, gr = c - pad[:, 1:-1, :-2], pad[:, 1:-1, 2:] - c
    gu, gd = c - pad[:, :-2, 1:-1], pad[:, 2:, 1:-1] - c
    gx, gy = (gr + gl) / 2, (gd + gu) / 2
    mask = ((gl.norm(dim=0, keepdim=True) < threshold) & (gr.norm(dim=0, keepdim=True) < threshold)
            & (gu.norm(dim=0, keepdim=True) < threshold) & (gd.norm(dim=0, keepdim=True) < threshold))
    n = torch.cross(gy, gx, dim=0)
    return F.normalize(n, p=2, dim=0, eps=1e-6), mask.to(dt)


def norm_reg_loss(norm, depth, world_view_transform, tanfovx, tanfovy, gt_alpha, threshold=1e-2):
    """losses/norm_reg_loss.py:73-78."""
    norm2, mask = norm_from_depth(depth.detach(), world_view_transform, tanfovx, tanfovy, threshold)
    return norm_loss(norm, norm2, gt_alpha * mask)
