"""Fused loss front-end: L1 + SSIM on the image and L1 on alpha, the always-on terms of
TextureGaussian3D.compute_loss (models/texture_gaussian3d.py:333-345), as two HBM-bound HIP kernels instead of the
reference's 5 depthwise conv2d + ~25 elementwise kernels and their autograd.  The gradient w.r.t. the rendered image /
alpha is computed in the same pass and handed to autograd, i.e. straight to the rasterizer's backward."""
import ctypes as C

import torch

from . import _lib


class _RgbAlphaLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image, gt_image, alpha, gt_alpha, lambda_dssim, lambda_alpha):
        lib = _lib.load()
        dev = image.device
        if dev.type != "cuda":
            raise RuntimeError("texgs.losses runs on an AMD GPU; there is no CPU fallback")
        _, H, W = image.shape
        img = image.detach().to(torch.float32).contiguous()
        gt = gt_image.detach().to(torch.float32).contiguous()
        if img.shape != gt.shape or img.shape[0] != 3:
            raise ValueError("image and gt_image must both be [3,H,W]")
        use_alpha = alpha is not None and gt_alpha is not None and lambda_alpha != 0.0
        a = alpha.detach().to(torch.float32).contiguous() if use_alpha else None
        ga = gt_alpha.detach().to(torch.float32).contiguous() if use_alpha else None
        scratch = torch.empty(9 * H * W, dtype=torch.float32, device=dev)
        sums = torch.empty(4, dtype=torch.float32, device=dev)
        d_img = torch.empty_like(img)
        d_a = torch.empty_like(a) if use_alpha else None
        p = lambda t: None if t is None else t.data_ptr()
        with torch.cuda.device(dev):
            _lib.check(lib.texgs_rgb_alpha_loss(p(img), p(gt), p(a), p(ga), H, W, float(lambda_dssim), float(lambda_alpha),
                                                p(scratch), p(sums), p(d_img), p(d_a),
                                                torch.cuda.current_stream(dev).cuda_stream), "texgs_rgb_alpha_loss")
        n = 3.0 * H * W
        l1 = sums[0] / n
        ssim = sums[1] / n
        loss = (1.0 - lambda_dssim) * l1 + lambda_dssim * (1.0 - ssim)
        la = None
        if use_alpha:
            la = sums[2] / (H * W)
            loss = loss + lambda_alpha * la
        ctx.save_for_backward(d_img, d_a if use_alpha else torch.empty(0, device=dev))
        ctx.use_alpha = use_alpha
        ctx.stats = dict(Ll1=l1, Lssim=1.0 - ssim, Lalpha=la)
        return loss

    @staticmethod
    def backward(ctx, g):
        d_img, d_a = ctx.saved_tensors
        return g * d_img, None, (g * d_a if ctx.use_alpha else None), None, None, None


def rgb_alpha_loss(image, gt_image, alpha=None, gt_alpha=None, lambda_dssim=0.2, lambda_alpha=0.0):
    """(1-l)*l1_loss(image, gt) + l*(1 - ssim_loss(image, gt)) [+ la*l1_loss(alpha, gt_alpha)] with the reference's
    definitions (losses/pixelwise_loss.py, losses/ssim_loss.py:16-54); differentiable w.r.t. image and alpha."""
    return _RgbAlphaLoss.apply(image, gt_image, alpha, gt_alpha, float(lambda_dssim), float(lambda_alpha))


class _GeomLosses(torch.autograd.Function):
    @staticmethod
    def forward(ctx, norm, gt_norm, gt_image, mask, depth, gt_depth, lambda_norm, lambda_smooth, gamma, lambda_depth):
        lib = _lib.load()
        ref = norm if norm is not None else depth
        dev = ref.device
        if dev.type != "cuda":
            raise RuntimeError("texgs.losses runs on an AMD GPU; there is no CPU fallback")
        H, W = ref.shape[-2:]
        c = lambda t: None if t is None else t.detach().to(torch.float32).contiguous()
        n, gn, gi, m, d, gd = c(norm), c(gt_norm), c(gt_image), c(mask), c(depth), c(gt_depth)
        use_n = n is not None and (lambda_norm != 0.0 or lambda_smooth != 0.0)
        use_d = d is not None and lambda_depth != 0.0
        sums = torch.empty(12, dtype=torch.float32, device=dev)
        d_n = torch.empty_like(n) if use_n else None
        d_d = torch.empty_like(d) if use_d else None
        p = lambda t: None if t is None else t.data_ptr()
        with torch.cuda.device(dev):
            _lib.check(lib.texgs_geom_losses(p(n), p(gn), p(gi), p(m), p(d), p(gd), H, W, float(lambda_norm), float(lambda_smooth),
                                             float(gamma), float(lambda_depth) if use_d else 0.0, p(sums), p(d_n), p(d_d),
                                             torch.cuda.current_stream(dev).cuda_stream), "texgs_geom_losses")
        loss = sums.new_zeros(())
        stats = {}
        if lambda_norm != 0.0:
            stats["Lnorm"] = sums[1] / (sums[0] + 1e-6)
            loss = loss + lambda_norm * stats["Lnorm"]
        if lambda_smooth != 0.0:
            stats["Lnorm_smooth"] = (sums[6:10] / (sums[2:6] + 1e-6)).sum() / 4
            loss = loss + lambda_smooth * stats["Lnorm_smooth"]
        if use_d:
            stats["Ldepth"] = sums[10] / (H * W)
            loss = loss + lambda_depth * stats["Ldepth"]
        ctx.save_for_backward(d_n if use_n else torch.empty(0, device=dev), d_d if use_d else torch.empty(0, device=dev))
        ctx.use = (use_n, use_d)
        ctx.stats = stats
        return loss

    @staticmethod
    def backward(ctx, g):
        d_n, d_d = ctx.saved_tensors
        use_n, use_d = ctx.use
        return (g * d_n if use_n else None), None, None, None, (g * d_d if use_d else None), None, None, None, None, None


def geom_losses(norm=None, gt_norm=None, gt_image=None, mask=None, depth=None, gt_depth=None, lambda_norm=0.0,
                lambda_smooth=0.0, gamma=0.1, lambda_depth=0.0):
    """lambda_norm * norm_loss(norm, gt_norm, mask) + lambda_smooth * smooth_loss(gt_image, norm, mask, gamma)
    + lambda_depth * l1_loss(depth, gt_depth) with the reference's definitions (losses/norm_reg_loss.py:66-71,
    losses/smooth_loss.py:4-27, losses/pixelwise_loss.py), as the terms of models/texture_gaussian3d.py:347-368 use them;
    differentiable w.r.t. norm and depth -- the gradients go straight into the rasterizer's backward."""
    return _GeomLosses.apply(norm, gt_norm, gt_image, mask, depth, gt_depth, float(lambda_norm), float(lambda_smooth),
                             float(gamma), float(lambda_depth))


def norm_from_depth(depth, world_view_transform, tanfovx, tanfovy, threshold=1e-2):
    """Pseudo-normal [3,H,W] and validity mask [1,H,W] from a depth map [1,H,W] (losses/norm_reg_loss.py:16-63): pixels are
    back-projected with the camera (`world_view_transform` as the reference stores it: row-vector convention, [4,4]), the
    normal is the normalised cross product of the central differences and the mask keeps pixels whose four one-sided
    differences are all shorter than `threshold`.  No gradient (the reference calls it on depth.detach())."""
    lib = _lib.load()
    depth = depth.detach()
    if depth.device.type != "cuda":
        raise RuntimeError("norm_from_depth runs on an AMD GPU (torch device 'cuda' = HIP); there is no CPU fallback")
    if depth.dim() != 3 or depth.shape[0] != 1 or depth.dtype != torch.float32:
        raise ValueError(f"depth must be float32 [1,H,W], got {depth.dtype} {tuple(depth.shape)}")
    depth = depth.contiguous()
    _, H, W = depth.shape
    # the kernel inverts the (device-resident) view matrix itself: no .cpu(), no host-side inverse, no sync inside the loss
    vm = world_view_transform.detach().to(device=depth.device, dtype=torch.float32).contiguous()
    if vm.shape != (4, 4):
        raise ValueError("world_view_transform must be [4,4]")
    norm = torch.empty(3, H, W, dtype=torch.float32, device=depth.device)
    mask = torch.empty(1, H, W, dtype=torch.float32, device=depth.device)
    with torch.cuda.device(depth.device):
        _lib.check(lib.texgs_norm_from_depth(depth.data_ptr(), vm.data_ptr(), float(tanfovx), float(tanfovy), H, W, float(threshold),
                                             norm.data_ptr(), mask.data_ptr(),
                                             torch.cuda.current_stream(depth.device).cuda_stream), "texgs_norm_from_depth")
    return norm, mask


def norm_reg_loss(norm, depth, world_view_transform, tanfovx, tanfovy, gt_alpha, threshold=1e-2):
    """norm_reg_loss of losses/norm_reg_loss.py:73-78 (models/texture_gaussian3d.py:360-363): the rendered normal against the
    pseudo-normal of the rendered (detached) depth, masked by gt_alpha * validity; differentiable w.r.t. `norm`."""
    norm2, mask = norm_from_depth(depth, world_view_transform, tanfovx, tanfovy, threshold)
    return geom_losses(norm=norm, gt_norm=norm2, mask=gt_alpha * mask, lambda_norm=1.0)


def zero_one_loss(value, epsilon: float = 1e-3):
    """losses/zero_one_loss.py:3-7 (`lambda_opacity_reg`, models/texture_gaussian3d.py:370-373): mean(log v + log(1 - v)) of
    the clamped opacities -- a regulariser on the opacity PARAMETERS, not on the operator's outputs, so it stays plain torch
    (one elementwise pass over N values; autograd gives the gradient)."""
    val = torch.clamp(value, epsilon, 1.0 - epsilon)
    return torch.mean(torch.log(val) + torch.log(1.0 - val))


def depth2world(depth, full_proj_transform, zfar: float, znear: float):
    """World-space points [H,W,3] of a depth map [H,W] (the operator's depth output, detached and squeezed:
    models/texture_gaussian3d.py:299-309, used by the inverse-UV term at :394): clip-space point
    (ndc_x d, ndc_y d, zfar d / (zfar - znear) - zfar znear / (zfar - znear), d) times the inverse of the reference's
    row-vector full projection.  Host-side torch on whatever device `depth` lives on (one 4x4 inverse + one [HW,4]x[4,4])."""
    if depth.dim() != 2:
        raise ValueError(f"depth must be [H,W], got {tuple(depth.shape)}")
    H, W = depth.shape
    dev, dt = depth.device, depth.dtype
    ndc_x = (torch.arange(W, device=dev, dtype=dt) * 2 + 1) / W - 1.0
    ndc_y = (torch.arange(H, device=dev, dtype=dt) * 2 + 1) / H - 1.0
    gy, gx = torch.meshgrid(ndc_y, ndc_x, indexing="ij")
    z = zfar * depth / (zfar - znear) - zfar * znear / (zfar - znear)
    clip = torch.stack([gx * depth, gy * depth, z, depth], dim=-1).reshape(-1, 4)
    world = clip @ torch.linalg.inv(full_proj_transform.to(device=dev, dtype=dt))
    return world[:, :3].reshape(H, W, 3)
