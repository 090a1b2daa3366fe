#!/usr/bin/env python3
"""`bench.py --leg iteration`: what ONE training iteration of the reference's texture stage costs on this stack.

The reference's iteration after iteration 10 000 with the shipped configs/texture_gaussian3d.yaml
(models/texture_gaussian3d.py:315-418, :420-444):
    get_uvs / get_grad_uvs (UVNet + its Jacobian)            :216-236      -> texgs.uvnet.UVNet.uvs_and_jacobian_with_grad
    render at the active SH degree                           :318          -> diff_gauss_uv_tex.GaussianRasterizer
    L1 + D-SSIM on the image, L1 on alpha                    :333-345      -> texgs.losses.rgb_alpha_loss
    normal loss + bilateral normal smoothness                :354-368      -> texgs.losses.geom_losses
    render AGAIN at sh_degree 0, L1 + D-SSIM x lambda_no_sh  :375-389      -> second render on the first one's lists
    loss.backward()                                          :410          -> both rasterizer backwards, activations, UVNet
    zero_grad(set_to_none=True)                              :442-444
Optimizer steps are not part of the path (plain torch.optim in the reference) and are not timed.

Two variants: `uv_per_render` evaluates the UV map for each render as the reference's getters do (:216-236 run per render()
call); `uv_once` evaluates it once per iteration for both renders (the reference's own `_uv` / `_grad_uv` cache, :218-219,
231-232, is the licence: same Gaussians, same weights within an iteration).  Prints one JSON object: ms per iteration of both,
the split of an iteration by phase (HIP events, phases synchronised -- so their sum exceeds the free-running figure), and the
library's per-kernel table."""
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "texture-gs_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def run(N=300_000, R=1024, W=800, H=800, iters=12, warm=4, dev_index=0, precision="mixed"):
    from texgs import synth, _lib, losses as LS
    from texgs import rasterizer as RZ
    from texgs.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
    from texgs.uvnet import UVNet
    dev = torch.device("cuda", dev_index)
    scene = synth.make_scene(N, R, seed=0)
    cams = synth.fibonacci_cameras(64, W, H)
    bg = torch.zeros(3, device=dev)
    P = lambda t: t.to(dev).requires_grad_(True)
    raw = dict(xyz=P(scene.means3D), shs=P(scene.shs), rotation=P(scene.rotations), texture=P(scene.texture),
               scaling=P(scene.scales.log()), opacity=P(torch.logit(scene.opacities.clamp(1e-6, 1 - 1e-6))))
    torch.manual_seed(0)
    net = UVNet(precision=precision).to(dev)
    emb = (0.2 * torch.randn(128)).to(dev).requires_grad_(True)
    # A UV map that is a UV map: fitted to phi(x) = x / |x| on this scene (the reference's stage 2 trains it to a bijection of the
    # surface onto the sphere; an untrained MLP sends every Gaussian to the same few texels -- all 19 M footprints of a view in a
    # handful of texture bins: a load no trained model produces and not what an iteration costs)
    opt = torch.optim.Adam(list(net.parameters()) + [emb], lr=2e-3)
    xs = raw["xyz"].detach()
    for _ in range(400):
        idx = torch.randint(0, N, (16384,), device=dev)
        x = xs[idx]
        loss = (1.0 - (net(x, emb) * torch.nn.functional.normalize(x, dim=1)).sum(1)).mean()
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
    with torch.no_grad():
        fit = float((1.0 - (net(xs[:65536], emb) * torch.nn.functional.normalize(xs[:65536], dim=1)).sum(1)).mean())
    net.invalidate_packed()
    params = list(raw.values()) + list(net.parameters()) + [emb]
    g = torch.Generator().manual_seed(7)
    gt_image = torch.rand(3, H, W, generator=g).to(dev)
    gt_alpha = (torch.rand(1, H, W, generator=g) > 0.3).float().to(dev)
    gt_norm = torch.nn.functional.normalize(torch.randn(3, H, W, generator=g), dim=0).to(dev)

    def settings(v, deg):
        cam = cams[v]
        return GaussianRasterizationSettings(
            image_height=H, image_width=W, tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5), bg=bg,
            scale_modifier=1.0, viewmatrix=cam.world_view_transform.to(dev), projmatrix=cam.full_proj_transform.to(dev),
            sh_degree=deg, campos=cam.camera_center.to(dev), prefiltered=False, debug=False)
    sts = {v: (settings(v, 3), None) for v in range(16)}
    sts = {v: (s3, s3._replace(sh_degree=0)) for v, (s3, _) in sts.items()}      # same camera tensors: the lists are shared

    def render(st, uvs, juv):
        m2 = torch.zeros_like(raw["xyz"], requires_grad=True) + 0
        return GaussianRasterizer(st)(means3D=raw["xyz"], means2D=m2, shs=raw["shs"], opacities=torch.sigmoid(raw["opacity"]),
                                      scales=torch.exp(raw["scaling"]), rotations=torch.nn.functional.normalize(raw["rotation"]),
                                      uvs=uvs, gradient_uvs=juv, texture=raw["texture"], extra_attrs=None)

    ev = lambda: torch.cuda.Event(enable_timing=True)

    def iteration(v, uv_once, marks=None):
        def mark(name):
            if marks is not None:
                e = ev(); e.record(); marks.append((name, e))
        mark("start")
        uvs, juv = net.uvs_and_jacobian_with_grad(raw["xyz"], emb)
        mark("uv_map")
        image, depth, norm, alpha, radii, _ = render(sts[v][0], uvs, juv)
        mark("render_sh3")
        loss = LS.rgb_alpha_loss(image, gt_image, alpha, gt_alpha, 0.2, 1.0)
        loss = loss + LS.geom_losses(norm=norm, gt_norm=gt_norm, gt_image=gt_image, mask=gt_alpha, lambda_norm=0.1, lambda_smooth=0.5,
                                     gamma=0.1)
        mark("losses")
        if not uv_once:
            uvs, juv = net.uvs_and_jacobian_with_grad(raw["xyz"], emb)
            mark("uv_map_2")
        image0 = render(sts[v][1], uvs, juv)[0]
        mark("render_sh0")
        loss = loss + 2.0 * LS.rgb_alpha_loss(image0, gt_image, None, None, 0.2, 0.0)
        mark("loss_sh0")
        loss.backward()
        mark("backward")
        for p_ in params:
            p_.grad = None
        return loss

    out = {}
    for label, once in (("uv_per_render", False), ("uv_once", True)):
        RZ.release_scratch(dev)
        for k in range(warm):
            iteration(k % 16, once)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for k in range(iters):
            iteration(k % 16, once)
        torch.cuda.synchronize(dev)
        out[label + "_ms_per_iteration"] = round(1e3 * (time.perf_counter() - t0) / iters, 4)
    # split by phase (events on the op's stream; no host sync inside an iteration, so the phases are GPU time between marks)
    acc = {}
    for k in range(iters):
        marks = []
        iteration(k % 16, True, marks)
        torch.cuda.synchronize(dev)
        for (n0, e0), (n1, e1) in zip(marks[:-1], marks[1:]):
            acc[n1] = acc.get(n1, 0.0) + e0.elapsed_time(e1)
    out["split_uv_once_ms"] = {n: round(v / iters, 4) for n, v in acc.items()}
    _lib.profile_enable(True)
    _lib.profile_read()
    for k in range(4):
        iteration(k % 16, True)
    torch.cuda.synchronize(dev)
    kt = _lib.profile_read()
    _lib.profile_enable(False)
    out["rasterizer_kernels_us_per_iteration"] = {n: round(1e3 * ms / 4, 1) for n, (ms, c) in kt.items() if c}
    # the UV map's own backward, alone: the fused kernel against the per-layer library chain it replaced (same inputs)
    from texgs.uvnet import uvnet_backward
    gq = torch.randn(N, 3, device=dev) / N
    lins = net._linears()
    ws, bs = [l.weight.detach() for l in lins], [l.bias.detach() for l in lins]

    def timed(fn, n=6):
        fn(); torch.cuda.synchronize(dev)
        e0, e1 = ev(), ev()
        e0.record()
        for _ in range(n):
            fn()
        e1.record(); torch.cuda.synchronize(dev)
        return round(1e3 * e0.elapsed_time(e1) / n, 1)
    with torch.no_grad():
        out["uvnet_backward_us"] = {"fused_hip_kernel": timed(lambda: net.backward_fused(xs, emb, gq)),
                                    "library_gemm_chain": timed(lambda: uvnet_backward(net._norm_in(xs), emb.detach(), ws, bs, gq))}
    out["geometry_cache"] = RZ.geometry_cache_stats()
    out["uv_map_fit_mean_1_minus_cos"] = round(fit, 6)
    out["config"] = (f"N={N}, R={R}, {W}x{H}, sh_degree 3 then 0, UVNet 3-128-128|128-128-128-3 fitted to x/|x| (400 Adam steps), "
                     f"fused {net.precision} MFMA forward + Jacobian")
    out["note"] = ("one view per iteration through plain autograd (reference call pattern), gradients dropped with set_to_none after "
                   "every iteration; optimizer steps not included")
    return out


if __name__ == "__main__":
    print(json.dumps({"metric": "reference training iteration (texture stage), ms per iteration", **run()}), flush=True)
