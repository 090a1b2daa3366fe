#!/usr/bin/env python3
"""Experiments / evidence: per-kernel times (HIP events, views one after the other) of the backward flavours at C3 size:
full, texture-only (Gaussians frozen), frozen texture, and the untextured `diff_gauss` surface.  One JSON line per variant."""
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "texture-gs_amd")):
    sys.path.insert(0, p)
from texgs import synth, _lib                                               # noqa: E402
from texgs.rasterizer import GaussianRasterizationSettings, GaussianRasterizer   # noqa: E402
import diff_gauss as dg                                                     # noqa: E402

N, R, W, H = 300_000, 1024, 800, 800
dev = torch.device("cuda:0")
scene = synth.make_scene(N, R, seed=0)
cams = synth.fibonacci_cameras(64, W, H)
bg = torch.zeros(3, device=dev)
g = torch.Generator().manual_seed(1234)
P = W * H
g_img = ((torch.rand(3, H, W, generator=g) > 0.5).float() * 2 - 1).to(dev) / (3 * P)
g_alpha = ((torch.rand(1, H, W, generator=g) > 0.5).float() * 2 - 1).to(dev) / P
nh = torch.randn(3, H, W, generator=g)
g_norm = (-0.1 * nh / nh.norm(dim=0, keepdim=True)).to(dev) / P


def settings(cam, cls):
    return cls(image_height=H, image_width=W, tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5), bg=bg,
               scale_modifier=1.0, viewmatrix=cam.world_view_transform.to(dev), projmatrix=cam.full_proj_transform.to(dev),
               sh_degree=3, campos=cam.camera_center.to(dev), prefiltered=False, debug=False)


def run(name, train, surface="textured", views=8, reps=3):
    names = ["means3D", "shs", "opacities", "scales", "rotations", "uvs", "texture"]
    leaves = {n: getattr(scene, n).to(dev).requires_grad_(n in train) for n in names}
    juv = scene.gradient_uvs.to(dev)
    sts = [settings(cams[v], GaussianRasterizationSettings) for v in range(views)]
    if surface == "diff_gauss":
        shs_full = torch.cat([torch.zeros(N, 1, 3), scene.shs], 1).to(dev).requires_grad_("shs" in train)

    def view(v):
        if surface == "diff_gauss":
            out = dg.GaussianRasterizer(sts[v])(means3D=leaves["means3D"], means2D=None, opacities=leaves["opacities"], shs=shs_full,
                                                scales=leaves["scales"], rotations=leaves["rotations"])
        else:
            out = GaussianRasterizer(sts[v])(means3D=leaves["means3D"], means2D=None, shs=leaves["shs"], opacities=leaves["opacities"],
                                             scales=leaves["scales"], rotations=leaves["rotations"], uvs=leaves["uvs"],
                                             gradient_uvs=juv, texture=leaves["texture"], extra_attrs=None)
        if train:
            torch.autograd.backward([out[0], out[3], out[2]], [g_img, g_alpha, g_norm])
    for v in range(views):
        view(v)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        for v in range(views):
            view(v)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / (reps * views)
    _lib.profile_enable(True)
    _lib.profile_read()
    for v in range(views):
        view(v)
    torch.cuda.synchronize()
    k = _lib.profile_read()
    _lib.profile_enable(False)
    table = {n: round(1e3 * ms / c, 1) for n, (ms, c) in k.items() if c}
    print(json.dumps({"variant": name, "surface": surface, "train": sorted(train), "ms_per_view_wall": round(1e3 * wall, 4),
                      "views_per_s": round(1 / wall, 1), "kernel_us": table, "sum_kernel_us": round(sum(table.values()), 1)}), flush=True)


ALL = {"means3D", "shs", "opacities", "scales", "rotations", "uvs", "texture"}
which = sys.argv[1:] or ["full", "texture_only", "frozen_texture", "forward_only", "diff_gauss", "diff_gauss_fwd"]
for w in which:
    if w == "full":
        run(w, ALL)
    elif w == "texture_only":
        run(w, {"texture"})
    elif w == "frozen_texture":
        run(w, ALL - {"texture"})
    elif w == "forward_only":
        run(w, set())
    elif w == "diff_gauss":
        run(w, ALL - {"texture", "uvs"}, surface="diff_gauss")
    elif w == "diff_gauss_fwd":
        run(w, set(), surface="diff_gauss")
