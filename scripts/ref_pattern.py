#!/usr/bin/env python3
"""The reference's call pattern around the operator, alone, for profiling: per view -- activation outputs as operator
inputs (sigmoid / exp / normalize), a fresh non-leaf zero `means2D`, a new GaussianRasterizer module (render/uv_tex_render.py:15-66),
plain autograd backward, grads dropped with set_to_none (models/texture_gaussian3d.py:442-444).  C3 sizes.
Prints one JSON line: wall ms per view, the library's own kernels' time per view (HIP events, separate pass), and -- when run under
`rocprofv3 --kernel-trace --stats` (scripts/prof_ref_pattern.sh) -- the trace holds every other kernel (torch's elementwise / fill /
copy kernels) for the table in profiles/.  usage: ref_pattern.py [views] [mode: plain | iteration]"""
import json, math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "texture-gs_amd")]
import torch
from texgs import synth, _lib
from texgs.rasterizer import GaussianRasterizationSettings, GaussianRasterizer

views = int(sys.argv[1]) if len(sys.argv) > 1 else 32
mode = sys.argv[2] if len(sys.argv) > 2 else "plain"
N, R, W, H = 300_000, 1024, 800, 800
dev = torch.device("cuda:0")
scene = synth.make_scene(N, R, seed=0)
cams = synth.fibonacci_cameras(64, W, H)
bg = torch.zeros(3, device=dev)
P = lambda t: t.to(dev).requires_grad_(True)
raw = dict(means3D=P(scene.means3D), shs=P(scene.shs), rotations=P(scene.rotations), uvs=P(scene.uvs), texture=P(scene.texture),
           scales=P(scene.scales.log()), opacities=P(torch.logit(scene.opacities.clamp(1e-6, 1 - 1e-6))))
juv = scene.gradient_uvs.to(dev)
g = torch.Generator().manual_seed(1234)
PX = W * H
g_img = ((torch.rand(3, H, W, generator=g) > 0.5).float() * 2 - 1).to(dev) / (3 * PX)
g_alpha = ((torch.rand(1, H, W, generator=g) > 0.5).float() * 2 - 1).to(dev) / PX
nh = torch.randn(3, H, W, generator=g)
g_norm = (-0.1 * nh / nh.norm(dim=0, keepdim=True)).to(dev) / PX


def settings(v, deg=3):
    cam = cams[v]
    return GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5),
                                         bg=bg, scale_modifier=1.0, viewmatrix=cam.world_view_transform.to(dev),
                                         projmatrix=cam.full_proj_transform.to(dev), sh_degree=deg, campos=cam.camera_center.to(dev),
                                         prefiltered=False, debug=False)


sts = {v: settings(v) for v in range(16)}
sts0 = {v: sts[v]._replace(sh_degree=0) for v in sts}


def render(st):
    m2 = torch.zeros_like(raw["means3D"], requires_grad=True) + 0
    m2.retain_grad()
    return GaussianRasterizer(st)(means3D=raw["means3D"], means2D=m2, shs=raw["shs"], opacities=torch.sigmoid(raw["opacities"]),
                                  scales=torch.exp(raw["scales"]), rotations=torch.nn.functional.normalize(raw["rotations"]),
                                  uvs=raw["uvs"], gradient_uvs=juv, texture=raw["texture"], extra_attrs=None)


def view(v):
    out = render(sts[v])
    if mode == "iteration":
        out0 = render(sts0[v])
        torch.autograd.backward([out[0], out[3], out[2], out0[0]], [g_img, g_alpha, g_norm, 2.0 * g_img])
    else:
        torch.autograd.backward([out[0], out[3], out[2]], [g_img, g_alpha, g_norm])
    for p in raw.values():
        p.grad = None


for v in range(6):
    view(v % 16)
torch.cuda.synchronize()
t0 = time.perf_counter()
for v in range(views):
    view(v % 16)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / views
# host time alone: the same loop's Python / launch cost when the GPU is not waited for, measured as time to ISSUE (no sync inside)
t0 = time.perf_counter()
for v in range(views):
    view(v % 16)
issue = (time.perf_counter() - t0) / views
torch.cuda.synchronize()
_lib.profile_enable(True)
_lib.profile_read()
for v in range(8):
    view(v % 16)
torch.cuda.synchronize()
k = _lib.profile_read()
_lib.profile_enable(False)
ours = {n: round(1e3 * ms / 8, 1) for n, (ms, c) in k.items() if c}
print(json.dumps({"mode": mode, "views": views, "wall_ms_per_view": round(1e3 * wall, 4), "views_per_s": round(1 / wall, 1),
                  "host_issue_ms_per_view": round(1e3 * issue, 4), "library_kernels_us_per_view": ours,
                  "library_kernels_sum_us": round(sum(ours.values()), 1)}), flush=True)
