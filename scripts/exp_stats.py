#!/usr/bin/env python3
"""Dynamic work counters of K6 / K7 for one C3 view (experimental -DTG_STATS build; TEXGS_LIB selects it)."""
import ctypes as C, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "texture-gs_amd")]
import torch
from texgs import synth, _lib
from texgs.rasterizer import GaussianRasterizationSettings, forward_raw, backward_raw
wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
N, R, W, H = dict(c3=(300000, 1024, 800, 800), c5=(1000000, 2048, 1600, 1200), c2=(100000, 512, 800, 800))[wl]
dev = torch.device("cuda:0")
scene = synth.make_scene(N, R, seed=0)
cams = synth.fibonacci_cameras(64, W, H)
t = lambda x: x.to(dev)
inp = [t(scene.means3D), t(scene.shs), t(scene.opacities), t(scene.scales), t(scene.rotations), t(scene.uvs), t(scene.gradient_uvs), t(scene.texture)]
g = torch.Generator().manual_seed(1234)
P = W * H
g_img = ((torch.rand(3, H, W, generator=g) > 0.5).float() * 2 - 1).to(dev) / (3 * P)
g_alpha = ((torch.rand(1, H, W, generator=g) > 0.5).float() * 2 - 1).to(dev) / P
lib = _lib.load()
lib.texgs_debug_stats.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
buf = (C.c_ulonglong * 32)()
views = [0, 7, 19]
for v in views[:1]:      # warm the pool
    cam = cams[v]
    st = GaussianRasterizationSettings(H, W, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), torch.zeros(3, device=dev), 1.0,
                                       t(cam.world_view_transform), t(cam.full_proj_transform), 3, t(cam.camera_center), False, False)
    for _ in range(3):
        outs, s = forward_raw(st, *inp); backward_raw(s, g_img, None, None, g_alpha)
torch.cuda.synchronize(); lib.texgs_debug_stats(buf, 1)
for v in views:
    cam = cams[v]
    st = GaussianRasterizationSettings(H, W, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), torch.zeros(3, device=dev), 1.0,
                                       t(cam.world_view_transform), t(cam.full_proj_transform), 3, t(cam.camera_center), False, False)
    outs, s = forward_raw(st, *inp); backward_raw(s, g_img, None, None, g_alpha)
torch.cuda.synchronize(); lib.texgs_debug_stats(buf, 1)
nv = len(views)
k6 = ["waves", "raw batches", "raw instances", "survivors 8x8", "chunks", "quadrant list entries", "sum tmax", "iterations", "iterations any-ok", "items", "drains", "sum max over 16 quad lists (bound)", "16 quad list entries (bound)", "sum max over 16 quad lists (exact)"]
k7 = ["waves + C2 rounds if tasks <= 8 items were paired", "C2 tasks with <= 8 items", "C2 tasks with <= 4 items", "survivors 8x8", "chunks", "quadrant list entries", "sum tmax", "iterations", "productive iterations", "items", "segments", "B rounds", "C2 tasks", "C2 rounds", "distinct Gaussians per segment (sum)", "C2 tasks if merged per Gaussian"]
print("workload", wl, "D", s.D, "mean of", nv, "views")
for i, n in enumerate(k6): print(f"K6 {n:26s} {buf[i] / nv:14.0f}")
for i, n in enumerate(k7): print(f"K7 {n:26s} {buf[16 + i] / nv:14.0f}")
