#!/usr/bin/env python3
"""Host-side cost of one forward (experiment): cProfile of forward-only calls at C2 size through the public module."""
import cProfile
import math
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "texture-gs_amd")):
    sys.path.insert(0, p)
from texgs import synth                                                      # noqa: E402
from texgs.rasterizer import GaussianRasterizationSettings, GaussianRasterizer   # noqa: E402

dev = torch.device("cuda:0")
N, R, W, H = 100_000, 512, 800, 800
scene = synth.make_scene(N, R, seed=0)
cams = synth.fibonacci_cameras(16, W, H)
L = {n: getattr(scene, n).to(dev) for n in ["means3D", "shs", "opacities", "scales", "rotations", "uvs", "texture"]}
juv = scene.gradient_uvs.to(dev)
m2 = torch.zeros(N, 3, device=dev)
bg = torch.zeros(3, device=dev)
rasters = []
for cam in cams:
    st = GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5), bg=bg,
                                       scale_modifier=1.0, viewmatrix=cam.world_view_transform.to(dev), projmatrix=cam.full_proj_transform.to(dev),
                                       sh_degree=3, campos=cam.camera_center.to(dev), prefiltered=False, debug=False)
    rasters.append(GaussianRasterizer(st))


def run(n):
    with torch.no_grad():
        for i in range(n):
            rasters[i % 16](means3D=L["means3D"], means2D=m2, shs=L["shs"], opacities=L["opacities"], scales=L["scales"], rotations=L["rotations"],
                            uvs=L["uvs"], gradient_uvs=juv, texture=L["texture"])


run(64)
torch.cuda.synchronize()
t0 = time.perf_counter(); run(400); torch.cuda.synchronize(); t1 = time.perf_counter()
print("wall us per forward", round(1e6 * (t1 - t0) / 400, 1))
pr = cProfile.Profile()
pr.enable(); run(400); pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)
