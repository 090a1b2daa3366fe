#!/usr/bin/env python3
"""Experiment (GPU box): where the blocks of one item-stream K7 launch spend their time.  Needs the trace build
(scripts/exp_build.sh k7strace "-DK7S_TRACE" texture-gs_amd/csrc/render.hip) selected with TEXGS_LIB.  Phases per segment (core
clocks, summed per block; every phase boundary but the first is a full s_waitcnt):
  0 front: item wait, record gathers + their round trip, UV / cube address, taps issued, next items issued
  1 stage C of the previous segment (+ the memory wait it did not cover)      2 derive      3 back      4 the last stage C"""
import ctypes as C
import json
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "texture-gs_amd")):
    sys.path.insert(0, p)
from texgs import synth, _lib                                               # noqa: E402
from texgs.rasterizer import GaussianRasterizationSettings, GaussianRasterizer   # noqa: E402

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
names = ["means3D", "shs", "opacities", "scales", "rotations", "uvs", "texture"]
leaves = {n: getattr(scene, n).to(dev).requires_grad_(True) for n in names}
juv = scene.gradient_uvs.to(dev)


def view(v):
    cam = cams[v]
    st = GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5),
                                       bg=bg, scale_modifier=1.0, viewmatrix=cam.world_view_transform.to(dev),
                                       projmatrix=cam.full_proj_transform.to(dev), sh_degree=3, campos=cam.camera_center.to(dev),
                                       prefiltered=False, debug=False)
    out = GaussianRasterizer(st)(means3D=leaves["means3D"], means2D=None, shs=leaves["shs"], opacities=leaves["opacities"],
                                 scales=leaves["scales"], rotations=leaves["rotations"], uvs=leaves["uvs"], gradient_uvs=juv,
                                 texture=leaves["texture"], extra_attrs=None)
    torch.autograd.backward([out[0], out[3], out[2]], [g_img, g_alpha, g_norm])
    torch.cuda.synchronize()


lib = _lib.load()
dbg = C.CDLL(os.environ["TEXGS_LIB"])
dbg.texgs_debug_k7s_trace.argtypes = [C.c_void_p]
dbg.texgs_debug_k7s_trace.restype = C.c_int
for v in (0, 0, 17):
    view(v)
    buf = np.zeros(8 * 32768, dtype=np.uint64)
    assert dbg.texgs_debug_k7s_trace(buf.ctypes.data_as(C.c_void_p)) == 0
    tr = buf.reshape(-1, 8)
    tr = tr[tr[:, 2] > 0]
    n = tr[:, 2].astype(np.float64)
    segs = np.ceil(n / 64)
    ph = tr[:, 3:8].astype(np.float64)
    wall_us = (tr[:, 1] - tr[:, 0]).astype(np.float64) / 100.0          # wall_clock64: 100 MHz
    tot = ph.sum(1)
    print(json.dumps({
        "view": v, "blocks": int(len(tr)), "items": int(n.sum()), "segments": int(segs.sum()),
        "span_us": float((tr[:, 1].max() - tr[:, 0].min()) / 100.0),
        "block_us_p50_p90_max": [float(np.percentile(wall_us, q)) for q in (50, 90, 100)],
        "clocks_per_segment_by_phase": [round(float(ph[:, k].sum() / segs.sum()), 1) for k in range(5)],
        "share_by_phase": [round(float(ph[:, k].sum() / tot.sum()), 4) for k in range(5)],
        "clocks_per_segment_total": round(float(tot.sum() / segs.sum()), 1),
        "core_clock_GHz_est": round(float((tot / np.maximum(wall_us, 1e-9)).mean() / 1e3), 3),
        "longest_block": {"items": int(n.max()), "us": float(wall_us[n.argmax()]), "us_per_segment": float(wall_us[n.argmax()] / segs[n.argmax()])},
    }), flush=True)
