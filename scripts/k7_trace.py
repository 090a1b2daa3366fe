#!/usr/bin/env python3
"""Experiment (GPU box): WHEN and WHERE each 8x8 block of one K7 launch ran.  Needs the trace build
(scripts/exp_build.sh trace "-DK7_TRACE" texture-gs_amd/csrc/render.hip) selected with TEXGS_LIB.
Question: K7 keeps 2.96 of its 4 wave slots per SIMD busy on average (SQ_WAVE_CYCLES / duration) -- is the idle quarter a tail
(a few long blocks finishing alone), an imbalance between the 8 XCDs, or spread over the launch?  Prints one JSON object."""
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
dbg.texgs_debug_k7_trace.argtypes = [C.c_void_p]
dbg.texgs_debug_k7_trace.restype = C.c_int
res = []
for v in (0, 0, 17, 40):                                  # (the first pass of view 0 warms up)
    view(v)
    buf = np.zeros(4 * 32768, dtype=np.uint64)
    rc = dbg.texgs_debug_k7_trace(buf.ctypes.data_as(C.c_void_p))
    assert rc == 0, rc
    tr = buf.reshape(-1, 4)
    nb = int(os.environ.get("K7_TRACE_NB", "10016"))
    t0, t1 = tr[:nb, 0].astype(np.int64), tr[:nb, 1].astype(np.int64)
    ran = t1 > 0
    base = t0[ran].min()
    s, e = (t0 - base) / 100.0, (t1 - base) / 100.0           # us (100 MHz)
    dur = (e - s)
    xcc = (tr[:nb, 2] & np.uint64(15)).astype(np.int64)
    hwid = (tr[:nb, 2] >> np.uint64(8)).astype(np.int64)
    cu = (hwid >> 8) & 15; se = (hwid >> 13) & 7; simd = (hwid >> 4) & 3
    ns = (tr[:nb, 3] & np.uint64(0xFFFFFFFF)).astype(np.int64)
    todo = (tr[:nb, 3] >> np.uint64(32)).astype(np.int64)
    span = float(e[ran].max())
    slots = 256 * 16
    # occupancy over time: active blocks at 200 sample points
    ts = np.linspace(0, span, 201)[:-1] + span / 400
    occ = np.array([int(((s <= t) & (e > t) & ran).sum()) for t in ts])
    per_xcc_end = {int(x): round(float(e[ran & (xcc == x)].max()), 1) for x in sorted(set(xcc[ran].tolist()))}
    per_xcc_work = {int(x): round(float(dur[ran & (xcc == x)].sum()) / 1e3, 2) for x in sorted(set(xcc[ran].tolist()))}
    per_xcc_blocks = {int(x): int((ran & (xcc == x)).sum()) for x in sorted(set(xcc[ran].tolist()))}
    order = np.argsort(-dur)
    # how well does the dispatch order (rank = list length of the 16x16 tile) predict the block's time?
    bidx = np.arange(nb)
    r_todo = float(np.corrcoef(todo[ran], dur[ran])[0, 1])
    r_ns = float(np.corrcoef(ns[ran], dur[ran])[0, 1])
    # start time vs block index: is dispatch in order?
    late_start = float(np.percentile(s[ran], 99))
    res.append({
        "view": v, "blocks_ran": int(ran.sum()), "span_us": round(span, 1),
        "sum_block_time_ms": round(float(dur[ran].sum()) / 1e3, 2),
        "slot_utilisation": round(float(dur[ran].sum()) / (span * slots), 3),
        "ideal_span_if_all_slots_busy_us": round(float(dur[ran].sum()) / slots, 1),
        "block_us_percentiles": {str(p): round(float(np.percentile(dur[ran], p)), 1) for p in (10, 50, 90, 99, 100)},
        "longest_blocks": [{"block": int(i), "start": round(float(s[i]), 1), "us": round(float(dur[i]), 1), "survivors": int(ns[i]), "list": int(todo[i]),
                            "xcc": int(xcc[i])} for i in order[:6]],
        "occupancy_of_4096_slots_by_time_decile": [round(float(occ[20 * k:20 * k + 20].mean()) / slots, 3) for k in range(10)],
        "time_when_occupancy_drops_below_half_us": round(float(ts[int(np.argmax(occ)) + np.nonzero(occ[int(np.argmax(occ)):] < slots / 2)[0][0]])
                                                         if (occ[int(np.argmax(occ)):] < slots / 2).any() else span, 1),
        "max_concurrent_blocks": int(occ.max()),
        "xcc_last_end_us": per_xcc_end, "xcc_sum_block_ms": per_xcc_work, "xcc_blocks": per_xcc_blocks,
        "start_p99_us": round(late_start, 1),
        "corr_block_time_vs_list_length": round(r_todo, 3), "corr_block_time_vs_survivors": round(r_ns, 3),
        "distinct_cu_se_simd": [int(len(set(zip(xcc[ran].tolist(), se[ran].tolist(), cu[ran].tolist())))), int(len(set(simd[ran].tolist())))],
    })
print(json.dumps({"k7_trace": res[1:]}), flush=True)
