"""bench.py's `cpu_baseline` leg: the C oracle (oracle/texgs_ref.c -- a CPU port of the operator; the reference has no
CPU rasterizer) timed on this box's host cores, OpenMP over Gaussians / tiles, on whole views
of the same scene and camera.  Imported ONLY by bench.py; the product path never touches oracle/."""
import math
import os
import time

import numpy as np
import torch


def cpu_baseline(scene, cam, W, H, with_bwd, budget_s=20.0):
    from oracle import texgs_ref as CR
    from oracle import texgs_torch as O
    st = O.Settings(cam.image_height, cam.image_width, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5),
                    torch.zeros(3), 1.0, cam.world_view_transform, cam.full_proj_transform, 3, cam.camera_center,
                    False, False)
    run = CR.RefRun(scene, st)
    g = np.random.RandomState(1234)
    dout = (g.randn(8, H, W) / (H * W)).astype(np.float32)
    dout[3] = 0.0                                   # the bench's upstream grads: image, norm, alpha
    views, t0 = 0, time.perf_counter()
    while True:
        run.forward()
        if with_bwd:
            run.backward(dout)
        views += 1
        el = time.perf_counter() - t0
        if el > budget_s or views >= 8:
            break
    return {"value": round(views / el, 4), "unit": "views/s", "cores": run.threads, "kind": "port",
            "sample": f"{views} whole view(s) fwd{'+bwd' if with_bwd else ''} of the same scene/camera by oracle/texgs_ref.c "
                      f"(gcc -O2 -fopenmp, fp32, {run.threads} OpenMP threads) in {el:.1f} s; D={run.D}",
            "seconds_measured": round(el, 2)}
