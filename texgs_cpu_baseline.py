"""bench.py's `cpu_baseline` leg: the oracle (a CPU port of the operator; the reference has no CPU rasterizer,
SURVEY.md section 0.4) timed on this box's host cores on a bounded sample of the same workload.  Imported ONLY by
bench.py; the product path never touches oracle/."""
import math
import os
import time

import torch


def cpu_baseline(scene, cam, W, H, with_bwd, cpu_tiles=0):
    from oracle import texgs_torch as O
    from texgs import synth
    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    dtype = torch.float32
    st = O.Settings(cam.image_height, cam.image_width, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5),
                    torch.zeros(3), 1.0, cam.world_view_transform, cam.full_proj_transform, 3, cam.camera_center,
                    False, False)
    names = ["means3D", "shs", "opacities", "scales", "rotations", "uvs", "texture"]
    leaves = {n: getattr(scene, n).clone().to(dtype).requires_grad_(with_bwd) for n in names}
    m2 = torch.zeros_like(leaves["means3D"], requires_grad=with_bwd)
    T = ((W + 15) // 16) * ((H + 15) // 16)
    t0 = time.perf_counter()
    pre = O.preprocess(leaves["means3D"], m2, leaves["shs"], leaves["opacities"], leaves["scales"],
                       leaves["rotations"], leaves["uvs"], scene.gradient_uvs.to(dtype), st, dtype)
    binning = O.bin_and_sort(pre)
    t_geom = time.perf_counter() - t0
    # bounded sample of the blend: every `stride`-th tile, chosen so the sample is O(10 s)
    ntiles = cpu_tiles if cpu_tiles > 0 else min(T, 96)
    stride = max(1, T // ntiles)
    subset = list(range(stride // 2, T, stride))
    t1 = time.perf_counter()
    out, _, _, _ = O.render(pre, binning, leaves["texture"], st, dtype, tile_subset=subset)
    if with_bwd:
        target, nhat = synth.make_targets(H, W, seed=1)
        L = synth.synthetic_loss(out[0:3], out[7:8], out[4:7], target, nhat)
        L.backward()
    t_blend = time.perf_counter() - t1
    # geometry (all Gaussians; its backward ran inside L.backward) is not sampled; the blend scales by tiles
    est = t_geom + t_blend * (T / len(subset))
    return {"value": round(1.0 / est, 5), "unit": "views/s", "cores": cores, "kind": "port",
            "sample": f"1 view of the same scene/camera: preprocess+binning of all Gaussians ({t_geom:.1f} s) + "
                      f"blend fwd{'+bwd' if with_bwd else ''} of {len(subset)} of {T} tiles ({t_blend:.1f} s), blend time "
                      f"scaled by {T / len(subset):.1f}; torch {torch.__version__} float32, {cores} threads",
            "seconds_measured": round(t_geom + t_blend, 2)}
