"""Dynamic work counters of K7 for one view of a bench workload (diagnostics; needs a -DK7_STATS build:
  TEXGS_LIB_NAME=libtexgs_stats.so TEXGS_OBJ_DIR=build_stats TEXGS_EXTRA_FLAGS=-DK7_STATS python texture-gs_amd/build.py
  TEXGS_LIB=$PWD/texture-gs_amd/libtexgs_stats.so python scripts/k7_stats.py [c3|c2|c5])"""
import ctypes as C, json, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "texture-gs_amd"))
sys.path.insert(0, ROOT)
import torch
from texgs import synth, _lib
from texgs.rasterizer import GaussianRasterizationSettings, forward_raw, backward_raw
import bench as B

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
N, R, W, H, mode = B.WORKLOADS[wl]
dev = torch.device("cuda:0")
scene = synth.make_scene(N, R, seed=0)
cams = synth.fibonacci_cameras(64, W, H)
lib = _lib.load()
lib.texgs_debug_k7_stats.argtypes = [C.POINTER(C.c_uint64), C.c_int]
names = ["chunks", "instances", "after_cull", "stageA_iters", "inst_with_items", "items", "c2_tasks", "segments", "b_rounds",
         "bin_group_iters", "c2_rounds", "waves"]
tot = [0] * 16
views = [0, 7, 21]
for v in views:
    cam = cams[v]
    st = GaussianRasterizationSettings(cam.image_height, cam.image_width, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5),
                                       torch.zeros(3, device=dev), 1.0, cam.world_view_transform.to(dev),
                                       cam.full_proj_transform.to(dev), 3, cam.camera_center.to(dev), False, False)
    t = {k: getattr(scene, k).to(dev) for k in ["means3D", "shs", "opacities", "scales", "rotations", "uvs", "gradient_uvs", "texture"]}
    outs, s = forward_raw(st, t["means3D"], t["shs"], t["opacities"], t["scales"], t["rotations"], t["uvs"], t["gradient_uvs"], t["texture"])
    g = torch.Generator().manual_seed(1)
    gi = (torch.rand(3, H, W, generator=g) - 0.5).to(dev) / (W * H)
    buf = (C.c_uint64 * 16)()
    lib.texgs_debug_k7_stats(buf, 1)
    backward_raw(s, gi, None, gi * 0.1, gi[:1].contiguous())
    torch.cuda.synchronize()
    lib.texgs_debug_k7_stats(buf, 1)
    for i in range(16):
        tot[i] += buf[i]
d = {n: tot[i] / len(views) for i, n in enumerate(names)}
d["D"] = s.D
d["items_per_b_round"] = d["items"] / max(d["b_rounds"], 1)
d["items_per_inst"] = d["items"] / max(d["inst_with_items"], 1)
d["tasks_per_c2_round"] = d["c2_tasks"] / max(d["c2_rounds"], 1)
d["items_per_task"] = d["items"] / max(d["c2_tasks"], 1)
d["bins_per_round"] = d["bin_group_iters"] / max(d["b_rounds"], 1)
print(json.dumps({"workload": wl, **d}))
