"""Distribution of texture-gradient records over the 32x32-texel bins of C3 views, and what launch order / list splitting
would buy k_texgrad_reduce (a list-scheduling model: `slots` workgroups in flight, cost = c0 + records).  GPU only."""
import heapq, json, math, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "texture-gs_amd"))
from texgs import synth
from texgs import rasterizer as RZ
from texgs.rasterizer import GaussianRasterizationSettings, GaussianRasterizer

N, R, W, H = 300000, 1024, 800, 800
dev = torch.device("cuda", 0)
scene = synth.make_scene(N, R, seed=0)
cams = synth.fibonacci_cameras(64, W, H)
leaves = {n: getattr(scene, n).to(dev).requires_grad_(True) for n in ["means3D", "shs", "opacities", "scales", "rotations", "uvs", "texture"]}
juv = scene.gradient_uvs.to(dev)
means2D = torch.zeros(N, 3, device=dev, requires_grad=True)
bg = torch.zeros(3, device=dev)


def makespan(costs, slots):
    h = [0.0] * slots
    heapq.heapify(h)
    for c in costs:
        heapq.heappush(h, heapq.heappop(h) + c)
    return max(h)


out = []
for v in (0, 17, 40):
    cam = cams[v]
    st = GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5),
                                       bg=bg, scale_modifier=1.0, viewmatrix=cam.world_view_transform.to(dev),
                                       projmatrix=cam.full_proj_transform.to(dev), sh_degree=3, campos=cam.camera_center.to(dev),
                                       prefiltered=False, debug=False)
    r = GaussianRasterizer(st)(means3D=leaves["means3D"], means2D=means2D, shs=leaves["shs"], opacities=leaves["opacities"],
                               scales=leaves["scales"], rotations=leaves["rotations"], uvs=leaves["uvs"], gradient_uvs=juv,
                               texture=leaves["texture"], extra_attrs=None)
    (r[0].sum() + r[3].sum()).backward()
    torch.cuda.synchronize()
    bins = [sc.bins for sc in RZ._SCRATCH.values() if sc.bins is not None][0]
    base = bins.base[:bins.nbins + 1].cpu().numpy().astype(np.int64)
    cnt = np.diff(base)
    nz = cnt[cnt > 0]
    c0 = 600.0          # records' worth of fixed cost per workgroup (tile clear + write-out of 33x37x3 texels)
    cost = np.where(cnt > 0, c0 + cnt, 5.0)
    slots = 256 * 5
    lpt = np.sort(cost)[::-1]
    S = 8192
    split = np.concatenate([np.full(int(c // S), c0 + S) for c in nz] + [np.array([c0 + (c % S) for c in nz if c % S])])
    d = dict(view=v, records=int(cnt.sum()), bins=int(cnt.size), nonempty=int(nz.size), mean_nonempty=float(nz.mean()),
             p50=float(np.percentile(nz, 50)), p90=float(np.percentile(nz, 90)), p99=float(np.percentile(nz, 99)), max=int(nz.max()),
             ideal=float(cost.sum() / slots), makespan_index_order=makespan(cost, slots), makespan_lpt=makespan(lpt, slots),
             makespan_split8k_lpt=makespan(np.sort(split)[::-1], slots), blocks_split=int(split.size))
    out.append(d)
    print(json.dumps(d))
