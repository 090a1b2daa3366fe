#!/usr/bin/env python3
"""A/B check on the GPU box: the same seeded scene through an OLD build of the package (an exported tree under build_exp/old,
see the usage line) and through the working tree.  K6's colour sum is order-independent fixed point and the per-pixel
transmittance chain runs in list order in both, so the forward images must agree to a few ulp wherever the per-item math is
unchanged; gradients agree to atomics' rounding.

usage:  python scripts/ab_compare.py dump <pkg_dir> <out.pt> [N R W H views]     (run once per tree)
        python scripts/ab_compare.py cmp <a.pt> <b.pt>
"""
import math
import os
import sys

import torch


def dump(pkg, out, N=20000, R=256, W=400, H=304, views=3):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [pkg, root]
    from texgs import synth
    from texgs.rasterizer import GaussianRasterizationSettings, forward_raw, backward_raw
    dev = torch.device("cuda:0")
    scene = synth.make_scene(N, R, seed=3, random_jacobian=True)
    cams = synth.fibonacci_cameras(max(views, 4), W, H)
    t = lambda x: x.to(dev)
    g = torch.Generator().manual_seed(7)
    dimg = (torch.randn(3, H, W, generator=g) / (H * W)).to(dev)
    dnorm = (torch.randn(3, H, W, generator=g) / (H * W)).to(dev)
    dalpha = (torch.randn(1, H, W, generator=g) / (H * W)).to(dev)
    ddepth = (torch.randn(1, H, W, generator=g) / (H * W)).to(dev)
    res = {}
    for v in range(views):
        cam = cams[v]
        st = GaussianRasterizationSettings(H, W, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), torch.tensor([0.1, 0.2, 0.3], device=dev),
                                           1.0, t(cam.world_view_transform), t(cam.full_proj_transform), 3, t(cam.camera_center), False, False)
        outs, s = forward_raw(st, t(scene.means3D), t(scene.shs), t(scene.opacities), t(scene.scales), t(scene.rotations),
                              t(scene.uvs), t(scene.gradient_uvs), t(scene.texture))
        gr = backward_raw(s, dimg, ddepth, dnorm, dalpha)
        torch.cuda.synchronize()
        res[f"v{v}"] = dict(out=[o.cpu() for o in outs[:4]], radii=outs[4].cpu(), final_T=s.tensors["final_T"].cpu(),
                            n_contrib=s.tensors["n_contrib"].cpu(), grads=[None if x is None else x.cpu() for x in gr], D=s.D)
    torch.save(res, out)
    print("dumped", out, {k: v["D"] for k, v in res.items()})


def cmp(a, b):
    A, B = torch.load(a), torch.load(b)
    names = ["image", "depth", "norm", "alpha"]
    gn = ["means3D", "means2D", "shs", "opacities", "scales", "rotations", "uvs", "texture"]
    for k in A:
        x, y = A[k], B[k]
        print(k, "D", x["D"], y["D"], "radii equal", bool(torch.equal(x["radii"], y["radii"])),
              "n_contrib equal", bool(torch.equal(x["n_contrib"], y["n_contrib"])), "final_T equal", bool(torch.equal(x["final_T"], y["final_T"])))
        for n, p, q in zip(names, x["out"], y["out"]):
            d = (p.double() - q.double()).abs()
            print(f"   {n:6s} bit-equal {bool(torch.equal(p, q))}  max|diff| {float(d.max()):.3e}  pixels differing {int((d > 0).sum())}")
        for n, p, q in zip(gn, x["grads"], y["grads"]):
            if p is None or q is None:
                continue
            rel = float((p.double() - q.double()).norm() / q.double().norm().clamp_min(1e-300))
            print(f"   d{n:10s} rel L2 {rel:.3e}  max|diff|/max {float((p.double() - q.double()).abs().max() / q.double().abs().max()):.3e}")


if __name__ == "__main__":
    if sys.argv[1] == "dump":
        extra = [int(x) for x in sys.argv[4:]]
        dump(sys.argv[2], sys.argv[3], *extra)
    else:
        cmp(sys.argv[2], sys.argv[3])
