#!/usr/bin/env python3
"""Drive the kernels of the "next" rows for profiling: k_uv_taylor at N = 300k, the loss front-end kernels
at 800x800.  Run it under `rocprofv3 --kernel-trace --stats` / `--pmc ...` (scripts/prof_next_rows.sh); prints HIP-event timings
and the roofline figures that go with them (MFMA fp32 peak 157.3 TF for the former, HBM 8 TB/s for the rest)."""
import json, math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "texture-gs_amd")]
import torch
from texgs.uvnet import UVNet
from texgs import losses as LS
dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10


def timed(fn, n=reps):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3      # us


out = {}
torch.manual_seed(0)
N = 300_000
net = UVNet(xyz_offset=[0.1, -0.2, 0.05], xyz_scale=[1.5, 0.8, 1.2]).to(dev)
emb = (torch.randn(128) * 0.2).to(dev)
xyz = torch.nn.functional.normalize(torch.randn(N, 3), dim=1).to(dev)
us = timed(lambda: net.uv_and_jacobian(xyz, emb))
flop = N * 4 * (3 * 128 * 128) * 2                       # three 128x128 layers x (value + 3 tangents), the MFMA part
out["uv_taylor"] = dict(us=us, gflop=flop / 1e9, tflops=flop / us / 1e6, frac_of_fp32_mfma_peak=flop / us / 1e6 / 157.3, bound="mfma")
H = W = 800
P = H * W
img = torch.rand(3, H, W, device=dev, requires_grad=True); gt = torch.rand(3, H, W, device=dev)
al = torch.rand(1, H, W, device=dev, requires_grad=True); ga = (torch.rand(1, H, W, device=dev) > 0.5).float()


def l1():
    img.grad = None; al.grad = None
    LS.rgb_alpha_loss(img, gt, al, ga, 0.2, 0.1).backward()


us = timed(l1)
by = (3 * 2 + 3 + 3) * P * 4 + 2 * 9 * P * 4        # I, Igt read, dI written; A, Agt read, dA written; 9 scratch planes written + read
out["rgb_alpha_loss(fwd+bwd)"] = dict(us=us, alg_MB=by / 1e6, GBps=by / us / 1e3, frac_of_hbm_peak=by / us / 1e3 / 8000, bound="hbm")
nrm = torch.nn.functional.normalize(torch.randn(3, H, W, device=dev), dim=0).requires_grad_(True)
gn = torch.nn.functional.normalize(torch.randn(3, H, W, device=dev), dim=0)
dep = (torch.rand(1, H, W, device=dev) + 2).requires_grad_(True); gd = torch.rand(1, H, W, device=dev) + 2


def l2():
    nrm.grad = None; dep.grad = None
    LS.geom_losses(norm=nrm, gt_norm=gn, gt_image=gt, mask=ga, depth=dep, gt_depth=gd, lambda_norm=0.1, lambda_smooth=0.5, gamma=0.1, lambda_depth=0.1).backward()


us = timed(l2)
by = (3 + 3 + 3 + 1 + 1 + 1 + 3 + 1) * P * 4
out["geom_losses(fwd+bwd)"] = dict(us=us, alg_MB=by / 1e6, GBps=by / us / 1e3, frac_of_hbm_peak=by / us / 1e3 / 8000, bound="hbm")
vm = torch.eye(4, device=dev); vm[3, 2] = 3.2
us = timed(lambda: LS.norm_from_depth(dep.detach(), vm, 0.36, 0.36))
by = (1 + 3 + 1) * P * 4
out["norm_from_depth"] = dict(us=us, alg_MB=by / 1e6, GBps=by / us / 1e3, frac_of_hbm_peak=by / us / 1e3 / 8000, bound="hbm")
print(json.dumps(out))
