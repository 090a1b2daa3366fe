#!/usr/bin/env python3
"""The UV map's kernels alone at N = 300 000 (for rocprofv3): fused forward + Jacobian (fp32 and split-bf16), fused backward."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "texture-gs_amd")):
    sys.path.insert(0, p)
from texgs.uvnet import UVNet      # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = torch.device("cuda:0")
torch.manual_seed(0)
N = 300_000
xyz = torch.nn.functional.normalize(torch.randn(N, 3, device=dev), dim=1)
g = torch.randn(N, 3, device=dev) / N
emb = (0.2 * torch.randn(128)).to(dev)
for prec in ("fp32", "bf16x3", "mixed"):
    net = UVNet(precision=prec).to(dev)
    for _ in range(reps):
        net.uv_and_jacobian(xyz, emb)
ev = lambda: torch.cuda.Event(enable_timing=True)
out = {"N": N}
grads = {}
for prec in ("fp32", "mixed"):          # (the backward of "bf16x3" is the mixed one)
    net = UVNet(precision=prec).to(dev)
    torch.manual_seed(1)
    for lin in net._linears():
        torch.nn.init.normal_(lin.weight, std=0.1)
    grads[prec] = net.backward_fused(xyz, emb, g)
    torch.cuda.synchronize()
    e0, e1 = ev(), ev()
    e0.record()
    for _ in range(reps):
        net.backward_fused(xyz, emb, g)
    e1.record()
    torch.cuda.synchronize()
    out[f"uv_backward_{prec}_us"] = round(1e3 * e0.elapsed_time(e1) / reps, 1)
out["mixed_vs_fp32_rel_l2"] = [round(float((a - b).double().norm() / b.double().norm()), 8) for a, b in zip(grads["mixed"], grads["fp32"])]
print(out)
