"""Diagnostic: per-Gaussian record (K1) vs oracle preprocess; per-Gaussian grad errors vs grazing angle."""
import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "texture-gs_amd"), os.path.join(ROOT, "tests")]
from texgs import synth
import helpers as Hh
from test_parity_gpu import CASES, _scene
from oracle import texgs_torch as O

case = CASES[0]
scene, cam, deg, bg = _scene(case)
target, nhat = synth.make_targets(cam.image_height, cam.image_width, seed=5)
ref, dbg, gref = Hh.oracle_run(scene, cam, deg, bg, with_grad=True, target=target, nhat=nhat, depth_weight=0.05)
outs, s = Hh.hip_debug_state(scene, cam, deg, bg)
rec = s.tensors["rec"].cpu().double()
pre = dbg["pre"]
vis = pre["valid"]
def cmp(name, got, exp):
    d = (got - exp.detach()).abs()[vis]
    print(f"{name:8s} max abs {float(d.max()):.3e}  rel-to-max {float(d.max()/exp.detach().abs()[vis].max()):.3e}")
cmp("xy", rec[:, 0:2], pre["xy"]); cmp("conic", rec[:, 2:5], pre["conic"]); cmp("g", rec[:, 6:8], pre["g"])
cmp("G", rec[:, 8:14], pre["G"].reshape(-1, 6)); cmp("phi", rec[:, 14:17], pre["phi"]); cmp("vd", rec[:, 17:20], pre["viewdep"])
cmp("depth", rec[:, 20], pre["depth"]); cmp("normal", rec[:, 21:24], pre["normal"])
# grazing measure
V = cam.world_view_transform.double(); Wr = V[:3, :3].t()
t = (torch.cat([scene.means3D.double(), torch.ones(len(vis), 1, dtype=torch.float64)], 1) @ V)[:, :3]
nv = pre["normal"].detach() @ Wr.t()
cosang = ((nv * t).sum(1) / t.norm(dim=1)).abs()
print("cos grazing quantiles", torch.quantile(cosang[vis], torch.tensor([0.0, 0.001, 0.01, 0.1, 0.5], dtype=torch.float64)))
gerr = (rec[:, 6:8] - pre["g"].detach()).abs().sum(1) / pre["g"].detach().abs().sum(1).clamp_min(1e-12)
idx = torch.argsort(gerr, descending=True)[:8]
print("worst g rel err", gerr[idx], "cos", cosang[idx])
out, ggot = Hh.hip_run(scene, cam, deg, bg, with_grad=True, target=target, nhat=nhat, depth_weight=0.05)
for n in ["means3D", "rotations", "means2D"]:
    e = (ggot[n].double() - gref[n]).abs().sum(1)
    idx = torch.argsort(e, descending=True)[:6]
    print(n, "worst abs err", e[idx].tolist(), "ref mag", gref[n].abs().sum(1)[idx].tolist(), "cos", cosang[idx].tolist())
# pixels with image error
err = (out[0].detach().cpu().double() - ref[0].double()).abs().max(0).values
ys, xs = torch.nonzero(err > 1e-4, as_tuple=True)
print("bad pixels", len(ys), list(zip(ys.tolist(), xs.tolist()))[:12])
