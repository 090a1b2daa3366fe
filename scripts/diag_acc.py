"""Diagnostic: K7 accumulators (24 per Gaussian) vs oracle autograd w.r.t. the preprocess intermediates."""
import sys, os, math, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "texture-gs_amd"), os.path.join(ROOT, "tests")]
from texgs import synth
import helpers as Hh
from test_parity_gpu import CASES, _scene
from oracle import texgs_torch as O
from texgs.rasterizer import GaussianRasterizationSettings, forward_raw, backward_raw

ci = int(sys.argv[1]) if len(sys.argv) > 1 else 2
scene, cam, deg, bg = _scene(CASES[ci])
target, nhat = synth.make_targets(cam.image_height, cam.image_width, seed=5)
d = torch.float64
st = Hh.settings_for(cam, deg, bg)
cv = lambda t: t.to(d)
m3 = cv(scene.means3D).requires_grad_(True)
rq = lambda t: cv(t).requires_grad_(True)
pre = O.preprocess(m3, torch.zeros_like(m3), rq(scene.shs), cv(scene.opacities), rq(scene.scales), rq(scene.rotations),
                   cv(scene.uvs), cv(scene.gradient_uvs), st, d)
keys = ["xy", "conic", "opacity", "g", "G", "phi", "viewdep", "depth", "normal"]
pre["opacity"] = pre["opacity"].clone().requires_grad_(True)
pre["phi"] = pre["phi"].clone().requires_grad_(True)
for k in keys:
    if pre[k].requires_grad and not pre[k].is_leaf: pre[k].retain_grad()
binning = O.bin_and_sort(pre)
out, fT, nc, amb = O.render(pre, binning, cv(scene.texture), st, d)
L = synth.synthetic_loss(out[0:3], out[7:8], out[4:7], target.to(d), nhat.to(d)) + 0.05 * out[3:4].mean()
L.backward()
ref = torch.cat([pre["xy"].grad, pre["conic"].grad, pre["opacity"].grad[:, None], pre["g"].grad, pre["G"].grad.reshape(-1, 6),
                 pre["phi"].grad, pre["viewdep"].grad, pre["depth"].grad[:, None], pre["normal"].grad], 1)
dev = torch.device("cuda:0")
sth = Hh.settings_for(cam, deg, bg, device=dev, cls=GaussianRasterizationSettings)
t = lambda x: x.to(dev)
outs, s = forward_raw(sth, t(scene.means3D), t(scene.shs), t(scene.opacities), t(scene.scales), t(scene.rotations), t(scene.uvs), t(scene.gradient_uvs), t(scene.texture))
# upstream grads from the same loss
img = outs[0].clone().requires_grad_(True); dep = outs[1].clone().requires_grad_(True); nrm = outs[2].clone().requires_grad_(True); alp = outs[3].clone().requires_grad_(True)
Lh = synth.synthetic_loss(img, alp, nrm, target.to(dev), nhat.to(dev)) + 0.05 * dep.mean()
Lh.backward()
res = backward_raw(s, img.grad, dep.grad, nrm.grad, alp.grad)
acc = res[-1].cpu().double()
names = ["xy0","xy1","ca","cb","cc","op","g0","g1","G00","G01","G10","G11","G20","G21","phi0","phi1","phi2","vd0","vd1","vd2","depth","n0","n1","n2"]
for k, n in enumerate(names):
    e = (acc[:, k] - ref[:, k]).abs()
    i = int(torch.argmax(e))
    print(f"{n:6s} rel {float((acc[:,k]-ref[:,k]).norm()/ref[:,k].norm().clamp_min(1e-30)):.2e} max|ref| {float(ref[:,k].abs().max()):.2e} worst idx {i} got {float(acc[i,k]):.4e} ref {float(ref[i,k]):.4e}")
e = (acc - ref).abs().sum(1)
idx = torch.argsort(e, descending=True)[:5]
V = cam.world_view_transform.double(); Wr = V[:3, :3].t()
tt = (torch.cat([scene.means3D.double(), torch.ones(len(e), 1, dtype=d)], 1) @ V)[:, :3]
nv = pre["normal"].detach() @ Wr.t()
cosang = ((nv * tt).sum(1) / tt.norm(dim=1)).abs()
print("worst gaussians", idx.tolist(), "cos", cosang[idx].tolist(), "radius", pre["radius"][idx].tolist(), "|g|", pre["g"].detach().norm(dim=1)[idx].tolist())
