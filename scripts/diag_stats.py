"""Diagnostic: workload statistics of one C3 view (tile list lengths, contributors, texel footprint per tile)."""
import sys, os, math, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "texture-gs_amd"), os.path.join(ROOT, "tests")]
from texgs import synth
import helpers as Hh
from texgs.rasterizer import GaussianRasterizationSettings, forward_raw, backward_raw
N, R, W, H = 300000, 1024, 800, 800
scene = synth.make_scene(N, R, seed=0)
cam = synth.fibonacci_cameras(64, W, H)[0]
dev = torch.device("cuda:0")
st = Hh.settings_for(cam, 3, torch.zeros(3), device=dev, cls=GaussianRasterizationSettings)
t = lambda x: x.to(dev)
outs, s = forward_raw(st, t(scene.means3D), t(scene.shs), t(scene.opacities), t(scene.scales), t(scene.rotations), t(scene.uvs), t(scene.gradient_uvs), t(scene.texture))
torch.cuda.synchronize()
rg = s.tensors["ranges"].cpu().long(); cnt = rg[:, 1] - rg[:, 0]
q = torch.tensor([0.5, 0.9, 0.99, 1.0])
print("D", s.D, "tiles", len(cnt), "nonempty", int((cnt > 0).sum()), "list len quantiles", torch.quantile(cnt[cnt > 0].float(), q).tolist(), "mean", float(cnt[cnt>0].float().mean()))
nc = s.tensors["n_contrib"].cpu().long()
print("n_contrib quantiles (covered px)", torch.quantile(nc[nc > 0].float(), q).tolist(), "covered px", int((nc > 0).sum()))
tmax = nc.reshape(50, 16, 50, 16).amax(dim=(1, 3)).reshape(-1)
print("tile max n_contrib / list len: mean ratio", float((tmax[cnt>0].float() / cnt[cnt>0].float()).mean()))
print("alpha mean", float(outs[3].mean()), "final_T<1e-3 frac", float((s.tensors["final_T"] < 1e-3).float().mean()))
vis = int((outs[4] > 0).sum()); print("visible", vis, "radii mean", float(outs[4][outs[4] > 0].float().mean()))
