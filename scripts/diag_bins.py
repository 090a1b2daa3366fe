"""Distribution of texture-gradient list lengths per bin at C3 (needs a -DTEXGS_EXPERIMENTS library and TEXGS_SKIP_REDUCE=1)."""
import sys, os, math
sys.path[:0] = ['/root/repo', '/root/repo/texture-gs_amd', '/root/repo/tests']
import torch
from texgs import synth, rasterizer as RZ
from texgs.rasterizer import GaussianRasterizationSettings, forward_raw, backward_raw
dev = torch.device('cuda:0')
N, R, W, H = 300_000, 1024, 800, 800
scene = synth.make_scene(N, R, seed=0)
cam = synth.fibonacci_cameras(64, W, H)[int(os.environ.get("VIEW", "0"))]
st = GaussianRasterizationSettings(H, W, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), torch.zeros(3, device=dev), 1.0,
                                   cam.world_view_transform.to(dev), cam.full_proj_transform.to(dev), 3, cam.camera_center.to(dev), False, False)
t = lambda x: x.to(dev)
RZ.TEX_BIN_CAP = 1 << 20
RZ.TEX_BIN_BYTES_MAX = 64 << 30
outs, s = forward_raw(st, t(scene.means3D), t(scene.shs), t(scene.opacities), t(scene.scales), t(scene.rotations), t(scene.uvs),
                      t(scene.gradient_uvs), t(scene.texture))
g = torch.Generator().manual_seed(1)
P = W * H
gi = ((torch.rand(3, H, W, generator=g) > 0.5).float() * 2 - 1).to(dev) / (3 * P)
res = backward_raw(s, gi, None, None, None)
torch.cuda.synchronize()
bins = list(RZ._TEX_BINS.values())[0]
c = bins.cursor[:bins.nbins].to(torch.int64).cpu()
print("cap", bins.cap, "nbins", bins.nbins, "total records", int(c.sum()), "nonempty bins", int((c > 0).sum()))
srt = torch.sort(c, descending=True).values
print("top 16:", srt[:16].tolist())
for q in (0.5, 0.9, 0.99, 0.999):
    print("quantile", q, int(torch.quantile(srt[srt > 0].double(), q)))
nb = 32
top = torch.topk(c, 8).indices.tolist()
print("top bins (face, by, bx):", [(b // (nb * nb), (b // nb) % nb, b % nb) for b in top])
