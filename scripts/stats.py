"""Diagnostic: per-view K7 statistics (needs a -DTEXGS_STATS build of render.hip)."""
import sys, os, ctypes as C, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "texture-gs_amd"), os.path.join(ROOT, "tests")]
from texgs import synth, _lib
import helpers as Hh
from texgs.rasterizer import GaussianRasterizationSettings, forward_raw, backward_raw
N, R, W, H = 300000, 1024, 800, 800
scene = synth.make_scene(N, R, seed=0)
cam = synth.fibonacci_cameras(64, W, H)[0]
dev = torch.device("cuda:0")
st = Hh.settings_for(cam, 3, torch.zeros(3), device=dev, cls=GaussianRasterizationSettings)
t = lambda x: x.to(dev)
outs, s = forward_raw(st, t(scene.means3D), t(scene.shs), t(scene.opacities), t(scene.scales), t(scene.rotations), t(scene.uvs), t(scene.gradient_uvs), t(scene.texture))
lib = _lib.load()
buf = (C.c_ulonglong * 16)()
lib.texgs_debug_stats(buf, 1)
g = torch.Generator().manual_seed(1)
dout = torch.randn(8, H, W, generator=g).to(dev) / (H * W)
res = backward_raw(s, dout[0:3].contiguous(), None, dout[4:7].contiguous(), dout[7:8].contiguous())
torch.cuda.synchronize()
lib.texgs_debug_stats(buf, 1)
names = ["tests after wave cull", "full tests (exp)", "items (pairs)", "rounds", "segments", "cache tap hits", "cache tap misses", "stage C heavy iters", "wave-chunks"]
for n, v in zip(names, buf):
    print(f"{n:26s} {v:>12d}")
print("D", s.D, "wave-instances", s.D * 4)
