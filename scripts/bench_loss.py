"""Timing: fused HIP loss front-end vs the torch formulation the reference runs (losses/ssim_loss.py), 800x800, fwd+bwd."""
import sys, os, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "texture-gs_amd")]
from texgs.losses import rgb_alpha_loss
from oracle import losses_torch as LO
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
H = W = 800
img = torch.rand(3, H, W, generator=g).to(dev).requires_grad_(True)
gt = torch.rand(3, H, W, generator=g).to(dev)
al = torch.rand(1, H, W, generator=g).to(dev).requires_grad_(True)
ga = (torch.rand(1, H, W, generator=g) > 0.5).float().to(dev)
def run(fn, n=50):
    for _ in range(5):
        img.grad = None; al.grad = None; fn().backward()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        img.grad = None; al.grad = None; fn().backward()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
t_hip = run(lambda: rgb_alpha_loss(img, gt, al, ga, 0.2, 0.1))
t_torch = run(lambda: LO.rgb_alpha_loss(img, gt, al, ga, 0.2, 0.1))
bytes_alg = (3 * 2 + 3 + 1 * 3) * H * W * 4          # read I, Igt; write dI; read A, Agt; write dA
print(f"fused HIP loss fwd+bwd: {t_hip:.1f} us  | torch (reference formulation, fp32 on the same GPU): {t_torch:.1f} us  | "
      f"algorithmic {bytes_alg/1e6:.1f} MB -> {bytes_alg/t_hip/1e3:.1f} GB/s")
