"""Seeded synthetic scenes and cameras for tests / bench.

The scene is a thin shell of flattened discs around the unit sphere (the reference's texture space is the
unit sphere, reference README.md:57; discs flattened like reset_min_scale, models/texture_gaussian3d.py:290-297).
Cameras restate utils/graphics.py:38-71 + utils/cameras.py:56-65 (row-vector / transposed matrices); the
restatement is pinned against the importable reference functions by tests/golden/cameras.npz.
Everything is generated on CPU with a torch.Generator so CPU oracle and GPU runs see identical inputs.
"""
import math
from typing import NamedTuple

import numpy as np
import torch


class Scene(NamedTuple):
    means3D: torch.Tensor      # [N,3]
    scales: torch.Tensor       # [N,3]  (already exp-activated)
    rotations: torch.Tensor    # [N,4]  unit (w,x,y,z)
    opacities: torch.Tensor    # [N,1]  (already sigmoid-activated)
    shs: torch.Tensor          # [N,15,3]
    uvs: torch.Tensor          # [N,3] unit
    gradient_uvs: torch.Tensor # [N,9]
    texture: torch.Tensor      # [6,R,R,3]


class Cam(NamedTuple):
    image_height: int
    image_width: int
    FoVx: float
    FoVy: float
    world_view_transform: torch.Tensor   # [4,4] transposed (row-vector) form
    full_proj_transform: torch.Tensor    # [4,4]
    camera_center: torch.Tensor          # [3]


def _quat_mul(a, b):
    w1, x1, y1, z1 = a.unbind(-1)
    w2, x2, y2, z2 = b.unbind(-1)
    return torch.stack([w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2,
                        w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                        w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2,
                        w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2], -1)


def make_scene(N, R, seed=0, scale_mean=0.006, sh_coeffs=15, random_jacobian=False):
    """random_jacobian=True replaces the analytic (symmetric) Jacobian of `normalize` by a general non-symmetric 3x3 per
    Gaussian -- what a trained UVNet produces (models/texture_gaussian3d.py:216-227) -- so that a transposed [3*i+j]
    layout cannot hide behind J = J^T."""
    g = torch.Generator().manual_seed(seed)
    rn = lambda *s: torch.randn(*s, generator=g, dtype=torch.float64)
    d = rn(N, 3)
    d = d / d.norm(dim=1, keepdim=True)
    r = 1.0 + 0.02 * rn(N, 1)
    means = d * r
    ls = math.log(scale_mean) + 0.3 * rn(N, 2)
    scales = torch.cat([ls.exp(), torch.full((N, 1), math.exp(-20.0), dtype=torch.float64)], 1)
    # quaternion taking local z to dir: axis = z x dir, angle = acos(z.dir)
    z = torch.tensor([0.0, 0.0, 1.0], dtype=torch.float64)
    axis = torch.cross(z.expand(N, 3), d, dim=1)
    an = axis.norm(dim=1, keepdim=True)
    axis = torch.where(an > 1e-9, axis / an.clamp_min(1e-30), torch.tensor([1.0, 0.0, 0.0], dtype=torch.float64).expand(N, 3))
    ang = torch.acos(d[:, 2:3].clamp(-1, 1))
    q_align = torch.cat([torch.cos(ang / 2), axis * torch.sin(ang / 2)], 1)
    spin = 2 * math.pi * torch.rand(N, 1, generator=g, dtype=torch.float64)
    q_spin = torch.cat([torch.cos(spin / 2), torch.zeros(N, 2, dtype=torch.float64), torch.sin(spin / 2)], 1)
    q = _quat_mul(q_align, q_spin)
    jitter = math.radians(5.0) * rn(N, 3)                   # ~5 degree tilt
    jn = jitter.norm(dim=1, keepdim=True).clamp_min(1e-12)
    q_j = torch.cat([torch.cos(jn / 2), jitter / jn * torch.sin(jn / 2)], 1)
    q = _quat_mul(q_j, q)
    q = q / q.norm(dim=1, keepdim=True)
    opac = torch.sigmoid(2.0 + rn(N, 1))
    uvs = d.clone()
    eye = torch.eye(3, dtype=torch.float64)
    Jm = (eye[None] - d[:, :, None] * d[:, None, :]) / r[:, :, None]   # analytic Jacobian of normalize at mu
    shs = 0.1 * rn(N, sh_coeffs, 3)
    tex = torch.randn(6, R, R, 3, generator=g, dtype=torch.float32)
    if random_jacobian:                      # drawn last: every other tensor is identical to the symmetric-J scene
        g2 = torch.Generator().manual_seed(seed + 7919)
        Jm = Jm + 0.3 * torch.randn(N, 3, 3, generator=g2, dtype=torch.float64) / r[:, :, None]
    f = lambda t: t.to(torch.float32).contiguous()
    return Scene(f(means), f(scales), f(q), f(opac), f(shs), f(uvs), f(Jm.reshape(N, 9)), tex)


def band_limited_texture(R, seed=0, period=32, amplitude=1.0):
    """A cubemap of LOW-PASSED noise: white noise drawn every `period` texels, bicubically interpolated to [6,R,R,3] -- slope
    ~ amplitude / period per texel (0.03-0.05 at the default) and a continuous first derivative, against the O(1) texel-to-texel
    steps of make_scene's white-noise texture.  Parity runs use it beside the white noise: a bilinear cell chosen differently by
    two fp32 implementations then moves a pair's dL/duv by a few per cent instead of by O(1), so that every gradient row can
    be compared at the plain tolerance (no cell-edge flags), and the sample itself is insensitive to the last bit of the texel
    coordinate (RGB within the literal 1e-4 at R = 2048 too)."""
    g = torch.Generator().manual_seed(seed)
    n = max(R // period, 1) + 3
    coarse = amplitude * torch.randn(6 * 3, 1, n, n, generator=g, dtype=torch.float32)
    fine = torch.nn.functional.interpolate(coarse, size=(R + 2 * period, R + 2 * period), mode="bicubic", align_corners=True)
    fine = fine[:, 0, period:period + R, period:period + R]            # away from the interpolation's clamped border
    return fine.reshape(6, 3, R, R).permute(0, 2, 3, 1).contiguous()


def world2view(Rm, t):
    """utils/graphics.py:38-49 with translate=0, scale=1 (the inverse-of-inverse is the identity there)."""
    Rt = np.zeros((4, 4))
    Rt[:3, :3] = Rm.transpose()
    Rt[:3, 3] = t
    Rt[3, 3] = 1.0
    return np.float32(Rt)


def projection(znear, zfar, fovX, fovY):
    """utils/graphics.py:51-71."""
    tx, ty = math.tan(fovX / 2), math.tan(fovY / 2)
    top, right = ty * znear, tx * znear
    P = torch.zeros(4, 4)
    P[0, 0] = 2.0 * znear / (2 * right)
    P[1, 1] = 2.0 * znear / (2 * top)
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


def look_at_camera(eye, W, H, fovx=0.6911, fovy=None, up=(0.0, -1.0, 0.0), znear=0.01, zfar=100.0):
    """Camera at `eye` looking at the origin.  R is camera-to-world (columns = camera axes) and
    T = -R^T eye, the (R, T) convention of utils/cameras.py:22-26."""
    eye = np.asarray(eye, dtype=np.float64)
    fwd = -eye / np.linalg.norm(eye)                   # camera +z looks at the origin
    upv = np.asarray(up, dtype=np.float64)
    right = np.cross(upv, fwd)
    if np.linalg.norm(right) < 1e-6:
        right = np.cross(np.array([1.0, 0.0, 0.0]), fwd)
    right /= np.linalg.norm(right)
    down = np.cross(fwd, right)
    Rm = np.stack([right, down, fwd], axis=1)          # c2w rotation
    T = -Rm.T @ eye
    if fovy is None:
        fovy = 2 * math.atan(math.tan(fovx / 2) * H / W)
    wvt = torch.tensor(world2view(Rm, T)).transpose(0, 1).contiguous()
    proj = projection(znear, zfar, fovx, fovy).transpose(0, 1)
    full = (wvt.unsqueeze(0).bmm(proj.unsqueeze(0))).squeeze(0).contiguous()
    center = wvt.inverse()[3, :3].contiguous()
    return Cam(H, W, fovx, fovy, wvt, full, center)


def fibonacci_cameras(V, W, H, dist=3.2, fovx=0.6911):
    cams = []
    ga = math.pi * (3.0 - math.sqrt(5.0))
    for i in range(V):
        y = 1.0 - 2.0 * (i + 0.5) / V
        rad = math.sqrt(max(0.0, 1.0 - y * y))
        th = ga * i
        eye = dist * np.array([math.cos(th) * rad, y, math.sin(th) * rad])
        cams.append(look_at_camera(eye, W, H, fovx=fovx))
    return cams


def make_targets(H, W, seed=1):
    g = torch.Generator().manual_seed(seed)
    target = torch.rand(3, H, W, generator=g)
    nhat = torch.randn(3, H, W, generator=g)
    nhat = nhat / nhat.norm(dim=0, keepdim=True)
    return target, nhat


def synthetic_loss(image, alpha, norm, target, nhat):
    """non-zero upstream grads into image, alpha and norm (the outputs the reference
    differentiates, models/texture_gaussian3d.py:333-368)."""
    return (image - target).abs().mean() + (alpha - 1.0).abs().mean() + 0.1 * (1.0 - (norm * nhat).sum(0)).mean()
