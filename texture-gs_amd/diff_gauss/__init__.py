"""Drop-in for the reference's `diff_gauss` extension module (imported at render/render.py:4, called :75-84): the
same tile rasterizer WITHOUT the texture -- colours from per-Gaussian SH (`shs f32[N,(deg+1)^2,3]`, DC first) or from
`colors_precomp f32[N,3]` -- returning the same 6-tuple (image, depth, norm, alpha, radii, extra).

It runs on the textured operator's kernels: a 1x1 zero cubemap makes the texture term vanish and the per-Gaussian
colour enters through the operator's `color_offset` input (C0*SH_DC, or colors_precomp - 0.5 so that
max(0, offset + 0.5) is the given colour).  `means2D.grad[:, :2]` is the lineage's dL/d(ndc xy) that stage-1
densification reads (models/gaussian3d.py:334-336).  SURVEY.md section 8f-1: built for import compatibility and
stages 1-2; not tuned (the texture machinery idles)."""
import torch
from torch import nn

from texgs.rasterizer import GaussianRasterizationSettings, _RasterizeGaussians  # noqa: F401

SH_C0 = 0.28209479177387814


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings: GaussianRasterizationSettings):
        super().__init__()
        self.raster_settings = raster_settings

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3Ds_precomp=None, extra_attrs=None):
        st = self.raster_settings
        if (shs is None) == (colors_precomp is None):
            raise ValueError("Please provide exactly one of either SHs or precomputed colors!")
        if cov3Ds_precomp is not None:
            raise NotImplementedError("cov3Ds_precomp (cfg.compute_cov3D_python, render/render.py:52-53) is not built; "
                                      "pass scales and rotations")
        if scales is None or rotations is None:
            raise ValueError("Please provide scales and rotations")
        if extra_attrs is not None:
            raise NotImplementedError("extra_attrs blending is not built yet")
        N = means3D.shape[0]
        dev = means3D.device
        if means2D is None:
            means2D = torch.zeros_like(means3D)
        rest = None
        if shs is not None:
            offset = SH_C0 * shs[:, 0, :]
            if shs.shape[1] > 1:
                rest = shs[:, 1:, :].contiguous()
        else:
            offset = colors_precomp - 0.5
        uvs = torch.zeros(N, 3, device=dev)
        uvs[:, 2] = 1.0
        juv = torch.zeros(N, 9, device=dev)
        tex = torch.zeros(6, 1, 1, 3, device=dev)
        color, depth, norm, alpha, radii = _RasterizeGaussians.apply(
            means3D, means2D, rest, opacities, scales, rotations, uvs, juv, tex, st, offset.contiguous())
        return color, depth, norm, alpha, radii, None
