"""Drop-in for the reference's `diff_gauss` extension module (imported at render/render.py:4, called :75-84): the
same tile rasterizer WITHOUT the texture -- colours from per-Gaussian SH (`shs f32[N,(deg+1)^2,3]`, DC first) or from
`colors_precomp f32[N,3]`, covariances from (scales, rotations) or from `cov3Ds_precomp f32[N,6]`
(render/render.py:52-53; layout of strip_lowerdiag, utils/general.py:73-82) -- returning the same 6-tuple
(image, depth, norm, alpha, radii, extra).

It runs the UNTEXTURED flavours of the textured operator's kernels (TexGSInputs.texture == NULL, texgs.h): no UV step, no
cubemap address, no taps, no texture-gradient machinery; the per-Gaussian colour enters through the operator's `color_offset`
input (C0*SH_DC, or colors_precomp - 0.5 so that max(0, offset + 0.5) is the given colour).  `means2D.grad[:, :2]` is the
lineage's dL/d(ndc xy) that stage-1 densification reads (models/gaussian3d.py:334-336).  With cov3Ds_precomp the splat normal
is the eigenvector of the smallest eigenvalue and carries no gradient (a selection, like the shortest-axis choice)."""
import torch
from torch import nn

from texgs.rasterizer import GaussianRasterizationSettings, _RasterizeGaussians, blend_extra_attrs  # noqa: F401

SH_C0 = 0.28209479177387814


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings: GaussianRasterizationSettings, grad_sink=None):
        super().__init__()
        self.raster_settings = raster_settings
        self.grad_sink = grad_sink          # optional texgs.multiview.GradBucket (fused multi-view accumulation; not reference API)

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3Ds_precomp=None, extra_attrs=None):
        st = self.raster_settings
        if (shs is None) == (colors_precomp is None):
            raise ValueError("Please provide exactly one of either SHs or precomputed colors!")
        if cov3Ds_precomp is not None:
            if scales is not None or rotations is not None:
                raise ValueError("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
        elif scales is None or rotations is None:
            raise ValueError("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
        if means2D is None:
            means2D = torch.zeros_like(means3D)
        rest = None
        if shs is not None:
            offset = SH_C0 * shs[:, 0, :]
            if shs.shape[1] > 1:
                rest = shs[:, 1:, :].contiguous()
        else:
            offset = colors_precomp - 0.5
        color, depth, norm, alpha, radii = _RasterizeGaussians.apply(
            means3D, means2D, rest, opacities, scales, rotations, None, None, None, st, offset.contiguous(), self.grad_sink,
            cov3Ds_precomp)
        extra = None
        if extra_attrs is not None:         # (always None in the reference, render/render.py:84; texgs.rasterizer.blend_extra_attrs)
            extra = blend_extra_attrs(st, means3D, means2D, opacities, scales, rotations, extra_attrs, self.grad_sink, cov3Ds_precomp)
        return color, depth, norm, alpha, radii, extra
