"""Drop-in for the reference's `diff_gauss_uv_tex` extension module (imported at render/uv_tex_render.py:4).

Put the directory that holds this package (texture-gs_amd/) on PYTHONPATH and the reference's
render/uv_tex_render.py runs unchanged against the MI355X-native rasterizer."""
from texgs.rasterizer import GaussianRasterizationSettings, GaussianRasterizer  # noqa: F401

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer"]
