"""Test helper: call the operator the way the reference's render glue does -- written from the documented contract
(render/uv_tex_render.py:15-19 zero grad-carrier, :25-38 settings by keyword, :56-66 the ten
kwargs, :70-77 the returned dict), NOT a copy of that file -- plus a duck-typed stand-in for the model object whose
getters produce the operator inputs exactly like TextureGaussian3D's do (models/texture_gaussian3d.py:196-240:
activation OUTPUTS, not leaves; `get_grad_uvs` = detached Jacobian of the UV map in the [n, 3*i+j] layout).
"""
import math

import torch


class ToyTexturedGaussians:
    """Raw parameters are the leaves; the operator sees activation outputs (non-leaf tensors), as in the reference:
    scaling = exp(raw), rotation = normalize(raw), opacity = sigmoid(raw) with shape [N,1], uvs = uv_map(xyz)."""

    def __init__(self, scene, device, dtype, sh_degree):
        t = lambda x: x.detach().clone().to(device=device, dtype=dtype)
        leaf = lambda x: t(x).requires_grad_(True)
        self._xyz = leaf(scene.means3D)
        self._scaling = leaf(scene.scales.log())
        self._rotation = leaf(scene.rotations * 1.7)                       # un-normalised raw quaternion
        op = scene.opacities.clamp(1e-4, 1 - 1e-4)
        self._opacity = leaf(torch.log(op / (1 - op)))                      # inverse sigmoid, [N,1]
        self._shs = leaf(scene.shs)
        self._texture = leaf(scene.texture)
        g = torch.Generator().manual_seed(5)
        self.uv_A = leaf(torch.eye(3) + 0.2 * torch.randn(3, 3, generator=g))    # stands in for the UVNet weights
        self.uv_b = leaf(0.05 * torch.randn(3, generator=g))
        self.active_sh_degree = sh_degree

    def leaves(self):
        return dict(xyz=self._xyz, scaling=self._scaling, rotation=self._rotation, opacity=self._opacity,
                    shs=self._shs, texture=self._texture, uv_A=self.uv_A, uv_b=self.uv_b)

    def _uv_map(self, x):
        return torch.nn.functional.normalize(x @ self.uv_A + self.uv_b, dim=-1)

    get_xyz = property(lambda self: self._xyz)
    get_scaling = property(lambda self: torch.exp(self._scaling))
    get_rotation = property(lambda self: torch.nn.functional.normalize(self._rotation))
    get_opacity = property(lambda self: torch.sigmoid(self._opacity))
    get_shs = property(lambda self: self._shs)
    get_texture = property(lambda self: self._texture)
    get_uvs = property(lambda self: self._uv_map(self._xyz).contiguous())

    @property
    def get_grad_uvs(self):
        """d uv_i / d x_j per Gaussian via three backward passes of the column sums, flattened to [n, 3*i+j], detached
        (what models/texture_gaussian3d.py:216-227 produces)."""
        x = self._xyz.detach()
        jac = torch.autograd.functional.jacobian(lambda inp: self._uv_map(inp).sum(dim=0), x)      # [3(i), n, 3(j)]
        return jac.permute(1, 0, 2).reshape(-1, 9).contiguous().detach()


def render_like_reference_glue(rasterizer_module, viewpoint_camera, gaussians, bg_color, scaling_modifier=1.0,
                               debug=False, **extra_settings):
    """`rasterizer_module` exports GaussianRasterizationSettings / GaussianRasterizer (diff_gauss_uv_tex here; the torch
    oracle is wrapped to the same surface by the test)."""
    xyz = gaussians.get_xyz
    # zero grad-carrier: a NON-LEAF (zeros(requires_grad=True) + 0) whose .grad is kept with retain_grad()
    carrier = torch.zeros_like(xyz, requires_grad=True) + 0
    carrier.retain_grad()
    settings = rasterizer_module.GaussianRasterizationSettings(
        image_height=int(viewpoint_camera.image_height), image_width=int(viewpoint_camera.image_width),
        tanfovx=math.tan(viewpoint_camera.FoVx * 0.5), tanfovy=math.tan(viewpoint_camera.FoVy * 0.5),
        bg=bg_color, scale_modifier=scaling_modifier, viewmatrix=viewpoint_camera.world_view_transform,
        projmatrix=viewpoint_camera.full_proj_transform, sh_degree=getattr(gaussians, "active_sh_degree", 0),
        campos=viewpoint_camera.camera_center, prefiltered=False, debug=debug)
    rast = rasterizer_module.GaussianRasterizer(raster_settings=settings, **extra_settings)
    image, depth, norm, alpha, radii, extra = rast(
        means3D=xyz, means2D=carrier, shs=gaussians.get_shs, opacities=gaussians.get_opacity,
        scales=gaussians.get_scaling, rotations=gaussians.get_rotation, uvs=gaussians.get_uvs,
        gradient_uvs=gaussians.get_grad_uvs, texture=gaussians.get_texture, extra_attrs=None)
    return dict(render=image, depth=depth, norm=norm, alpha=alpha, viewspace_points=carrier,
                visibility_filter=radii > 0, extra=extra, radii=radii)
