"""UV-Taylor producer: the operator inputs `uvs` = phi(mu) and `gradient_uvs` = d phi / d x.

`UVNet` is the reference's UV MLP (models/modules/uv_net.py:9-36) in its nn.Linear form (the fallback of
models/modules/utils.py:43-61, `use_tcnn: False`): same module names (`pre_mlp`, `mlp` = nn.Sequential of Linear / ReLU),
so a reference state_dict of that form loads directly; `forward` is plain torch (differentiable, for the training graph
of `uvs`).  `uv_and_jacobian` runs the fused HIP kernel (csrc/uvnet.hip, fp32 MFMA): phi and its analytic 3x3 Jacobian
by forward-mode propagation in one launch, instead of UVNet.forward plus the three backward passes of
torch.autograd.functional.jacobian (models/texture_gaussian3d.py:216-236).
"""
import ctypes as C

import torch
from torch import nn
import torch.nn.functional as F

from . import _lib

HIDDEN = 128


def _mlp(n_hidden, in_dims, out_dims, width=HIDDEN):
    mods, ch = [], in_dims
    for _ in range(n_hidden):
        mods += [nn.Linear(ch, width), nn.ReLU()]
        ch = width
    mods.append(nn.Linear(ch, out_dims))
    return nn.Sequential(*mods)


class UVNet(nn.Module):
    """pre_mlp: 3 -> 128 -> emb_dim(128); relu(. + emb); mlp: 128 -> 128 -> 128 -> 3; F.normalize  (configs/*.yaml uv_net_cfg)."""

    def __init__(self, xyz_offset=None, xyz_scale=None):
        super().__init__()
        self.pre_mlp = _mlp(1, 3, HIDDEN)
        self.mlp = _mlp(2, HIDDEN, 3)
        self.xyz_offset = None if xyz_offset is None else torch.as_tensor(xyz_offset, dtype=torch.float32)
        self.xyz_scale = None if xyz_scale is None else torch.as_tensor(xyz_scale, dtype=torch.float32)

    def forward(self, xyz, emb):
        if self.xyz_offset is not None and self.xyz_scale is not None:
            xyz = (xyz - self.xyz_offset.to(xyz)) / self.xyz_scale.to(xyz)
        x = F.relu(self.pre_mlp(xyz) + emb[None, :])
        return F.normalize(self.mlp(x), dim=-1)

    @torch.no_grad()
    def uv_and_jacobian(self, xyz, emb):
        """(uvs f32[N,3], gradient_uvs f32[N,9] with [3*i+j] = d uv_i / d x_j) from the fused HIP kernel.  No autograd graph
        (the reference detaches the Jacobian too, models/texture_gaussian3d.py:227; use forward() for a differentiable uvs)."""
        lib = _lib.load()
        dev = xyz.device
        if dev.type != "cuda":
            raise RuntimeError("UVNet.uv_and_jacobian runs on an AMD GPU; there is no CPU fallback (use forward + autograd)")
        f = lambda t: None if t is None else t.detach().to(device=dev, dtype=torch.float32).contiguous()
        x = f(xyz)
        N = x.shape[0]
        ws = [f(self.pre_mlp[0].weight), f(self.pre_mlp[0].bias), f(self.pre_mlp[2].weight), f(self.pre_mlp[2].bias), f(emb),
              f(self.mlp[0].weight), f(self.mlp[0].bias), f(self.mlp[2].weight), f(self.mlp[2].bias),
              f(self.mlp[4].weight), f(self.mlp[4].bias), f(self.xyz_offset), f(self.xyz_scale)]
        if ws[0].shape != (HIDDEN, 3) or ws[2].shape != (HIDDEN, HIDDEN) or ws[9].shape != (3, HIDDEN) or ws[4].numel() != HIDDEN:
            raise ValueError("the fused kernel supports the shipped UVNet shape only (3-128-128 | 128-128-128-3, emb 128)")
        p = lambda t: None if t is None else t.data_ptr()
        net = _lib.UVNetStruct(*[p(t) for t in ws], HIDDEN)
        uvs = torch.empty(N, 3, dtype=torch.float32, device=dev)
        juv = torch.empty(N, 9, dtype=torch.float32, device=dev)
        temp = torch.empty(lib.texgs_uv_taylor_temp_bytes(), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            _lib.check(lib.texgs_uv_taylor(C.byref(net), p(x), N, p(uvs), p(juv), p(temp), torch.cuda.current_stream(dev).cuda_stream),
                       "texgs_uv_taylor")
        return uvs, juv


def jacobian_by_autograd(net: UVNet, xyz, emb):
    """The reference's way (models/texture_gaussian3d.py:216-227): Jacobian of the column sums, three backward passes."""
    jac = torch.autograd.functional.jacobian(lambda inp: net(inp, emb).sum(dim=0), xyz.detach())
    return jac.permute(1, 0, 2).reshape(-1, 9).contiguous().detach()
