"""UV-Taylor producer.  PINNED: tests/golden/uvnet.npz holds weights, points, uvs and the Jacobian
computed by the reference's own UVNet class (nn.Linear form) and its autograd.functional.jacobian recipe
(tests/golden/make_golden.py).  CPU: the torch module of texgs.uvnet vs the golden.  GPU: the fused fp32-MFMA kernel vs the
golden and, at BASELINE's N = 300k, vs float64 torch autograd on a sample; non-default options (input normalisation,
bias-free = tiny-cuda-nn style) too."""
import os
import time

import numpy as np
import pytest
import torch

from texgs.uvnet import UVNet, jacobian_by_autograd
import helpers as Hh

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "uvnet.npz"))


def _golden_net(dtype=torch.float64):
    net = UVNet()
    net.load_state_dict({k.replace("__", "."): torch.tensor(G[k]) for k in G.files if "__" in k})
    return net.to(dtype), torch.tensor(G["emb"]).to(dtype), torch.tensor(G["xyz"]).to(dtype)


def test_torch_module_matches_reference_uvnet():
    net, emb, xyz = _golden_net()
    # the reference's forward casts its activations to float32 twice (`x = x.float()`, uv_net.py:31,35): 1e-7 apart from float64
    assert float((net(xyz, emb) - torch.tensor(G["uvs"])).abs().max()) < 5e-7
    J = jacobian_by_autograd(net, xyz, emb)
    assert float((J - torch.tensor(G["J"])).abs().max()) < 1e-5 * float(np.abs(G["J"]).max())
    assert float((net(xyz, emb).norm(dim=1) - 1).abs().max()) < 1e-12


@pytest.mark.gpu
def test_fused_kernel_matches_reference_golden(lib_built):
    net, emb, xyz = _golden_net(torch.float32)
    dev = torch.device("cuda:0")
    net = net.to(dev)
    uvs, J = net.uv_and_jacobian(xyz.to(dev), emb.to(dev))
    e_uv = float((uvs.cpu().double() - torch.tensor(G["uvs"])).abs().max())
    e_J = float((J.cpu().double() - torch.tensor(G["J"])).abs().max()) / float(np.abs(G["J"]).max())
    Hh.report("uv_taylor/golden", uvs_max_abs_err=e_uv, J_max_err_over_Jmax=e_J)
    assert e_uv < 2e-6 and e_J < 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["plain", "normalised_input", "no_bias"])
def test_fused_kernel_full_size_vs_float64_autograd(lib_built, variant):
    dev = torch.device("cuda:0")
    torch.manual_seed(11)
    kw = dict(xyz_offset=[0.1, -0.2, 0.05], xyz_scale=[1.5, 0.8, 1.2]) if variant == "normalised_input" else {}
    net = UVNet(**kw)
    if variant == "no_bias":
        for m in list(net.pre_mlp) + list(net.mlp):
            if isinstance(m, torch.nn.Linear):
                torch.nn.init.zeros_(m.bias)
    emb = torch.randn(128) * 0.2
    N = 300_000
    g = torch.Generator().manual_seed(1)
    xyz = torch.randn(N, 3, generator=g)
    xyz = xyz / xyz.norm(dim=1, keepdim=True) * (1 + 0.02 * torch.randn(N, 1, generator=g))
    netd = net.to(dev)
    uvs, J = netd.uv_and_jacobian(xyz.to(dev), emb.to(dev))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        netd.uv_and_jacobian(xyz.to(dev), emb.to(dev))
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    idx = torch.randperm(N, generator=g)[:4096]
    net64 = UVNet(**kw).double()
    net64.load_state_dict({k: v.double() for k, v in net.state_dict().items()})
    ref_uv = net64(xyz[idx].double(), emb.double())
    ref_J = jacobian_by_autograd(net64, xyz[idx].double(), emb.double())
    e_uv = float((uvs[idx.to(dev)].cpu().double() - ref_uv).abs().max())
    rel_J = float((J[idx.to(dev)].cpu().double() - ref_J).norm() / ref_J.norm())
    row = (J[idx.to(dev)].cpu().double() - ref_J).abs().amax(1) / ref_J.abs().amax(1).clamp_min(1e-12)
    Hh.report(f"uv_taylor/300k/{variant}", uvs_max_abs_err=e_uv, J_rel_l2=rel_J, J_worst_row_rel=float(row.max()),
              ms_per_call_incl_h2d=ms, gflop=N * 4 * (3 * 128 * 128 + 6 * 128) * 2 / 1e9)
    assert e_uv < 5e-6 and rel_J < 1e-4
    assert float((uvs.norm(dim=1) - 1).abs().max()) < 1e-5
    # [3*i+j] layout: J u = 0 would hold for J^T (d|uv|^2 = 0  =>  u^T J = 0), check the row index is the uv component
    uJ = torch.einsum("ni,nij->nj", uvs[idx.to(dev)].cpu().double(), J[idx.to(dev)].cpu().double().reshape(-1, 3, 3))
    assert float(uJ.abs().max()) < 1e-4 * float(ref_J.abs().max())
