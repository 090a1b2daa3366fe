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
    net = UVNet(precision="fp32")
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
    net = UVNet(precision="fp32", **kw)
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


@pytest.mark.gpu
def test_split_bf16_kernel_vs_f32_kernel_and_float64(lib_built):
    """precision="bf16x3" (csrc/uvnet.hip k_uv_taylor_bf16x3: every operand split into two bf16 halves, three bf16 MFMAs per product):
    uvs and the Jacobian within 2e-5 (relative to the largest entry) of the f32-MFMA kernel and of float64 autograd, at 300 k points,
    with the reference golden net too; a weight update re-packs (the two variants' packed buffers have different layouts)."""
    dev = torch.device("cuda:0")
    net0, emb0, xyz0 = _golden_net(torch.float32)
    netb = UVNet(precision="bf16x3")
    netb.load_state_dict(net0.state_dict())
    uvs, J = netb.to(dev).uv_and_jacobian(xyz0.to(dev), emb0.to(dev))
    e_uv = float((uvs.cpu().double() - torch.tensor(G["uvs"])).abs().max())
    e_J = float((J.cpu().double() - torch.tensor(G["J"])).abs().max()) / float(np.abs(G["J"]).max())
    Hh.report("uv_taylor_bf16x3/golden", uvs_max_abs_err=e_uv, J_max_err_over_Jmax=e_J)
    assert e_uv < 2e-5 and e_J < 1e-4
    torch.manual_seed(11)
    kw = dict(xyz_offset=[0.1, -0.2, 0.05], xyz_scale=[1.5, 0.8, 1.2])
    net = UVNet(precision="fp32", **kw)
    emb = torch.randn(128) * 0.2
    N = 300_000
    g = torch.Generator().manual_seed(1)
    xyz = torch.randn(N, 3, generator=g)
    xyz = (xyz / xyz.norm(dim=1, keepdim=True) * (1 + 0.02 * torch.randn(N, 1, generator=g))).to(dev)
    embd = emb.to(dev)
    n32 = net.to(dev)
    nb = UVNet(precision="bf16x3", **kw).to(dev)
    nb.load_state_dict(n32.state_dict())
    u32, J32 = n32.uv_and_jacobian(xyz, embd)
    ub, Jb = nb.uv_and_jacobian(xyz, embd)
    torch.cuda.synchronize()
    times = {}
    for name, m in (("fp32", n32), ("bf16x3", nb)):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            m.uv_and_jacobian(xyz, embd)
        e1.record()
        torch.cuda.synchronize()
        times[name] = e0.elapsed_time(e1) / 10 * 1e3
    jmax = float(J32.abs().max())
    d_uv = float((ub - u32).abs().max())
    # The Jacobian of a ReLU network is piecewise constant in its masks: a pre-activation within the arithmetic's error of zero may
    # take either side, and J then is that of the neighbouring linear region (the VALUE is continuous there).  With ~1e-5 relative
    # products a few of the 384 pre-activations of a few of the 300 000 points do: J is compared where the float64 network says no
    # pre-activation is that close, the rest is counted.
    idx = torch.randperm(N, generator=g)[:8192]
    net64 = UVNet(**kw).double()
    net64.load_state_dict({k: v.double().cpu() for k, v in n32.state_dict().items()})
    x64 = xyz[idx.to(dev)].cpu().double()
    ref_uv = net64(x64, emb.double())
    ref_J = jacobian_by_autograd(net64, x64, emb.double())
    with torch.no_grad():
        xn = net64._norm_in(x64)
        z1 = net64.pre_mlp[0](xn); h1 = z1.clamp_min(0)
        z2 = net64.pre_mlp[2](h1) + emb.double(); a2 = z2.clamp_min(0)
        z3 = net64.mlp[0](a2); h2 = z3.clamp_min(0)
        z4 = net64.mlp[2](h2)
        near = torch.stack([(z.abs() / z.abs().amax(dim=1, keepdim=True)).amin(dim=1) for z in (z1, z2, z3, z4)], 0).amin(0)
    safe = near > 1e-4          # (~10 % of the points have one of their 512 pre-activations that close to zero)
    Jb_s, J32_s = Jb[idx.to(dev)].cpu().double(), J32[idx.to(dev)].cpu().double()
    d_J = float((Jb_s - J32_s)[safe].abs().max()) / jmax
    e_uv = float((ub[idx.to(dev)].cpu().double() - ref_uv).abs().max())
    e_J = float((Jb_s - ref_J)[safe].abs().max()) / float(ref_J.abs().max())
    rel_J = float((Jb_s - ref_J)[safe].norm() / ref_J[safe].norm())
    rel_all = float((Jb - J32).norm() / J32.norm())
    Hh.report("uv_taylor_bf16x3/300k", uvs_max_abs_vs_f32_kernel=d_uv, J_max_over_Jmax_vs_f32_kernel_away_from_kinks=d_J, uvs_max_abs_vs_f64=e_uv,
              J_max_over_Jmax_vs_f64_away_from_kinks=e_J, J_rel_l2_vs_f64_away_from_kinks=rel_J, points_near_a_relu_kink_frac=float((~safe).float().mean()),
              J_rel_l2_vs_f32_kernel_all_points=rel_all, us_fp32=times["fp32"], us_bf16x3=times["bf16x3"])
    assert d_uv < 2e-5 and e_uv < 2e-5, (d_uv, e_uv)
    assert d_J < 5e-5 and e_J < 5e-5 and rel_J < 2e-5, (d_J, e_J, rel_J)
    assert float((~safe).float().mean()) < 0.15 and rel_all < 2e-2
    assert float((ub.norm(dim=1) - 1).abs().max()) < 1e-5
    with torch.no_grad():                        # a weight update re-packs in the variant's own layout
        nb.mlp[0].weight.mul_(1.5)
        n32.mlp[0].weight.mul_(1.5)
    ub2, _ = nb.uv_and_jacobian(xyz, embd)
    u32b, _ = n32.uv_and_jacobian(xyz, embd)
    assert float((ub2 - u32b).abs().max()) < 2e-5 and float((ub2 - ub).abs().max()) > 1e-3


@pytest.mark.gpu
def test_mixed_precision_kernel_keeps_value_column_exact(lib_built):
    """`UVNet(precision="mixed")` (k_uv_taylor_mixed): the value column runs on the f32-input MFMA exactly as in the f32 kernel, so
    uvs and the ReLU masks are that kernel's, and the Jacobian differs only by the split-bf16 arithmetic of the tangent columns,
    ~1e-5 relative, at EVERY point -- no kink exclusions as in the all-bf16 test; against the golden vector of the reference's own
    UVNet class; weight updates re-pack both layouts."""
    dev = torch.device("cuda:0")
    net0, emb0, xyz0 = _golden_net(torch.float32)
    netm = UVNet(precision="mixed")
    netm.load_state_dict(net0.state_dict())
    uvs, J = netm.to(dev).uv_and_jacobian(xyz0.to(dev), emb0.to(dev))
    e_uv = float((uvs.cpu().double() - torch.tensor(G["uvs"])).abs().max())
    e_J = float((J.cpu().double() - torch.tensor(G["J"])).abs().max()) / float(np.abs(G["J"]).max())
    Hh.report("uv_taylor_mixed/golden", uvs_max_abs_err=e_uv, J_max_err_over_Jmax=e_J)
    assert e_uv < 2e-6 and e_J < 1e-4
    torch.manual_seed(11)
    kw = dict(xyz_offset=[0.1, -0.2, 0.05], xyz_scale=[1.5, 0.8, 1.2])
    net = UVNet(precision="fp32", **kw)
    emb = torch.randn(128) * 0.2
    N = 300_000
    g = torch.Generator().manual_seed(1)
    xyz = torch.randn(N, 3, generator=g)
    xyz = (xyz / xyz.norm(dim=1, keepdim=True) * (1 + 0.02 * torch.randn(N, 1, generator=g))).to(dev)
    embd = emb.to(dev)
    n32 = net.to(dev)
    nm = UVNet(precision="mixed", **kw).to(dev)
    nm.load_state_dict(n32.state_dict())
    u32, J32 = n32.uv_and_jacobian(xyz, embd)
    um, Jm = nm.uv_and_jacobian(xyz, embd)
    torch.cuda.synchronize()
    times = {}
    for name, m in (("fp32", n32), ("mixed", nm)):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            m.uv_and_jacobian(xyz, embd)
        e1.record()
        torch.cuda.synchronize()
        times[name] = e0.elapsed_time(e1) / 10 * 1e3
    jmax = float(J32.abs().max())
    d_uv = float((um - u32).abs().max())
    same = float((um == u32).float().mean())
    dJ = (Jm - J32).abs().amax(dim=1) / jmax
    d_J_max = float(dJ.max())
    d_J_p9999 = float(torch.quantile(dJ[torch.randperm(N, device=dev)[:100_000]], 0.9999))
    rel_all = float((Jm - J32).norm() / J32.norm())
    Hh.report("uv_taylor_mixed/300k", uvs_max_abs_vs_f32_kernel=d_uv, uvs_bit_identical_frac=same, J_max_over_Jmax_vs_f32_kernel_all_points=d_J_max,
              J_p9999_over_Jmax=d_J_p9999, J_rel_l2_vs_f32_kernel_all_points=rel_all, us_fp32=times["fp32"], us_mixed=times["mixed"])
    # the first layer and the bias / embedding adds are written with explicit roundings shared by all kernels of csrc/uvnet.hip, the
    # 128x128 layers' value column is the same MFMA sequence: same pre-activations, same masks, same uvs (up to how the compiler
    # contracts the 128 -> 3 layer's sums); the Jacobian differs by the tangents' split-bf16 arithmetic only -- at EVERY point
    assert d_uv < 2.5e-7, d_uv
    assert d_J_max < 1e-4 and rel_all < 2e-5, (d_J_max, rel_all)
    with torch.no_grad():
        nm.mlp[0].weight.mul_(1.5)
        n32.mlp[0].weight.mul_(1.5)
    um2, _ = nm.uv_and_jacobian(xyz, embd)
    u32b, _ = n32.uv_and_jacobian(xyz, embd)
    assert float((um2 - u32b).abs().max()) < 1e-6 and float((um2 - um).abs().max()) > 1e-3


def test_tcnn_flat_params_round_trip_and_first_layer_bias():
    """tiny-cuda-nn FullyFusedMLP state (`use_tcnn: True`, every shipped config): one flat bias-free tensor per network,
    input padded to 16 columns with ones (-> first-layer bias), output padded to 16 rows.  Layout restated from tiny-cuda-nn's
    published source (unpinned: the package is absent here); this pins the loader against that statement and against the
    nn.Linear evaluation of the same weights."""
    from texgs.uvnet import unpack_tcnn_params, pack_tcnn_params, HIDDEN
    g = torch.Generator().manual_seed(5)
    pre = [(torch.randn(HIDDEN, 3, generator=g), torch.randn(HIDDEN, generator=g)), (torch.randn(HIDDEN, HIDDEN, generator=g) * 0.1, None)]
    mlp = [(torch.randn(HIDDEN, HIDDEN, generator=g) * 0.1, None), (torch.randn(HIDDEN, HIDDEN, generator=g) * 0.1, None),
           (torch.randn(3, HIDDEN, generator=g), None)]
    fp, fm = pack_tcnn_params(pre, 3, HIDDEN), pack_tcnn_params(mlp, HIDDEN, 3)
    assert fp.numel() == 128 * 16 + 128 * 128 and fm.numel() == 2 * 128 * 128 + 16 * 128          # the sizes tcnn reports for these nets
    for flat, layers, (i, o, h) in ((fp, pre, (3, HIDDEN, 1)), (fm, mlp, (HIDDEN, 3, 2))):
        for (w, b), (w2, b2) in zip(unpack_tcnn_params(flat.half(), i, o, h), layers):      # fp16 storage, as tcnn's native precision
            assert torch.allclose(w, w2, atol=2e-3, rtol=1e-3) and ((b is None) == (b2 is None))
            if b is not None:
                assert torch.allclose(b, b2, atol=2e-3, rtol=1e-3)
    with pytest.raises(ValueError, match="expected"):
        unpack_tcnn_params(fp[:-1], 3, HIDDEN, 1)
    # a module loaded from the flat form evaluates like one built from the matrices -- and says, loudly, that the layout it
    # assumed is unpinned
    with pytest.warns(RuntimeWarning, match="UNPINNED"):
        net = UVNet().load_reference_state({"pre_mlp.params": fp, "mlp.params": fm})
    assert net.tcnn_layout_unpinned is True
    assert UVNet().load_reference_state(UVNet().state_dict()).tcnn_layout_unpinned is False
    ref = UVNet()
    with torch.no_grad():
        for lin, (w, b) in zip(ref._linears(), pre + mlp):
            lin.weight.copy_(w)
            lin.bias.zero_() if b is None else lin.bias.copy_(b)
    x = torch.randn(50, 3, generator=g)
    emb = torch.randn(HIDDEN, generator=g) * 0.2
    assert torch.allclose(net(x, emb), ref(x, emb), atol=1e-6)
    # the ones-padding: column 3 of the first matrix IS the bias
    x16 = torch.cat([x, torch.ones(50, 13)], 1)
    assert torch.allclose(x16 @ fp[:128 * 16].reshape(128, 16).t(), ref.pre_mlp[0](x), atol=1e-5)


def test_tcnn_loader_rejects_detectable_mislayouts():
    """What the loader CAN tell without the real package: wrong padding (sizes that only a 16-padded layout
    gives), a flat tensor of another network shape, and -- on the reference-format checkpoint fixture -- that the nn.Linear form
    loads without the unpinned flag.  A transposed square layer cannot be detected from sizes; a transposed FIRST layer
    ([16, 128] read as [128, 16]) changes which column carries the ones-padding bias, which the round trip catches."""
    from texgs.uvnet import unpack_tcnn_params, pack_tcnn_params, HIDDEN
    g = torch.Generator().manual_seed(9)
    pre = [(torch.randn(HIDDEN, 3, generator=g), torch.randn(HIDDEN, generator=g)), (torch.randn(HIDDEN, HIDDEN, generator=g), None)]
    fp = pack_tcnn_params(pre, 3, HIDDEN)
    # unpadded input (3 columns instead of 16), unpadded output (3 rows instead of 16), one hidden layer too many / too few
    for bad in (torch.zeros(128 * 3 + 128 * 128), torch.zeros(128 * 16 + 128 * 128 + 3 * 128), fp[: 128 * 16], torch.cat([fp, fp])):
        with pytest.raises(ValueError, match="expected"):
            unpack_tcnn_params(bad, 3, HIDDEN, 1)
    with pytest.raises(ValueError, match="expected"):
        unpack_tcnn_params(torch.zeros(2 * 128 * 128 + 3 * 128), HIDDEN, 3, 2)            # output not padded to 16 rows
    # first layer stored transposed ([16, 128] row-major): same element count, so sizes cannot tell -- the bias column can:
    # reading it back gives a "bias" that is the sum of 13 random weight columns instead of the one stored column
    t = fp.clone()
    t[: 128 * 16] = fp[: 128 * 16].reshape(128, 16).t().reshape(-1)
    (w, b), _ = unpack_tcnn_params(t, 3, HIDDEN, 1)
    assert not torch.allclose(w, pre[0][0]) and not torch.allclose(b, pre[0][1])
    # the reference-format fixture (nn.Linear keys, written by the reference's own state_dict): no flag, no warning
    import os
    import warnings
    fx = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ckpt_stage3.pth")
    if os.path.exists(fx):
        ck = torch.load(fx, map_location="cpu", weights_only=False)
        sd = ck[0] if isinstance(ck, (tuple, list)) else ck
        uv = {k[len("uv_net."):]: v for k, v in sd.get("uv_net", sd).items() if isinstance(v, torch.Tensor) and k.startswith("uv_net.")} \
            if not any(k.startswith("pre_mlp") for k in sd.get("uv_net", {})) else sd["uv_net"]
        if uv and any(k.endswith("weight") for k in uv):
            with warnings.catch_warnings():
                warnings.simplefilter("error")
                assert UVNet().load_reference_state(uv).tcnn_layout_unpinned is False


@pytest.mark.parametrize("n", [200, 20_000])        # 20 000: the tall weight-gradient products take the chunked (batched) path
def test_manual_backward_matches_autograd_float64(n):
    """texgs.uvnet.uvnet_backward (the GEMM chain behind uvs_and_jacobian_with_grad) vs torch autograd of UVNet.forward."""
    from texgs.uvnet import uvnet_backward
    torch.manual_seed(2)
    net = UVNet(xyz_offset=[0.1, -0.2, 0.05], xyz_scale=[1.5, 0.8, 1.2]).double()
    emb = (torch.randn(128) * 0.2).double().requires_grad_(True)
    xyz = torch.randn(n, 3).double().requires_grad_(True)
    g = torch.randn(n, 3).double()
    (net(xyz, emb) * g).sum().backward()
    lins = net._linears()
    with torch.no_grad():
        xn = net._norm_in(xyz)
        dxn, demb, dW, db = uvnet_backward(xn, emb, [l.weight for l in lins], [l.bias for l in lins], g)
    assert torch.allclose(dxn / net.xyz_scale, xyz.grad, atol=1e-10)
    assert torch.allclose(demb, emb.grad, atol=1e-10 * max(1, n // 200))
    for l, w, b in zip(lins, dW, db):
        assert torch.allclose(w, l.weight.grad, atol=1e-9 * max(1, n // 200)) and torch.allclose(b, l.bias.grad, atol=1e-9 * max(1, n // 200))


def test_fused_autograd_node_routes_gradients(monkeypatch):
    """The autograd node around the two fused kernels (texgs.uvnet._FusedUV), with both kernels replaced by their plain-torch
    statements so that it runs on the CPU: which gradient goes to which input, d emb = db2 in the embedding's own shape, inputs
    that do not require grad get None, d xyz = J^T g -- against autograd of UVNet.forward."""
    from texgs.uvnet import uvnet_backward
    torch.manual_seed(3)
    n = 50
    net = UVNet(xyz_offset=[0.1, -0.2, 0.05], xyz_scale=[1.5, 0.8, 1.2]).double()

    def fake_forward(self, xyz, emb):
        with torch.enable_grad():
            return self.forward(xyz.detach(), emb.detach()).detach(), jacobian_by_autograd(self, xyz, emb.detach())

    def fake_backward(self, xyz, emb, g, params=None):
        lins = self._linears()
        _, _, dW, db = uvnet_backward(self._norm_in(xyz.detach()), emb.detach().reshape(-1), [l.weight.detach() for l in lins],
                                      [l.bias.detach() for l in lins], g)
        return [t for pair in zip(dW, db) for t in pair]
    monkeypatch.setattr(UVNet, "uv_and_jacobian", fake_forward)
    monkeypatch.setattr(UVNet, "backward_fused", fake_backward)
    g = torch.randn(n, 3).double()
    for frozen in ((), ("mlp.2.weight", "pre_mlp.0.bias"), ("emb",), ("xyz",)):
        for name, p_ in net.named_parameters():
            p_.requires_grad_(name not in frozen)
            p_.grad = None
        emb = (torch.randn(128) * 0.2).double().requires_grad_("emb" not in frozen)
        xyz = torch.randn(n, 3).double().requires_grad_("xyz" not in frozen)
        uvs, juv = net.uvs_and_jacobian_with_grad(xyz, emb)
        assert not juv.requires_grad
        (uvs * g).sum().backward()
        got = {name: (None if p_.grad is None else p_.grad.clone()) for name, p_ in net.named_parameters()}
        got["emb"], got["xyz"] = emb.grad, xyz.grad
        for p_ in net.parameters():
            p_.grad = None
        e2, x2 = emb.detach().clone().requires_grad_(emb.requires_grad), xyz.detach().clone().requires_grad_(xyz.requires_grad)
        (net(x2, e2) * g).sum().backward()
        exp = {name: p_.grad for name, p_ in net.named_parameters()}
        exp["emb"], exp["xyz"] = e2.grad, x2.grad
        for name in exp:
            if name in frozen:
                assert got[name] is None, (frozen, name)
            else:
                assert got[name] is not None and got[name].shape == exp[name].shape, (frozen, name)
                # (the node takes the upstream gradient in float32, as the kernel does: 6e-8 relative)
                assert torch.allclose(got[name], exp[name], rtol=1e-5, atol=1e-6), (frozen, name, float((got[name] - exp[name]).abs().max()))
    for p_ in net.parameters():
        p_.requires_grad_(True)


@pytest.mark.gpu
@pytest.mark.parametrize("n,bias,norm", [(1, True, False), (63, True, True), (64, False, False), (200, True, True), (16_385, False, True),
                                         (40_000, True, False)])
@pytest.mark.parametrize("precision", ["fp32", "mixed"])
def test_fused_backward_kernel_vs_float64(lib_built, n, bias, norm, precision):
    """csrc/uvnet.hip k_uv_backward (+ its reduction) through UVNet.backward_fused against the plain-torch chain `uvnet_backward`
    in float64: every weight / bias gradient within 1e-4 relative L2 (fp32 MFMA, sums over up to 40 000 points in f32).  Sizes: a
    single point, one short of a tile, exactly one tile, a ragged tail, one point past 256 tiles (the persistent grid wraps: a
    workgroup owns two tiles), 625 tiles.  bias=False is the tiny-cuda-nn form (NULL bias pointers are not exercised through the
    module -- zero biases are; the C ABI's NULL handling is covered by the forward's tests sharing the same loads)."""
    from texgs.uvnet import uvnet_backward
    dev = torch.device("cuda:0")
    torch.manual_seed(11 + n)
    kw = dict(xyz_offset=[0.1, -0.2, 0.05], xyz_scale=[1.5, 0.8, 1.2]) if norm else {}
    net = UVNet(precision=precision, **kw)       # "mixed": the backward chain's six GEMMs as split-bf16 products (texgs_uv_backward_mixed)
    if not bias:
        with torch.no_grad():
            for lin in net._linears():
                lin.bias.zero_()
    emb = torch.randn(128) * 0.2
    lins = net._linears()
    # A ReLU mask is decided by the sign of a pre-activation: one within the f32 arithmetic's error of zero may take either side, and
    # that unit's gradient for that point is then all or nothing (first GPU run: ONE flipped mask among 16 385 x 512 put 4e-4 into
    # every gradient below it).  The kernel is compared on points the float64 network says have no pre-activation that close.
    cand = torch.randn(n + n // 4 + 64, 3)
    with torch.no_grad():
        net.double()
        xc = net._norm_in(cand.double())
        z1 = lins[0](xc); z2 = lins[1](z1.clamp_min(0)) + emb.double(); z3 = lins[2](z2.clamp_min(0)); z4 = lins[3](z3.clamp_min(0))
        near = torch.stack([(z.abs() / z.abs().amax(dim=1, keepdim=True)).amin(dim=1) for z in (z1, z2, z3, z4)], 0).amin(0)
    safe = torch.nonzero(near > 1e-5)[:, 0]
    assert safe.numel() >= n, (safe.numel(), n)
    xyz = cand[safe[:n]].contiguous()
    g = torch.randn(n, 3)
    with torch.no_grad():
        xn = net._norm_in(xyz.double())
        _, demb, dW, db = uvnet_backward(xn, emb.double(), [l.weight for l in lins], [l.bias for l in lins], g.double())
    net = net.float().to(dev)
    got = net.backward_fused(xyz.to(dev), emb.to(dev), g.to(dev))
    torch.cuda.synchronize()
    errs = {}
    for k in range(5):
        errs[f"W{k + 1}"] = Hh.rel_err(got[2 * k].cpu(), dW[k])
        errs[f"b{k + 1}"] = Hh.rel_err(got[2 * k + 1].cpu(), db[k])
    errs["emb"] = Hh.rel_err(got[3].cpu(), demb)
    Hh.report(f"uv_backward/{precision}/n{n}/bias{int(bias)}/norm{int(norm)}", **errs)
    assert max(errs.values()) < 1e-4, errs
    # deterministic: partial sums are added in workgroup order, no atomics
    again = net.backward_fused(xyz.to(dev), emb.to(dev), g.to(dev))
    assert all(torch.equal(a_, b_) for a_, b_ in zip(got, again))


@pytest.mark.gpu
def test_fused_backward_no_points(lib_built):
    dev = torch.device("cuda:0")
    net = UVNet(precision="fp32").to(dev)
    got = net.backward_fused(torch.zeros(0, 3, device=dev), torch.zeros(128, device=dev), torch.zeros(0, 3, device=dev))
    assert all(float(t.abs().max()) == 0.0 for t in got)
    with pytest.raises(RuntimeError):
        UVNet().backward_fused(torch.zeros(4, 3), torch.zeros(128), torch.zeros(4, 3))        # no CPU fallback


@pytest.mark.gpu
def test_fused_forward_with_gradients(lib_built):
    """uvs_and_jacobian_with_grad: one fused launch forward, gradients to xyz (= J^T g), the embedding and all weights vs torch
    autograd of the module's plain forward (float64 on the CPU)."""
    dev = torch.device("cuda:0")
    torch.manual_seed(4)
    net = UVNet(precision="fp32", xyz_offset=[0.1, -0.2, 0.05], xyz_scale=[1.5, 0.8, 1.2])
    emb0 = torch.randn(128) * 0.2
    N = 20000
    xyz0 = torch.randn(N, 3)
    xyz0 = xyz0 / xyz0.norm(dim=1, keepdim=True)
    g = torch.randn(N, 3)
    net64 = UVNet(xyz_offset=[0.1, -0.2, 0.05], xyz_scale=[1.5, 0.8, 1.2]).double()
    net64.load_state_dict({k: v.double() for k, v in net.state_dict().items()})
    e64, x64 = emb0.double().requires_grad_(True), xyz0.double().requires_grad_(True)
    (net64(x64, e64) * g.double()).sum().backward()
    netd = net.to(dev)
    emb, xyz = emb0.to(dev).requires_grad_(True), xyz0.to(dev).requires_grad_(True)
    uvs, juv = netd.uvs_and_jacobian_with_grad(xyz, emb)
    assert not juv.requires_grad and uvs.requires_grad
    (uvs * g.to(dev)).sum().backward()
    errs = dict(xyz=Hh.rel_err(xyz.grad.cpu(), x64.grad), emb=Hh.rel_err(emb.grad.cpu(), e64.grad))
    for k, (l, l64) in enumerate(zip(netd._linears(), net64._linears())):
        errs[f"W{k + 1}"] = Hh.rel_err(l.weight.grad.cpu(), l64.weight.grad)
        errs[f"b{k + 1}"] = Hh.rel_err(l.bias.grad.cpu(), l64.bias.grad)
    Hh.report("uv_taylor/with_grad/20k", **errs)
    assert max(errs.values()) < 2e-4, errs
    # the packed-weight cache (one buffer per (device, stream)): a second call re-uses it, an in-place weight update re-packs it in
    # place, a side stream gets a buffer of its own, invalidate_packed() forgets them all
    slot = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
    key, buf = netd._packed[slot]
    netd.uv_and_jacobian(xyz, emb)
    assert netd._packed[slot][0] == key and netd._packed[slot][1] is buf
    with torch.no_grad():
        netd.mlp[0].weight.mul_(1.01)
    u2, _ = netd.uv_and_jacobian(xyz, emb)
    assert netd._packed[slot][0] != key and netd._packed[slot][1] is buf and float((u2 - uvs.detach()).abs().max()) > 1e-6
    side = torch.cuda.Stream(dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        u3, _ = netd.uv_and_jacobian(xyz, emb)
    side.synchronize()
    assert len(netd._packed) == 2 and torch.equal(u3, u2)
    netd.invalidate_packed()
    assert netd._packed == {}


@pytest.mark.gpu
def test_mixed_jacobian_renders_the_same_image_at_c3(lib_built):
    """The DEFAULT precision of the fused UV map is "mixed" (round 6): uvs -- and with them the anchor of every texture sample -- are
    bit-identical to the f32 kernel's, only the Jacobian's tangent columns run split-bf16 (~1e-5 relative).  What that does to the
    operator's output: the C3 scene rendered with the mixed-precision J of a UV map fitted to x / |x| (as the iteration bench fits
    it; an untrained MLP has a Jacobian tens of times larger and sends all Gaussians to a few texels) against the same render with
    the f32 J.  J only moves a texture sample INSIDE a splat, by ~1e-4 texel: depth / normals / alpha / radii are identical, and EVERY
    pixel of the image agrees within 1e-4 (the north star's tolerance; white-noise texture, the worst case for a displaced
    sample)."""
    from texgs import synth
    from texgs.rasterizer import GaussianRasterizationSettings, forward_raw
    dev = torch.device("cuda:0")
    assert UVNet().precision == "mixed"
    torch.manual_seed(3)
    scene = synth.make_scene(300_000, 1024, seed=0)
    xyz = scene.means3D.to(dev)
    fit = UVNet(precision="fp32").to(dev)
    emb = (torch.randn(128) * 0.2).to(dev).requires_grad_(True)
    opt = torch.optim.Adam(list(fit.parameters()) + [emb], lr=2e-3)
    for _ in range(400):
        x = xyz[torch.randint(0, xyz.shape[0], (16384,), device=dev)]
        loss = (1.0 - (fit(x, emb) * torch.nn.functional.normalize(x, dim=1)).sum(1)).mean()
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
    emb = emb.detach()
    nets = {"fp32": fit, "mixed": UVNet(precision="mixed").to(dev)}
    nets["mixed"].load_state_dict(fit.state_dict())
    fit.invalidate_packed()
    cam = synth.fibonacci_cameras(64, 800, 800)[0]
    st = Hh.settings_for(cam, 3, torch.zeros(3), device=dev, cls=GaussianRasterizationSettings)
    outs = {}
    for p, net in nets.items():
        uvs, J = net.uv_and_jacobian(xyz, emb)
        outs[p] = (uvs, J) + forward_raw(st, xyz, scene.shs.to(dev), scene.opacities.to(dev).reshape(-1), scene.scales.to(dev),
                                         scene.rotations.to(dev), uvs, J, scene.texture.to(dev), for_backward=False)[0]
    torch.cuda.synchronize()
    a, b = outs["fp32"], outs["mixed"]
    assert torch.equal(a[0], b[0])                                          # uvs
    jerr = float((a[1] - b[1]).abs().max() / a[1].abs().max())
    diff = (a[2] - b[2]).abs().amax(dim=0)
    over = float((diff > 1e-4).float().mean())
    Hh.report("uv_taylor_mixed/c3_image", J_max_err_over_Jmax=jerr, image_max_abs_diff=float(diff.max()), pixels_over_1e4_frac=over,
              image_rms_diff=float((a[2] - b[2]).pow(2).mean().sqrt()))
    assert 0.0 < jerr < 1e-4
    assert float(diff.max()) < 1e-4 and over == 0.0                        # measured on MI355X: max 8.8e-5, rms 2.9e-6
    assert float((a[2] - b[2]).pow(2).mean().sqrt()) < 1e-5
    for k in (3, 4, 5, 6):                                                  # depth, normals, alpha, radii: no J in them
        assert torch.equal(a[k], b[k])
