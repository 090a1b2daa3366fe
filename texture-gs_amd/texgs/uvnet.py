"""UV-Taylor producer: the operator inputs `uvs` = phi(mu) and `gradient_uvs` = d phi / d x.

`UVNet` is the reference's UV MLP (models/modules/uv_net.py:9-36): 3 -> 128 -> emb_dim(128); relu(. + emb); 128 -> 128 -> 128 -> 3;
F.normalize.  It holds its weights as nn.Linear modules under the reference's module names (`pre_mlp`, `mlp` = nn.Sequential of
Linear / ReLU, models/modules/utils.py:43-54), so a reference state_dict of the `use_tcnn: False` form loads directly, and
`load_reference_state` also accepts the form every SHIPPED config produces (`use_tcnn: True`, configs/*.yaml): tiny-cuda-nn
FullyFusedMLP networks, whose state_dict is ONE flat `params` tensor per network (see `unpack_tcnn_params`).

Three ways to evaluate:
  forward(xyz, emb)                       plain torch, differentiable (any device)
  uv_and_jacobian(xyz, emb)               the fused HIP kernel (csrc/uvnet.hip, fp32 MFMA): phi and its analytic 3x3 Jacobian by
                                          forward-mode propagation in one launch, instead of UVNet.forward plus the three backward
                                          passes of torch.autograd.functional.jacobian (models/texture_gaussian3d.py:216-236)
  uvs_and_jacobian_with_grad(xyz, emb)    the same launch inside an autograd node: `uvs` carries gradients to xyz (= J^T g, free:
                                          J is there), to the embedding and to every weight (ONE fused HIP backward kernel that
                                          recomputes the activations per tile, `backward_fused`), so the training graph of `uvs`
                                          stays on the device with one fused forward and one fused backward instead of a second,
                                          torch-side evaluation of the network and autograd's per-layer passes.
"""
import ctypes as C

import torch
from torch import nn
import torch.nn.functional as F

from . import _lib

HIDDEN = 128
TCNN_ALIGN = 16          # tiny-cuda-nn pads input / output widths of FullyFusedMLP to multiples of 16


def _mlp(n_hidden, in_dims, out_dims, width=HIDDEN, bias=True):
    mods, ch = [], in_dims
    for _ in range(n_hidden):
        mods += [nn.Linear(ch, width, bias=bias), nn.ReLU()]
        ch = width
    mods.append(nn.Linear(ch, out_dims, bias=bias))
    return nn.Sequential(*mods)


def unpack_tcnn_params(params, n_in, n_out, n_hidden_layers, width=HIDDEN):
    """The flat `params` tensor of a tiny-cuda-nn `Network(n_in, n_out, FullyFusedMLP, n_neurons=width, n_hidden_layers)` ->
    [(weight [out, in], bias or None)] per layer, nn.Linear convention.

    Layout restated from tiny-cuda-nn's published source (the package is an un-vendored, unpinned dependency of the reference,
    requirements.txt; it cannot be imported here, so this is UNPINNED against the real thing): FullyFusedMLP stores its weight
    matrices back to back, each ROW-major [fan_out, fan_in] -- first [width, pad16(n_in)], then n_hidden_layers - 1 matrices
    [width, width], last [pad16(n_out), width] -- no biases.  `tcnn.Network` feeds the MLP through an identity encoding that
    pads the input to 16 columns WITH ONES, so the weights of the padded input columns act as a learned first-layer bias
    (returned as that layer's bias); padded output rows are dropped.  fp16 or fp32 storage is accepted, math is float32."""
    p = params.detach().reshape(-1).to(torch.float32)
    pin, pout = -(-n_in // TCNN_ALIGN) * TCNN_ALIGN, -(-n_out // TCNN_ALIGN) * TCNN_ALIGN
    shapes = [(width, pin)] + [(width, width)] * (n_hidden_layers - 1) + [(pout, width)]
    need = sum(a * b for a, b in shapes)
    if p.numel() != need:
        raise ValueError(f"tcnn params: expected {need} values for {n_in}->{n_out} with {n_hidden_layers} hidden layer(s) of {width}, "
                         f"got {p.numel()}")
    mats, off = [], 0
    for a, b in shapes:
        mats.append(p[off:off + a * b].reshape(a, b))
        off += a * b
    first = mats[0]
    layers = [(first[:, :n_in].contiguous(), first[:, n_in:].sum(dim=1) if pin > n_in else None)]
    layers += [(m.contiguous(), None) for m in mats[1:-1]]
    layers.append((mats[-1][:n_out].contiguous(), None))
    return layers


def pack_tcnn_params(layers, n_in, n_out, width=HIDDEN):
    """Inverse of unpack_tcnn_params (fixtures / tests): the first layer's bias goes to the first padded input column."""
    pin, pout = -(-n_in // TCNN_ALIGN) * TCNN_ALIGN, -(-n_out // TCNN_ALIGN) * TCNN_ALIGN
    out = []
    for k, (w, b) in enumerate(layers):
        w = w.detach().to(torch.float32)
        if k == 0:
            m = torch.zeros(width, pin)
            m[:, :n_in] = w
            if b is not None:
                if pin == n_in:
                    raise ValueError("no padded input column to carry the first-layer bias")
                m[:, n_in] = b
        elif k == len(layers) - 1:
            m = torch.zeros(pout, width)
            m[:n_out] = w
        else:
            m = w
        out.append(m.reshape(-1))
    return torch.cat(out)


class UVNet(nn.Module):
    """pre_mlp: 3 -> 128 -> emb_dim(128); relu(. + emb); mlp: 128 -> 128 -> 128 -> 3; F.normalize  (configs/*.yaml uv_net_cfg)."""

    def __init__(self, xyz_offset=None, xyz_scale=None, precision="mixed"):
        super().__init__()
        # arithmetic of the fused kernel's three 128x128 layers:
        # "mixed" (the default since round 6) = the VALUE column on the f32-input MFMA -- uvs and every ReLU mask are exactly the
        #   "fp32" kernel's, so the rasterizer's discrete decisions (cube face, bilinear cell) do not move --, the three TANGENT
        #   columns (3/4 of the work) split-bf16: J within ~1.4e-5 of J_max at every point, the C3 image within 1e-4 of the one
        #   rendered with the f32 J (tests/test_uvnet.py::test_mixed_jacobian_renders_the_same_image_at_c3); ~1.6x faster;
        # "fp32" = f32-input MFMA throughout (exact f32 products; what the float64 parity tests check);
        # "bf16x3" = every operand split into two bf16 halves, three bf16 MFMAs per product (~2.5x faster, uvs / J within ~2e-5; the
        #   ReLU masks then come from the split value column: near a kink the forward is that of the neighbouring linear region and
        #   the fused backward -- which recomputes its masks in f32 -- differentiates a slightly different function: inference only)
        if precision not in ("fp32", "bf16x3", "mixed"):
            raise ValueError("precision must be 'fp32', 'bf16x3' or 'mixed'")
        self.precision = precision
        self.pre_mlp = _mlp(1, 3, HIDDEN)
        self.mlp = _mlp(2, HIDDEN, 3)
        # device-resident (buffers follow .to() / .cuda()): the reference moves them to the device on every call (uv_net.py:23-24)
        self.register_buffer("xyz_offset", None if xyz_offset is None else torch.as_tensor(xyz_offset, dtype=torch.float32), persistent=False)
        self.register_buffer("xyz_scale", None if xyz_scale is None else torch.as_tensor(xyz_scale, dtype=torch.float32), persistent=False)
        # W2..W4 in MFMA operand order, ONE buffer per (device, stream) that evaluates the net: a buffer is written and read on
        # its own stream only, so no event is needed between the pack and the evaluation, and a re-pack never frees memory that
        # another stream may still be reading (ViewPipeline runs views on several streams).  Key = (data_ptr, _version) of the
        # three weights: updates through `.data` (p.data.copy_(), some legacy optimizers) do NOT bump _version -- call
        # invalidate_packed() after such an update.
        self._packed = {}
        self.tcnn_layout_unpinned = False       # set by load_reference_state when the weights came from tiny-cuda-nn's flat layout

    def invalidate_packed(self):
        """Forget the MFMA-ordered weight copies (after changing weights in a way autograd's version counters do not see)."""
        self._packed = {}

    # ---- weights --------------------------------------------------------------------------------------------------------
    def _linears(self):
        return [self.pre_mlp[0], self.pre_mlp[2], self.mlp[0], self.mlp[2], self.mlp[4]]

    def load_reference_state(self, state):
        """A reference `uv_net.state_dict()` in either form: nn.Linear keys (`pre_mlp.0.weight`, ...) or tiny-cuda-nn keys
        (`pre_mlp.params`, `mlp.params`: flat, bias-free, 16-padded -- every shipped config)."""
        if "pre_mlp.params" in state and "mlp.params" in state:
            import warnings
            pre = unpack_tcnn_params(state["pre_mlp.params"], 3, HIDDEN, 1)
            mlp = unpack_tcnn_params(state["mlp.params"], HIDDEN, 3, 2)
            with torch.no_grad():
                for lin, (w, b) in zip(self._linears(), pre + mlp):
                    lin.weight.copy_(w)
                    lin.bias.zero_() if b is None else lin.bias.copy_(b)
            self.tcnn_layout_unpinned = True
            warnings.warn("UVNet weights were read from tiny-cuda-nn's flat FullyFusedMLP parameter tensor.  That layout is restated "
                          "from the published source and is UNPINNED here (tiny-cuda-nn cannot be installed in the build container, "
                          "no fixture produced by it exists): check a rendered view against the reference before trusting the "
                          "checkpoint (module.tcnn_layout_unpinned is set).", RuntimeWarning, stacklevel=2)
            self.invalidate_packed()
            return self
        self.load_state_dict(state)
        self.tcnn_layout_unpinned = False
        self.invalidate_packed()
        return self

    def _norm_in(self, xyz):
        # both or neither, as uv_net.py:22-25
        if self.xyz_offset is not None and self.xyz_scale is not None:
            return (xyz - self.xyz_offset.to(xyz)) / self.xyz_scale.to(xyz)
        return xyz

    def forward(self, xyz, emb):
        x = F.relu(self.pre_mlp(self._norm_in(xyz)) + emb[None, :])
        return F.normalize(self.mlp(x), dim=-1)

    # ---- fused kernel ---------------------------------------------------------------------------------------------------
    def _kernel_args(self, dev, emb, params=None):
        """`params` = [W1, b1, ..., W5, b5] to use INSTEAD of the module's current weights (the autograd node hands its saved
        tensors: a weight rebound through `.data` between forward and backward must not change what the backward differentiates)."""
        f = lambda t: None if t is None else t.detach().to(device=dev, dtype=torch.float32).contiguous()
        both = self.xyz_offset is not None and self.xyz_scale is not None
        if params is None:
            params = [t for lin in self._linears() for t in (lin.weight, lin.bias)]
        W1, b1, W2, b2, W3, b3, W4, b4, W5, b5 = params
        ws = [f(W1), f(b1), f(W2), f(b2), f(emb), f(W3), f(b3), f(W4), f(b4), f(W5), f(b5),
              f(self.xyz_offset) if both else None, f(self.xyz_scale) if both else None]
        if ws[0].shape != (HIDDEN, 3) or ws[2].shape != (HIDDEN, HIDDEN) or ws[9].shape != (3, HIDDEN) or ws[4].numel() != HIDDEN:
            raise ValueError("the fused kernel supports the shipped UVNet shape only (3-128-128 | 128-128-128-3, emb 128)")
        return ws

    @torch.no_grad()
    def uv_and_jacobian(self, xyz, emb):
        """(uvs f32[N,3], gradient_uvs f32[N,9] with [3*i+j] = d uv_i / d x_j) from the fused HIP kernel.  No autograd graph
        (the reference detaches the Jacobian too, models/texture_gaussian3d.py:227; see uvs_and_jacobian_with_grad).  The
        MFMA-ordered copy of W2..W4 is cached on the module and re-made only when one of those weights changed."""
        lib = _lib.load()
        dev = xyz.device
        if dev.type != "cuda":
            raise RuntimeError("UVNet.uv_and_jacobian runs on an AMD GPU; there is no CPU fallback (use forward + autograd)")
        x = xyz.detach().to(dtype=torch.float32).contiguous()
        N = x.shape[0]
        ws = self._kernel_args(dev, emb)
        p = lambda t: None if t is None else t.data_ptr()
        net = _lib.UVNetStruct(*[p(t) for t in ws], HIDDEN)
        stream = torch.cuda.current_stream(dev).cuda_stream
        split = self.precision
        pack_fn, eval_fn = {"fp32": (lib.texgs_uv_pack, lib.texgs_uv_taylor_packed),
                            "bf16x3": (lib.texgs_uv_pack_bf16x3, lib.texgs_uv_taylor_packed_bf16x3),
                            "mixed": (lib.texgs_uv_pack_mixed, lib.texgs_uv_taylor_packed_mixed)}[split]
        key = tuple((t.data_ptr(), t._version) for t in (self.pre_mlp[2].weight, self.mlp[0].weight, self.mlp[2].weight)) + (dev, split)
        uvs = torch.empty(N, 3, dtype=torch.float32, device=dev)
        juv = torch.empty(N, 9, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            slot = (dev.index, int(stream))
            ent = self._packed.get(slot)
            if ent is None or ent[0] != key:
                buf = ent[1] if ent is not None else torch.empty(2 * lib.texgs_uv_taylor_temp_bytes(), dtype=torch.uint8, device=dev)   # ("mixed" holds both layouts)
                _lib.check(pack_fn(C.byref(net), p(buf), stream), "texgs_uv_pack")       # (re-packed in place: same stream, in order)
                self._packed[slot] = ent = (key, buf)
            _lib.check(eval_fn(C.byref(net), p(ent[1]), p(x), N, p(uvs), p(juv), stream), "texgs_uv_taylor_packed")
        return uvs, juv

    @torch.no_grad()
    def backward_fused(self, xyz, emb, g, params=None):
        """[dW1, db1, dW2, db2, dW3, db3, dW4, db4, dW5, db5] of sum(uvs * g) from the fused HIP backward (csrc/uvnet.hip
        k_uv_backward: one persistent kernel, activations recomputed per tile in LDS, weight gradients in registers; d emb = db2).
        What autograd does through models/modules/uv_net.py:19-36 under loss.backward() in the reference.  No CPU fallback
        (`uvnet_backward` below is the plain-torch statement the tests check this kernel against)."""
        lib = _lib.load()
        dev = xyz.device
        if dev.type != "cuda":
            raise RuntimeError("UVNet.backward_fused runs on an AMD GPU; there is no CPU fallback")
        x = xyz.detach().to(dtype=torch.float32).contiguous()
        gg = g.detach().to(dtype=torch.float32).contiguous()
        N = x.shape[0]
        if gg.shape != (N, 3):
            raise ValueError(f"g must be [N, 3], got {tuple(gg.shape)}")
        ws = self._kernel_args(dev, emb, params)
        p = lambda t: None if t is None else t.data_ptr()
        netp = _lib.UVNetStruct(*[p(t) for t in ws], HIDDEN)
        shapes = [(HIDDEN, 3), (HIDDEN,), (HIDDEN, HIDDEN), (HIDDEN,), (HIDDEN, HIDDEN), (HIDDEN,), (HIDDEN, HIDDEN), (HIDDEN,),
                  (3, HIDDEN), (3,)]
        outs = [torch.empty(sh, dtype=torch.float32, device=dev) for sh in shapes]
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            temp = torch.empty(lib.texgs_uv_backward_temp_bytes(N), dtype=torch.uint8, device=dev)
            gr = _lib.UVNetGradStruct(*[p(t) for t in outs])
            # precision "fp32": all nine GEMMs on the f32-input MFMA; "mixed" / "bf16x3": the forward recomputation in f32 (the
            # forward launch's ReLU masks bit for bit), the six GEMMs of the backward chain as split-bf16 products (~1e-5 relative)
            fn = lib.texgs_uv_backward if self.precision == "fp32" else lib.texgs_uv_backward_mixed
            _lib.check(fn(C.byref(netp), p(x), p(gg), N, C.byref(gr), p(temp), stream), "texgs_uv_backward")
        return outs

    def uvs_and_jacobian_with_grad(self, xyz, emb):
        """(uvs, gradient_uvs) from ONE fused launch, with `uvs` differentiable w.r.t. xyz, emb and the network's weights
        (`gradient_uvs` carries no gradient, as in the reference)."""
        lins = self._linears()
        params = [t for lin in lins for t in (lin.weight, lin.bias)]
        uvs, juv = _FusedUV.apply(self, xyz, emb, *params)
        return uvs, juv


def _tn(a, b, chunk=2048):
    """a^T b for tall a [N, m], b [N, n] (weight gradients: N = 300 000 points, m, n <= 128).  As ONE GEMM this is a 128 x 128
    output with K = N: the library runs it on a handful of CUs (measured: the five of them were ~8 of the 10.5 ms of an iteration's
    backward).  As a batch of N / chunk products summed afterwards it fills the chip."""
    N = a.shape[0]
    if N < 8 * chunk:
        return a.t() @ b
    B = N // chunk
    main = B * chunk
    out = torch.bmm(a[:main].view(B, chunk, a.shape[1]).transpose(1, 2), b[:main].view(B, chunk, b.shape[1])).sum(0)
    if main < N:
        out = out + a[main:].t() @ b[main:]
    return out


def uvnet_backward(xn, emb, weights, biases, g, inv_scale=None):
    """Gradients of uvs = normalize(MLP(xn)) w.r.t. (xn, emb, weights, biases) for an upstream gradient g [N,3], by hand: the
    activations are recomputed with five GEMMs, the chain rule is six more.  Pure torch (runs anywhere; tests run it in
    float64 on the CPU against autograd).  xn = the (already offset / scaled) input; returns (d_xn, d_emb, [dW], [db])."""
    W1, W2, W3, W4, W5 = weights
    b1, b2, b3, b4, b5 = [None if b is None else b for b in biases]
    add = lambda z, b: z if b is None else z + b
    z1 = add(xn @ W1.t(), b1); h1 = z1.clamp_min(0)
    z2 = add(h1 @ W2.t(), b2) + emb; a = z2.clamp_min(0)
    z3 = add(a @ W3.t(), b3); h2 = z3.clamp_min(0)
    z4 = add(h2 @ W4.t(), b4); h3 = z4.clamp_min(0)
    o = add(h3 @ W5.t(), b5)
    n = o.norm(dim=1, keepdim=True).clamp_min(1e-12)
    u = o / n
    do = (g - u * (u * g).sum(dim=1, keepdim=True)) / n                 # F.normalize backward
    dW5 = _tn(do, h3); db5 = do.sum(0)
    d4 = (do @ W5) * (z4 > 0)
    dW4 = _tn(d4, h2); db4 = d4.sum(0)
    d3 = (d4 @ W4) * (z3 > 0)
    dW3 = _tn(d3, a); db3 = d3.sum(0)
    d2 = (d3 @ W3) * (z2 > 0)
    dW2 = _tn(d2, h1); db2 = d2.sum(0); demb = db2
    d1 = (d2 @ W2) * (z1 > 0)
    dW1 = _tn(d1, xn); db1 = d1.sum(0)
    dxn = d1 @ W1
    return dxn, demb, [dW1, dW2, dW3, dW4, dW5], [db1, db2, db3, db4, db5]


class _FusedUV(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net, xyz, emb, *params):
        uvs, juv = net.uv_and_jacobian(xyz, emb)
        ctx.net = net
        ctx.save_for_backward(xyz, emb, juv, *params)
        ctx.mark_non_differentiable(juv)
        return uvs, juv

    @staticmethod
    def backward(ctx, g, _gj):
        xyz, emb, juv, *params = ctx.saved_tensors
        net = ctx.net
        need = ctx.needs_input_grad           # (net, xyz, emb, W1, b1, ..., W5, b5)
        g = g.to(torch.float32).contiguous()
        d_xyz = None
        if need[1]:                           # d uv_i / d x_j = J[3 i + j]: d_xyz_j = sum_i g_i J_ij  -- J is already there
            d_xyz = (g[:, :, None] * juv.reshape(-1, 3, 3)).sum(dim=1)      # (elementwise: as an einsum this is 300 000 batched 1x3 . 3x3 products)
        d_emb, d_params = None, [None] * 10
        if any(need[2:]):
            grads = net.backward_fused(xyz, emb, g, params=params)        # (the forward's weights: ADVICE r5)
            for k in range(10):
                d_params[k] = grads[k] if need[3 + k] else None
            if need[2]:
                d_emb = (grads[3].clone() if need[6] else grads[3]).reshape(emb.shape)        # d emb = db2 (the embedding is added where b2 is)
        return (None, d_xyz, d_emb, *d_params)


def jacobian_by_autograd(net: UVNet, xyz, emb):
    """The reference's way (models/texture_gaussian3d.py:216-227): Jacobian of the column sums, three backward passes."""
    jac = torch.autograd.functional.jacobian(lambda inp: net(inp, emb).sum(dim=0), xyz.detach())
    return jac.permute(1, 0, 2).reshape(-1, 9).contiguous().detach()
