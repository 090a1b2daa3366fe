"""-m gpu: the UNTEXTURED surface (`diff_gauss`, reference render/render.py:52-53,66-68,75-84; used by stages 1-2:
models/gaussian3d.py:359, models/uv_map_gaussian3d.py:171-175) at BASELINE's full size against the fp32 C oracle, under the
same attribution bars as the textured operator (tests/test_parity_c_oracle_gpu.py): integer stages bit-exact, every
unambiguous pixel within 1e-4 (depth 4e-4) with identical n_contrib, every unflagged gradient row within 1e-3 relative +
1e-4 of the largest entry -- zero unexplained pixels / rows.

  shs      300 k Gaussians, 800x800, SH degree 3 with a DC row (colour offset = C0 * DC)          render/render.py:56-65
  precomp  300 k Gaussians, 800x800, colors_precomp (offset = colour - 0.5)                       render/render.py:66-68
  cov      100 k Gaussians, 800x800, cov3Ds_precomp [N,6] instead of scales / rotations           render/render.py:52-53

The kernels that run are K1<COV>, k_render_fwd<false>, k_render_bwd<false,true,false,false>, K8<COV>.  The product surface
(`diff_gauss.GaussianRasterizer`, autograd) is then run on the same inputs and must hand back the raw path's gradients, with
`means2D.grad[:, :2]` the lineage's dL/d(ndc xy) that stage-1 densification reads (models/gaussian3d.py:334-336)."""
import numpy as np
import pytest
import torch

from texgs import synth
from oracle import texgs_ref as CR
from oracle import texgs_torch as O
import helpers as Hh

pytestmark = pytest.mark.gpu

MODES = {"shs": (300_000, 3), "precomp": (300_000, 0), "cov": (100_000, 0)}
W = H = 800


def _dev(t, dev):
    return None if t is None else t.to(dev).contiguous()


@pytest.mark.parametrize("mode", ["shs", "precomp", "cov"])
def test_untextured_full_size_vs_c_oracle(lib_built, mode):
    import diff_gauss as dg
    from texgs.rasterizer import GaussianRasterizationSettings, forward_raw, backward_raw
    from texgs import rasterizer as RZ
    N, deg = MODES[mode]
    dev = torch.device("cuda:0")
    base = synth.make_scene(N, 4, seed=0, scale_mean=0.006)
    cam = synth.fibonacci_cameras(64, W, H)[3]
    bg = torch.tensor([0.0, 0.0, 0.0])
    us, colour = Hh.untextured_from(base, mode, seed=5)
    st_cpu = Hh.settings_for(cam, deg, bg)
    ref = CR.RefRun(us, st_cpu)
    ref.forward()
    margin, gflag, _ = ref.ambiguity(tau_fwd=Hh.TAU_FWD, tau_cell=0.0, tau_relu=Hh.tau_relu(1))
    st = Hh.settings_for(cam, deg, bg, device=dev, cls=GaussianRasterizationSettings)
    RZ.release_scratch()
    outs, s = forward_raw(st, _dev(us.means3D, dev), _dev(us.shs, dev), _dev(us.opacities, dev), _dev(us.scales, dev),
                          _dev(us.rotations, dev), None, None, None, color_offset=_dev(us.color_offset, dev),
                          cov3D_precomp=_dev(us.cov3D_precomp, dev))
    torch.cuda.synchronize()
    # ---- integer / index stages: identical
    Hh.assert_integer_stages_bit_exact(ref, outs, s)
    # ---- forward: every unambiguous pixel within north_star's tolerance
    got = torch.cat([outs[0], outs[1], outs[2], outs[3]], 0)
    nc = s.tensors["n_contrib"].cpu().numpy().astype(np.uint32)
    Hh.forward_attributed(f"untextured/{mode}/fwd", got, ref, margin, n_contrib=nc)
    # ---- backward
    g = torch.Generator().manual_seed(78)
    dout = torch.randn(8, H, W, generator=g) / (H * W)
    res = backward_raw(s, _dev(dout[0:3], dev), _dev(dout[3:4], dev), _dev(dout[4:7], dev), _dev(dout[7:8], dev))
    gref = ref.backward(dout.numpy())
    sens = ref.accumulation_sensitive()
    Hh.report(f"untextured/{mode}/bwd/accumulation_sensitive_rows", rows=int(sens.sum()), frac=float(sens.mean()))
    assert sens.mean() < 1e-3
    flag = gflag | sens
    got_g = dict(means3D=res[0], means2D=res[1], shs=res[2], opacities=res[3], scales=res[4], rotations=res[5],
                 color_offset=s.tensors["d_color_offset"], cov3D=s.tensors["d_cov3D"])
    assert res[6] is None and res[7] is None                    # no uvs, no texture
    checked = 0
    for name, exp in gref.items():
        if exp is None:
            assert got_g.get(name) is None, name
            continue
        assert got_g[name] is not None, name
        r = Hh.grad_attributed(f"untextured/{mode}/bwd/{name}", got_g[name].cpu(), torch.tensor(exp), flag, flagged_frac_max=0.15)
        checked += 1
    assert checked >= (7 if mode == "shs" else 6 if mode == "precomp" else 5)
    raw = {k: (None if v is None else v.detach().clone()) for k, v in got_g.items()}
    del res, got_g
    # ---- the product surface on the same inputs hands back the same gradients (run-to-run floor of the fp32 atomics)
    leaves = {n: _dev(getattr(us, n), dev).requires_grad_(True) for n in ("means3D", "opacities", "scales", "rotations", "cov3D_precomp")
              if getattr(us, n) is not None}
    col = _dev(torch.cat([colour[:, None, :], us.shs], 1) if mode == "shs" else colour, dev).requires_grad_(True)
    m2 = torch.zeros(N, 3, device=dev, requires_grad=True)
    kw = dict(shs=col) if mode == "shs" else dict(colors_precomp=col)
    out = dg.GaussianRasterizer(st)(means3D=leaves["means3D"], means2D=m2, opacities=leaves["opacities"], scales=leaves.get("scales"),
                                    rotations=leaves.get("rotations"), cov3Ds_precomp=leaves.get("cov3D_precomp"), **kw)
    assert out[5] is None
    for k in range(4):
        assert torch.equal(out[k], outs[k]), k                  # same kernels, same lists: bit-identical images
    assert torch.equal(out[4], outs[4])
    torch.autograd.backward([out[0], out[1], out[2], out[3]],
                            [_dev(dout[0:3], dev), _dev(dout[3:4], dev), _dev(dout[4:7], dev), _dev(dout[7:8], dev)])
    pairs = [("means3D", leaves["means3D"].grad), ("opacities", leaves["opacities"].grad.reshape(N, 1)), ("means2D", m2.grad)]
    if mode == "cov":
        pairs.append(("cov3D", leaves["cov3D_precomp"].grad))
    else:
        pairs += [("scales", leaves["scales"].grad), ("rotations", leaves["rotations"].grad)]
    if mode == "shs":
        pairs += [("shs", col.grad[:, 1:, :]), ("color_offset", col.grad[:, 0, :] / O.SH_C0)]
    else:
        pairs.append(("color_offset", col.grad))
    for name, got in pairs:
        rel = Hh.rel_err(got, raw[name].reshape(got.shape))
        Hh.report(f"untextured/{mode}/surface_vs_raw/{name}", rel_l2=rel)
        assert rel < 1e-5, (name, rel)
    assert float(m2.grad[:, 2].abs().max()) == 0.0              # z of the grad carrier stays 0 (lineage convention)
    RZ.release_scratch()
