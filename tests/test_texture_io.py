"""-m "not gpu": texture / checkpoint I/O against fixtures computed by the reference's own
rgb2sh0 / sh02rgb / cube_map / change_texture (tests/golden/make_golden.py runs them from the reference source), plus
round trips and the PNG / checkpoint file formats."""
import os

import numpy as np
import pytest
import torch

from texgs import texture_io as TIO

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "texture_io.npz"))


def _inputs():
    cross = torch.tensor(G["cross_u8"].astype(np.float32) / 255.0)
    return cross, torch.tensor(G["texture0"])


def test_texel_rgb_maps_match_reference():
    cross, tex0 = _inputs()
    assert np.array_equal(TIO.sh02rgb(tex0).numpy(), G["sh02rgb"])
    assert np.array_equal(TIO.rgb2sh0(cross).numpy(), G["rgb2sh0"])
    inside = (TIO.sh02rgb(tex0) > 0) & (TIO.sh02rgb(tex0) < 1)
    assert torch.allclose(TIO.rgb2sh0(TIO.sh02rgb(tex0))[inside], tex0[inside], atol=1e-5)      # inverse where not clamped


def test_cross_layout_matches_reference_cube_map():
    cross, tex0 = _inputs()
    assert np.array_equal(TIO.texture_to_cross(tex0).numpy(), G["cube_map"])
    faces = TIO.sh02rgb(tex0)
    assert torch.equal(TIO.cross_to_cube(TIO.cube_to_cross(faces)), faces)                       # bijection on the six faces
    r = faces.shape[1]
    x = TIO.cube_to_cross(faces)
    assert float(x[:r, :r].abs().max()) == 0.0 and float(x[2 * r:, 2 * r:].abs().max()) == 0.0   # unused cells stay empty
    with pytest.raises(ValueError):
        TIO.cross_to_cube(torch.zeros(30, 41, 3))


@pytest.mark.parametrize("mode", [-1, 0, 1, 2, 3])
def test_change_texture_modes_match_reference(mode):
    cross, tex0 = _inputs()
    got = TIO.change_texture(tex0, cross, mode).numpy()
    exp = G[f"change_texture_m{mode + 1}"]
    assert got.shape == exp.shape
    both_inf = np.isinf(got) & np.isinf(exp) & (np.sign(got) == np.sign(exp))                    # mode 2 divides by black texels
    both_nan = np.isnan(got) & np.isnan(exp)
    ok = both_inf | both_nan | (np.abs(got - exp) <= 1e-6 * np.maximum(1.0, np.abs(exp)))
    assert ok.all(), (mode, int((~ok).sum()))


def test_png_round_trip_and_resize(tmp_path):
    cross, tex0 = _inputs()
    p = str(tmp_path / "tex.png")
    TIO.save_cross_png(tex0, p)
    back = TIO.load_cross_png(p, tex0.shape[1])
    exp = (torch.clamp(TIO.texture_to_cross(tex0), 0, 1) * 255).numpy().astype(np.uint8).astype(np.float32) / 255.0
    assert np.array_equal(back.numpy(), exp)                                                      # same resolution: lossless
    up = TIO.load_cross_png(p, 2 * tex0.shape[1])
    assert tuple(up.shape) == (6 * tex0.shape[1], 8 * tex0.shape[1], 3)
    # bilinear, half-pixel centres: a 2x upsample keeps every value inside the range of its 2x2 source neighbourhood
    assert float(up.max()) <= float(back.max()) + 1e-6 and float(up.min()) >= float(back.min()) - 1e-6
    const = np.full((6, 8, 3), 77, np.uint8)
    assert np.array_equal(TIO.resize_bilinear_u8(const, 15, 20), np.full((15, 20, 3), 77, np.uint8))
    ramp = np.tile(np.arange(8, dtype=np.uint8)[None, :, None] * 30, (6, 1, 3))
    r2 = TIO.resize_bilinear_u8(ramp, 6, 16).astype(np.int32)
    assert np.all(np.diff(r2[0, :, 0]) >= 0) and r2[0, 0, 0] == 0 and r2[0, -1, 0] == 210         # monotone, edges clamped
    with pytest.raises(ValueError):
        from PIL import Image
        Image.fromarray(np.zeros((10, 10, 3), np.uint8)).save(str(tmp_path / "bad.png"))
        TIO.load_cross_png(str(tmp_path / "bad.png"), 4)


def test_checkpoint_round_trip(tmp_path):
    g = torch.Generator().manual_seed(0)
    N, R = 50, 4
    st = TIO.GaussianState(torch.randn(N, 3, generator=g), torch.randn(N, 3, generator=g), torch.randn(N, 4, generator=g),
                           torch.randn(N, 1, generator=g), torch.randn(N, 15, 3, generator=g), torch.randn(6, R, R, 3, generator=g),
                           3, 1.5, ({"w": torch.ones(2)}, {}, {"weight": torch.zeros(1, 8)}), None, 40000)
    p = str(tmp_path / "40000.pth")
    TIO.save_checkpoint(st, p)
    # the file is the reference's layout: (state_dict, iteration) with the four top-level keys and the 6-tuple of RAW params
    sd, it = torch.load(p, weights_only=True)
    assert it == 40000 and set(sd) == {"hyperparams", "optim_state", "net_state", "params"} and len(sd["params"]) == 6
    back = TIO.load_checkpoint(p)
    for a, b in zip(st[:6], back[:6]):
        assert torch.equal(a, b)
    assert back.active_sh_degree == 3 and back.spatial_lr_scale == 1.5 and back.iteration == 40000
    assert torch.allclose(back.get_rotation().norm(dim=1), torch.ones(N)) and bool((back.get_opacity() < 1).all())
    assert torch.equal(back.get_scaling(), torch.exp(st.scaling))
    torch.save(({"hyperparams": (0, 1.0), "params": (1, 2, 3)}, 1), p)
    with pytest.raises(ValueError, match="stage-3"):
        TIO.load_checkpoint(p)


def test_reference_format_checkpoint_fixture():
    """tests/golden/ckpt_stage3.pth was written by the REFERENCE's own TextureGaussian3D.setup_optim / state_dict
    (models/texture_gaussian3d.py:99-172, run via `ast` by tests/golden/make_golden.py) and train.py's torch.save: nn.Parameter
    entries, three Adam states, the scheduler state, tiny-cuda-nn style flat `params` for the UV networks.  load_checkpoint
    must take it with weights_only=True, and the UV network it carries must evaluate to the stored uvs."""
    from texgs.uvnet import UVNet
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    st = TIO.load_checkpoint(os.path.join(here, "ckpt_stage3.pth"))            # weights_only=True by default
    E = np.load(os.path.join(here, "ckpt_stage3_expect.npz"))
    N, R = 40, 4
    assert st.iteration == 40000 and st.active_sh_degree == 3 and abs(st.spatial_lr_scale - 1.7) < 1e-12
    assert tuple(st.xyz.shape) == (N, 3) and tuple(st.scaling.shape) == (N, 3) and tuple(st.rotation.shape) == (N, 4)
    assert tuple(st.opacity.shape) == (N, 1) and tuple(st.shs.shape) == (N, 15, 3) and tuple(st.texture.shape) == (6, R, R, 3)
    assert not any(t.requires_grad for t in st[:6])                              # nn.Parameter in the file, plain tensors out
    assert np.array_equal(st.xyz.numpy(), E["xyz_param"]) and np.array_equal(st.texture.numpy(), E["texture"])
    assert torch.allclose(st.get_rotation().norm(dim=1), torch.ones(N), atol=1e-6) and bool((st.get_scaling() > 0).all())
    uv_state, inv_state, emb_state = st.net_state
    assert set(uv_state) == {"pre_mlp.params", "mlp.params"} and set(emb_state) == {"weight"}
    assert len(st.optim_state) == 4 and "param_groups" in st.optim_state[0]
    net = UVNet().load_reference_state(uv_state).double()
    uvs = net(torch.tensor(E["xyz"]), emb_state["weight"][0].double())
    assert float((uvs - torch.tensor(E["uvs"])).abs().max()) < 1e-6
    # and it goes back out in the same layout
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        TIO.save_checkpoint(st, os.path.join(d, "x.pth"))
        sd, it = torch.load(os.path.join(d, "x.pth"), weights_only=True)
        assert it == 40000 and set(sd) == {"hyperparams", "optim_state", "net_state", "params"} and len(sd["params"]) == 6
