"""-m gpu: HIP operator vs the fp32 C oracle (oracle/texgs_ref.c).

Bit-exact (integer / index work): radii, tile rects, tiles_touched, offsets, D, unsorted keys, sorted keys, point list,
tile ranges -- at small sizes AND at BASELINE's full sizes (configs[1] 100k/512^2 and configs[2] 300k/1024^2, 800x800).
Floating point at full size, by ATTRIBUTION (no outlier budgets): the C oracle computes, for the same inputs, where two fp32
implementations may legitimately differ -- pixels with a discrete decision (alpha >= 1/255, T >= 1e-4, power <= 0, cubemap
face, den >= DEN_MIN) within rounding of its threshold, and gradient rows (Gaussians / texels) with a (pixel, Gaussian) pair
within rounding of a bilinear-cell edge or of the colour clamp (oracle/texgs_ref.c texgs_ref_ambiguity, thresholds in
helpers.py).  EVERY other pixel must be within 1e-4 (depth 4e-4) -- north_star's tolerance, literally -- with n_contrib
identical, ambiguous pixels < 0.1 % of the image; EVERY unflagged gradient row within 1e-3 relative + 1e-4 of the largest
entry: zero unexplained outliers.  Measured figures: profiles/r04_parity_report.jsonl.  Plus size-independent properties on
the GPU outputs themselves."""
import math

import numpy as np
import pytest
import torch

from texgs import synth
from oracle import texgs_ref as CR
import helpers as Hh

pytestmark = pytest.mark.gpu

SIZES = {
    "small": (2000, 64, 200, 136, 0.02, 0),
    "c2": (100_000, 512, 800, 800, 0.006, 5),
    "c3": (300_000, 1024, 800, 800, 0.006, 0),
}
_cache = {}


def _run(name):
    if name in _cache:
        return _cache[name]
    N, R, W, H, sm, view = SIZES[name]
    scene = synth.make_scene(N, R, seed=0, scale_mean=sm)
    cam = synth.fibonacci_cameras(64 if N > 5000 else 4, W, H)[view]
    bg = torch.tensor([0.0, 0.0, 0.0]) if name != "small" else torch.tensor([0.3, 0.1, 0.2])
    st = Hh.settings_for(cam, 3, bg)
    ref = CR.RefRun(scene, st)
    ref.forward()
    ref.amb = ref.ambiguity(tau_fwd=Hh.TAU_FWD, tau_cell=Hh.tau_cell(R), tau_relu=Hh.tau_relu(R))
    outs, s = Hh.hip_debug_state(scene, cam, 3, bg)
    _cache.clear()                      # keep only one full-size scene alive
    _cache[name] = (scene, cam, bg, ref, outs, s)
    return _cache[name]


@pytest.mark.parametrize("name", ["small", "c2", "c3"])
def test_integer_stages_bit_exact(lib_built, name):
    scene, cam, bg, ref, outs, s = _run(name)
    Hh.assert_integer_stages_bit_exact(ref, outs, s)


@pytest.mark.parametrize("name", ["small", "c2", "c3"])
def test_forward_full_size_vs_c_oracle(lib_built, name):
    """north_star: per-pixel RGB / alpha within 1e-4 -- asserted on EVERY pixel the C oracle does not mark ambiguous."""
    scene, cam, bg, ref, outs, s = _run(name)
    got = torch.cat([outs[0], outs[1], outs[2], outs[3]], 0)
    nc = s.tensors["n_contrib"].cpu().numpy().astype(np.uint32)
    Hh.forward_attributed(f"hip_vs_c32/{name}/fwd", got, ref, ref.amb[0], n_contrib=nc)


@pytest.mark.parametrize("name", ["small", "c3"])
def test_backward_full_size_vs_c_oracle(lib_built, name):
    from texgs.rasterizer import backward_raw
    scene, cam, bg, ref, outs, s = _run(name)
    H, W = cam.image_height, cam.image_width
    g = torch.Generator().manual_seed(77)
    dout = torch.randn(8, H, W, generator=g) / (H * W)
    dev = outs[0].device
    res = backward_raw(s, dout[0:3].to(dev).contiguous(), dout[3:4].to(dev).contiguous(),
                       dout[4:7].to(dev).contiguous(), dout[7:8].to(dev).contiguous())
    gref = ref.backward(dout.numpy())
    _, gflag, tflag = ref.amb
    sens = ref.accumulation_sensitive()          # ill-conditioned scale gradients of needle-shaped splats (oracle/texgs_ref.py)
    Hh.report(f"hip_vs_c32/{name}/bwd/accumulation_sensitive_rows", rows=int(sens.sum()), frac=float(sens.mean()))
    assert sens.mean() < 1e-3
    gflag = gflag | sens
    names = ["means3D", "means2D", "shs", "opacities", "scales", "rotations", "uvs", "texture"]
    for name_, got in zip(names, res[:8]):
        Hh.grad_attributed(f"hip_vs_c32/{name}/bwd/{name_}", got.cpu(), torch.tensor(gref[name_]),
                           tflag if name_ == "texture" else gflag)
    # the same gradients at PAIR level: every row checked, tolerance widened by what its own near-cell-edge pairs can contribute
    Hh.pair_level_gradient_check(f"hip_vs_c32/{name}/bwd_pair_level", ref, res, dout, scene.texture.shape[1], sens)


def test_band_limited_texture_c3(lib_built):
    """C3 geometry, band-limited texture: no cell-edge excuses, flagged rows < 5 %, zero unexplained pixels / rows."""
    scene, cam, bg, ref, outs, s = _run("c3")
    Hh.band_limited_parity("hip_vs_c32/c3_band_limited", scene, cam, bg)


def test_size_independent_properties_full_size(lib_built):
    """On BASELINE's full size (configs[2]): alpha = 1 - T_final; image is affine in bg with slope T_final; sorted
    keys are sorted and tile-contiguous; ranges partition [0, D); n_contrib never exceeds the tile's list."""
    scene, cam, bg, ref, outs, s = _run("c3")
    t = s.tensors
    alpha, Tf = outs[3][0], t["final_T"]
    assert float((alpha - (1.0 - Tf)).abs().max()) < 2e-5
    bg2 = torch.tensor([0.7, 0.2, 0.4])
    outs2, s2 = Hh.hip_debug_state(scene, cam, 3, bg2)
    lin = outs2[0] - outs[0] - Tf[None] * (bg2 - bg).to(Tf.device)[:, None, None]
    assert float(lin.abs().max()) < 1e-5
    assert torch.equal(outs2[3], outs[3]) and torch.equal(outs2[4], outs[4])           # geometry independent of bg
    D = s.D
    ks = t["keys_sorted"][:D]
    assert bool((ks[1:] >= ks[:-1]).all())
    rg = t["ranges"].to(torch.int64)
    ne = rg[:, 1] > rg[:, 0]
    assert int((rg[ne, 1] - rg[ne, 0]).sum()) == D
    tile_of = (ks >> 32)
    starts = rg[ne, 0]
    assert bool((tile_of[starts] == torch.nonzero(ne).reshape(-1)).all())
    H, W = cam.image_height, cam.image_width
    nc = t["n_contrib"].to(torch.int64)
    lens = (rg[:, 1] - rg[:, 0]).reshape((H + 15) // 16, (W + 15) // 16)
    per_px = lens.repeat_interleave(16, 0).repeat_interleave(16, 1)[:H, :W]
    assert bool((nc <= per_px).all())


def test_stress_scene_vs_c_oracle(lib_built):
    from texgs.rasterizer import backward_raw
    scene = Hh.stress_scene()
    cam = synth.look_at_camera((0.3, -0.2, -3.0), 640, 360, fovx=1.2)
    bg = torch.tensor([0.05, 0.1, 0.15])
    ref = CR.RefRun(scene, Hh.settings_for(cam, 3, bg))
    ref.forward()
    outs, s = Hh.hip_debug_state(scene, cam, 3, bg)
    N, D, t = ref.N, ref.D, s.tensors
    assert D > 100000 and int((ref.radii[:N] == 0).sum()) > N // 10         # big lists AND many culled
    assert np.array_equal(outs[4].cpu().numpy(), ref.radii[:N])
    assert s.D == D
    assert np.array_equal(t["point_list"][:D].cpu().numpy().astype(np.uint32), ref.point_list[:D])
    assert np.array_equal(t["ranges"].cpu().numpy().astype(np.uint32), ref.ranges)
    # by attribution, like the benchmark scenes (VERDICT r4 #3c; until round 5: "2e-3 of the pixels over 1e-4, worst 2e-2").  Depth
    # reaches ~8 scene units here (the benchmark sphere: 3.2, tolerance 4e-4): the same 1.25e-4 of the range.
    R = scene.texture.shape[1]
    margin, gflag, tflag = ref.ambiguity(tau_fwd=Hh.TAU_FWD, tau_cell=Hh.tau_cell(R), tau_relu=Hh.tau_relu(R))
    got = torch.cat([outs[0], outs[1], outs[2], outs[3]], 0).cpu()
    zmax = float(ref.out[3].max())
    Hh.forward_attributed("hip_vs_c32/stress/fwd", got, ref, margin, depth_tol=1.25e-4 * max(zmax, 3.2),
                          n_contrib=t["n_contrib"].cpu().numpy().astype(np.uint32), amb_frac_max=5e-3)
    H, W = cam.image_height, cam.image_width
    gen = torch.Generator().manual_seed(11)
    dout = torch.randn(8, H, W, generator=gen) / (H * W)
    dev = outs[0].device
    res = backward_raw(s, dout[0:3].to(dev).contiguous(), dout[3:4].to(dev).contiguous(), dout[4:7].to(dev).contiguous(),
                       dout[7:8].to(dev).contiguous())
    # gradients: this scene's screen-filling splats make the falloff exponent itself ill-conditioned, and with it every gradient that
    # passes through alpha.  Until round 5: row budgets (0.5 % outlier rows, 1e-2 global L2).  Now the oracle's backward books what that
    # conditioning can move, pair by pair, and EVERY row is checked against plain tolerance + its own bound: zero unexplained rows
    # (helpers.stress_gradient_check; validated on the CPU against a differently-rounded build, tests/test_c_oracle.py)
    Hh.stress_gradient_check("hip_vs_c32/stress", ref, list(res[:8]), dout.numpy(), R)


def test_more_than_65536_tiles_three_digit_tile_sort(lib_built):
    """ADVICE r2 / VERDICT r2 #5: above 65 536 tiles (here 4112 x 4112 px = 257 x 257 = 66 049 tiles) the tile id no longer fits
    two 8-bit radix digits; the tile sort then runs three passes.  Sorted keys, point list, ranges and the image must equal
    the C oracle's -- and the middle pass may use keys_unsorted as scratch, nothing else is allowed to change."""
    W = H = 4112
    scene = synth.make_scene(3000, 32, seed=5, scale_mean=0.02)
    cam = synth.fibonacci_cameras(4, W, H)[2]
    bg = torch.zeros(3)
    st = Hh.settings_for(cam, 2, bg)
    ref = CR.RefRun(scene, st)
    ref.forward()
    outs, s = Hh.hip_debug_state(scene, cam, 2, bg)
    t, N, D = s.tensors, ref.N, ref.D
    assert t["ranges"].shape[0] == 257 * 257 and s.D == D and D > 10000
    assert np.array_equal(outs[4].cpu().numpy(), ref.radii[:N])
    assert np.array_equal(t["keys_sorted"][:D].cpu().numpy().view(np.uint64), ref.keys_sorted[:D])
    assert np.array_equal(t["point_list"][:D].cpu().numpy().astype(np.uint32), ref.point_list[:D])
    assert np.array_equal(t["ranges"].cpu().numpy().astype(np.uint32), ref.ranges)
    assert int((ref.keys_sorted[:D] >> np.uint64(32)).max()) >= 65536          # the third digit is really exercised
    got = torch.cat([outs[0], outs[1], outs[2], outs[3]], 0).cpu()
    margin, _, _ = ref.ambiguity(tau_fwd=Hh.TAU_FWD, tau_cell=Hh.tau_cell(32), tau_relu=Hh.tau_relu(32))
    Hh.forward_attributed("hip_vs_c32/tiles66049/fwd", got, ref, margin, n_contrib=t["n_contrib"].cpu().numpy().astype(np.uint32))
    from texgs import _lib
    import ctypes as C
    lib = _lib.load()
    fr2 = _lib.Frame(4097 * 16, 4097 * 16, 0.3, 0.3, 1.0, 0, 0, 4, 0, 0, 1, 1, 1, 1)  # 16.8 M tiles (> 2^24): refused, not mis-sorted
    assert lib.texgs_mark_visible(C.byref(fr2), 1, 1, None) != 0 and b"2^24 tiles" in lib.texgs_last_error()


@pytest.mark.parametrize("kind", ["one_depth", "two_clusters", "far_outlier"])
def test_depth_sort_with_crowded_depth_bins(lib_built, kind):
    """K2 partitions the Gaussians into depth bins (equal width in key space between the smallest and the largest valid key), cuts
    the bins into balanced groups and sorts every group: <= 1 024 pairs in registers, <= 2 048 in LDS, more (one BIN holds them: many
    Gaussians at (nearly) the same view depth, a wall seen frontally) by a one-wave LSD radix in global memory.  one_depth: 12 000
    Gaussians on 3 distinct depths (range narrower than the bin count: a bin holds one key, thousands of times: sorted by index
    only); two_clusters: two sheets of 48 depths and one far Gaussian that stretches the range (key passes AND index passes);
    far_outlier: 30 000 Gaussians spread over two units of depth and one at 90 -- the content falls into a few bins of several
    ~1 000 either side of 1 024 (the widest register network and the LDS sort).  Rank order, offsets, instance keys, point list and
    ranges must be the C oracle's, bit for bit."""
    g = torch.Generator().manual_seed(3)
    N = {"one_depth": 12000, "two_clusters": 9001, "far_outlier": 30000}[kind]
    cam = synth.look_at_camera((0.0, 0.0, -3.2), 320, 240, fovx=0.9)
    scene = synth.make_scene(N, 32, seed=8, scale_mean=0.02)
    xy = (torch.rand(N, 2, generator=g) - 0.5) * 1.8
    steps = torch.randint(0, 3 if kind == "one_depth" else 48, (N,), generator=g).float() * 2.0 ** -21
    if kind == "one_depth":
        z = -1.2 + steps
    elif kind == "two_clusters":
        z = torch.where(torch.arange(N) % 2 == 0, -1.2 + steps, 0.8 + steps)
        z[-1] = 90.0
    else:
        z = 1.5 + 2.0 * torch.rand(N, generator=g)
        z[-1] = 90.0
    means = torch.cat([xy, z[:, None]], 1).float().contiguous()
    scene = scene._replace(means3D=means)
    bg = torch.zeros(3)
    ref = CR.RefRun(scene, Hh.settings_for(cam, 2, bg))
    ref.forward()
    outs, s = Hh.hip_debug_state(scene, cam, 2, bg)
    t, D = s.tensors, ref.D
    vis = ref.radii[:N] > 0
    keys = np.where(vis, ref.depth[:N].view(np.uint32), np.uint32(0xFFFFFFFF))
    # the kernel's bin of a key (binning.hip depth_range / depth_bin): NB = 256 for these N
    kv = keys[vis].astype(np.uint64)
    lo, rng, nb = int(kv.min()), int(kv.max() - kv.min()), 256
    bins = (kv - lo) if rng < nb else (((kv - lo) * np.uint64((nb << 32) // (rng + 1))) >> np.uint64(32))
    assert int(bins.max()) < nb
    cnt = np.unique(bins, return_counts=True)[1]
    if kind == "far_outlier":
        assert 1024 < cnt.max() <= 2048 and np.sum((cnt > 512) & (cnt <= 1024)) > 0, np.sort(cnt)[-8:]
    else:
        assert cnt.max() > 2048, cnt.max()                                        # a bin beyond the in-LDS sort's capacity
    assert s.D == D and D > 10000
    assert np.array_equal(outs[4].cpu().numpy(), ref.radii[:N])
    order = np.argsort(keys.astype(np.uint64), kind="stable")
    tt_rank = ref.tiles[:N][order].astype(np.int64)
    offs = t["offsets"][:N].cpu().numpy().astype(np.uint32).astype(np.int64)
    nv = int(vis.sum())
    assert np.array_equal(offs[:nv], (np.cumsum(tt_rank) - tt_rank)[:nv]) and np.all(offs[nv:] == D)
    assert np.array_equal(t["keys_sorted"][:D].cpu().numpy().view(np.uint64), ref.keys_sorted[:D])
    assert np.array_equal(t["point_list"][:D].cpu().numpy().astype(np.uint32), ref.point_list[:D])
    assert np.array_equal(t["ranges"].cpu().numpy().astype(np.uint32), ref.ranges)
    got = torch.cat([outs[0], outs[1], outs[2], outs[3]], 0).cpu()
    margin, _, _ = ref.ambiguity(tau_fwd=Hh.TAU_FWD, tau_cell=Hh.tau_cell(32), tau_relu=Hh.tau_relu(32))
    zmax = float(ref.out[3].max())                   # (far_outlier / two_clusters: a splat at depth 90)
    Hh.forward_attributed(f"hip_vs_c32/crowded_{kind}/fwd", got, ref, margin, depth_tol=1.25e-4 * max(zmax, 3.2),
                          n_contrib=t["n_contrib"].cpu().numpy().astype(np.uint32), amb_frac_max=5e-3)


@pytest.mark.parametrize("N", [1, 2, 63, 64, 65, 200, 257, 1025, 2049, 4097])
def test_depth_sort_at_the_size_boundaries(lib_built, N):
    """K2 / K3 at the sizes where their paths change hands: one wave (<= 64 pairs per group), 256 / 512 / 1 024 pairs in registers,
    one and several blocks of the partition kernels (2 048 keys each), one and several K3 workgroups.  A third of the Gaussians sit
    behind the camera (culled: key 0xFFFFFFFF, their own bin).  Offsets in rank order, instance keys, point list and ranges are the
    C oracle's, bit for bit."""
    g = torch.Generator().manual_seed(100 + N)
    cam = synth.look_at_camera((0.0, 0.0, -3.2), 160, 128, fovx=0.9)
    scene = synth.make_scene(N, 16, seed=9, scale_mean=0.05)
    xy = (torch.rand(N, 2, generator=g) - 0.5) * 1.6
    z = torch.rand(N, generator=g) * 2.0 - 1.0
    z[torch.arange(N) % 3 == 2] = -5.0                  # behind the camera
    scene = scene._replace(means3D=torch.cat([xy, z[:, None]], 1).float().contiguous())
    bg = torch.zeros(3)
    ref = CR.RefRun(scene, Hh.settings_for(cam, 1, bg))
    ref.forward()
    outs, s = Hh.hip_debug_state(scene, cam, 1, bg)
    t, D = s.tensors, ref.D
    assert s.D == D
    vis = ref.radii[:N] > 0
    assert np.array_equal(outs[4].cpu().numpy(), ref.radii[:N])
    if D == 0:
        return
    keys = np.where(vis, ref.depth[:N].view(np.uint32), np.uint32(0xFFFFFFFF))
    order = np.argsort(keys.astype(np.uint64), kind="stable")
    tt_rank = ref.tiles[:N][order].astype(np.int64)
    offs = t["offsets"][:N].cpu().numpy().astype(np.uint32).astype(np.int64)
    nv = int(vis.sum())
    assert np.array_equal(offs[:nv], (np.cumsum(tt_rank) - tt_rank)[:nv]) and np.all(offs[nv:] == D)
    assert np.array_equal(t["keys_sorted"][:D].cpu().numpy().view(np.uint64), ref.keys_sorted[:D])
    assert np.array_equal(t["point_list"][:D].cpu().numpy().astype(np.uint32), ref.point_list[:D])
    assert np.array_equal(t["ranges"].cpu().numpy().astype(np.uint32), ref.ranges)
