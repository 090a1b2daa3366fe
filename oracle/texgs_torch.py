"""TEST INFRASTRUCTURE ONLY -- CPU restatement (torch, float64 by default) of the
Texture-GS textured Gaussian rasterizer operator.  Nothing under texture-gs_amd/
may import this file; only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg do, and only as the checker.

PARITY UNPINNED: the arithmetic of the reference operator lives in the un-vendored,
un-pinned pip dependency `diff_gauss_uv_tex` (reference requirements.txt:15, bare
git+https HEAD of slothfulxtx/diff-gauss-uv-tex) whose source is absent from
/root/reference, and the reference holds no test or golden vector for it.  This
file restates the published algorithm (3DGS tile rasterizer lineage acknowledged at
reference README.md:170 + Texture-GS paper arXiv 2403.10050 Sec. 3) and anchors on
the reference's own call sites and in-tree conventions:

  * operator surface / argument meaning    render/uv_tex_render.py:25-38,56-66
  * row-vector matrices, camera centre     utils/cameras.py:62-65, utils/graphics.py:22-29,51-71
  * pixel <-> ndc, depth = view-space z    models/texture_gaussian3d.py:299-309
  * quaternion (w,x,y,z) -> R, cov = RS(RS)^T  utils/general.py:87-119, models/gaussian3d.py:17-21
  * SH basis constants / signs             utils/sh.py:26-112 ; colour = clamp_min(sh+0.5,0) render/render.py:68
  * texel = SH-DC, rgb = C0*t+0.5           models/texture_gaussian3d.py:16-21
  * cubemap face convention                models/modules/NVDIFFREC/util.py:94-101
  * J layout  [n, 3*i+j] = d uv_i / d x_j  models/texture_gaussian3d.py:216-227

Backward is torch autograd of this forward (no hand-written gradient here), which makes it
an independent check of the hand-derived HIP backward kernels.  One deliberate deviation from
plain autograd, following the lineage: the gradient passes straight through the min(0.99, .)
clamp on alpha.

Every unverifiable decision is a named constant below and is listed in DESIGN.md section 3.
"""
import math
from typing import NamedTuple, Optional

import torch

# ---------------------------------------------------------------- constants (SPEC)
TILE = 16                 # 16x16 pixel tiles
NEAR_Z = 0.2              # cull view z <= 0.2
LOWPASS = 0.3             # +0.3 px^2 on the cov2D diagonal
FRUSTUM_CLAMP = 1.3       # tx/tz clamped to +-1.3 tanfov in the EWA Jacobian
ALPHA_MAX = 0.99
ALPHA_MIN = 1.0 / 255.0
T_EPS = 1e-4              # stop when T*(1-alpha) < 1e-4 (before accumulating)
PLANE_EPS = 5e-2          # |n.t| <= PLANE_EPS*|t| (plane seen > 87.1 deg from its normal, > 20x in-plane stretch): uv = phi (G = g = 0)
DEN_MIN = 0.2             # 1 + g.dp < DEN_MIN (intersection > 5x the centre depth): uv = phi
MA_MIN = 1e-20            # guard on the cubemap major axis

SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005,
         -1.0925484305920792, 0.5462742152960396]
SH_C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154,
         -0.4570457994644658, 1.445305721320277, -0.5900435899266435]


class Settings(NamedTuple):
    """Field order = keyword order at render/uv_tex_render.py:25-38."""
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


def build_rotation(q):
    """utils/general.py:87-108 without the normalisation (caller passes unit quaternions,
    models/texture_gaussian3d.py:201-202; the lineage uses q as given)."""
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=-1)
    return R.reshape(-1, 3, 3)


def sh_view_dependent(deg, shs, dirs):
    """Bands 1..deg of utils/sh.py:57-112 evaluated on shs[N, K, 3] whose index 0 is SH
    coefficient 1 (band 0 is the per-pixel texture, models/texture_gaussian3d.py:98)."""
    N = dirs.shape[0]
    res = torch.zeros(N, 3, dtype=dirs.dtype)
    if deg < 1 or shs is None:
        return res
    x, y, z = dirs[:, 0:1], dirs[:, 1:2], dirs[:, 2:3]
    sh = lambda k: shs[:, k - 1, :]
    res = res - SH_C1 * y * sh(1) + SH_C1 * z * sh(2) - SH_C1 * x * sh(3)
    if deg > 1:
        xx, yy, zz = x * x, y * y, z * z
        xy, yz, xz = x * y, y * z, x * z
        res = (res + SH_C2[0] * xy * sh(4) + SH_C2[1] * yz * sh(5)
               + SH_C2[2] * (2.0 * zz - xx - yy) * sh(6)
               + SH_C2[3] * xz * sh(7) + SH_C2[4] * (xx - yy) * sh(8))
        if deg > 2:
            res = (res + SH_C3[0] * y * (3 * xx - yy) * sh(9)
                   + SH_C3[1] * xy * z * sh(10)
                   + SH_C3[2] * y * (4 * zz - xx - yy) * sh(11)
                   + SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh(12)
                   + SH_C3[4] * x * (4 * zz - xx - yy) * sh(13)
                   + SH_C3[5] * z * (xx - yy) * sh(14)
                   + SH_C3[6] * x * (xx - 3 * yy) * sh(15))
    return res


def preprocess(means3D, means2D, shs, opacities, scales, rotations, uvs, gradient_uvs, st, dtype, color_offset=None,
               cov3D_precomp=None):
    """K1 of SURVEY Appendix A.2 (+ the texture pre-fold).  Returns a dict of per-Gaussian state.
    cov3D_precomp [N,6] (xx,xy,xz,yy,yz,zz -- strip_lowerdiag, utils/general.py:73-82; render/render.py:52-53): the world
    covariance, used instead of scales / rotations and without the scale modifier (lineage); the normal is then the
    eigenvector of the smallest eigenvalue, a selection like the shortest-axis choice: no gradient through it."""
    H, W = int(st.image_height), int(st.image_width)
    V = st.viewmatrix.to(dtype)      # row-vector: p_view = [x,y,z,1] @ V   (utils/cameras.py:62)
    P = st.projmatrix.to(dtype)      # full world->clip, row-vector          (utils/cameras.py:64)
    cam = st.campos.to(dtype)
    N = means3D.shape[0]
    fx = W / (2.0 * st.tanfovx)      # utils/graphics.py:73-74
    fy = H / (2.0 * st.tanfovy)
    gx, gy = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE

    ones = torch.ones(N, 1, dtype=dtype)
    hom = torch.cat([means3D, ones], dim=1)
    t = (hom @ V)[:, :3]
    tx, ty, tz = t[:, 0], t[:, 1], t[:, 2]
    valid = tz > NEAR_Z
    tzs = torch.where(valid, tz, torch.ones_like(tz))    # keep culled rows finite

    clip = hom @ P
    pw = 1.0 / (clip[:, 3] + 1e-7)
    ndc = clip[:, :2] * pw[:, None]
    if means2D is not None:          # zero grad-carrier (render/uv_tex_render.py:15): its gradient is
        ndc = ndc + means2D[:, :2]   # dL/d(ndc xy) = dL/d(pixel xy) * S/2, the lineage's convention
    xy = torch.stack([((ndc[:, 0] + 1.0) * W - 1.0) * 0.5, ((ndc[:, 1] + 1.0) * H - 1.0) * 0.5], dim=1)

    if cov3D_precomp is not None:
        c6 = cov3D_precomp
        Sigma = torch.stack([c6[:, 0], c6[:, 1], c6[:, 2], c6[:, 1], c6[:, 3], c6[:, 4],
                             c6[:, 2], c6[:, 4], c6[:, 5]], dim=1).reshape(N, 3, 3)
        R = None
    else:
        R = build_rotation(rotations)
        s = scales * st.scale_modifier
        M = R * s[:, None, :]            # R @ diag(s)
        Sigma = M @ M.transpose(1, 2)    # models/gaussian3d.py:17-21

    limx, limy = FRUSTUM_CLAMP * st.tanfovx, FRUSTUM_CLAMP * st.tanfovy
    txtz, tytz = tx / tzs, ty / tzs
    clx = (txtz < -limx) | (txtz > limx)      # clamped: lineage treats the clamped value as a constant
    cly = (tytz < -limy) | (tytz > limy)
    txc = torch.where(clx, (torch.clamp(txtz, -limx, limx) * tzs).detach(), tx)
    tyc = torch.where(cly, (torch.clamp(tytz, -limy, limy) * tzs).detach(), ty)
    zero = torch.zeros_like(tzs)
    J = torch.stack([fx / tzs, zero, -fx * txc / (tzs * tzs),
                     zero, fy / tzs, -fy * tyc / (tzs * tzs)], dim=1).reshape(N, 2, 3)
    Wr = V[:3, :3].t()               # world->view rotation, column-vector form
    Tm = J @ Wr
    cov = Tm @ Sigma @ Tm.transpose(1, 2)
    a = cov[:, 0, 0] + LOWPASS
    b = cov[:, 0, 1]
    c = cov[:, 1, 1] + LOWPASS
    det = a * c - b * b
    valid = valid & (det != 0)
    dets = torch.where(det != 0, det, torch.ones_like(det))
    conic = torch.stack([c / dets, -b / dets, a / dets], dim=1)
    mid = 0.5 * (a + c)
    lam = mid + torch.sqrt(torch.clamp(mid * mid - det, min=0.1))
    radius = torch.ceil(3.0 * torch.sqrt(lam.detach())).to(torch.int64)

    xyd = xy.detach()
    rf = radius.to(dtype)
    rminx = torch.clamp(((xyd[:, 0] - rf) / TILE).to(torch.int64), 0, gx)   # trunc toward zero
    rminy = torch.clamp(((xyd[:, 1] - rf) / TILE).to(torch.int64), 0, gy)
    rmaxx = torch.clamp(((xyd[:, 0] + rf + (TILE - 1)) / TILE).to(torch.int64), 0, gx)
    rmaxy = torch.clamp(((xyd[:, 1] + rf + (TILE - 1)) / TILE).to(torch.int64), 0, gy)
    tiles = (rmaxx - rminx) * (rmaxy - rminy)
    valid = valid & (tiles > 0)
    tiles = torch.where(valid, tiles, torch.zeros_like(tiles))
    radius = torch.where(valid, radius, torch.zeros_like(radius))

    dirs = means3D - cam[None, :]
    dirn = dirs / dirs.norm(dim=1, keepdim=True)          # render/render.py:65-66
    viewdep = sh_view_dependent(int(st.sh_degree), shs, dirn)
    if color_offset is not None:          # untextured surface (render/render.py:63-68): C0*SH_DC or colors_precomp - 0.5
        viewdep = viewdep + color_offset

    # normal = shortest axis, flipped to face the camera, world space
    if R is None:
        n = torch.linalg.eigh(Sigma.detach()).eigenvectors[:, :, 0]      # eigenvalues ascending: column 0 = smallest
    else:
        kmin = torch.argmin(scales.detach(), dim=1)
        n = R[torch.arange(N), :, kmin]
    flip = (n.detach() * dirs.detach()).sum(1) > 0
    n = torch.where(flip[:, None], -n, n)

    # texture pre-fold:  uv(p) = phi + G dp / (1 + g.dp),  dp = pix - xy
    nv = n @ Wr.t()                                       # view-space normal
    sdot = (nv * t).sum(1)
    tn = t.norm(dim=1)
    degen = sdot.detach().abs() <= PLANE_EPS * tn.detach()
    ssafe = torch.where(degen, torch.ones_like(sdot), sdot)
    avec = tzs[:, None] * nv / ssafe[:, None]
    g = torch.stack([avec[:, 0] / fx, avec[:, 1] / fy], dim=1)
    E = torch.zeros(3, 2, dtype=dtype)
    E[0, 0] = 1.0 / fx
    E[1, 1] = 1.0 / fy
    B = tzs[:, None, None] * E[None] - t[:, :, None] * g[:, None, :]     # (N,3,2)
    Jphi = gradient_uvs.reshape(N, 3, 3)
    G = Jphi @ Wr.t()[None] @ B                          # J * R_c2w * B
    keep = (~degen).to(dtype)
    G = G * keep[:, None, None]
    g = g * keep[:, None]

    # the values the integer decisions were rounded FROM (tests flag Gaussians whose float64 value sits within fp32 rounding of a
    # ceil / trunc boundary: there an fp32 evaluation of the same formula may legitimately land on the other side)
    radius_f = 3.0 * torch.sqrt(lam.detach())
    rect_f = torch.stack([(xyd[:, 0] - rf) / TILE, (xyd[:, 1] - rf) / TILE,
                          (xyd[:, 0] + rf + (TILE - 1)) / TILE, (xyd[:, 1] + rf + (TILE - 1)) / TILE], dim=1)
    return dict(valid=valid, xy=xy, depth=tz, conic=conic, radius=radius, tiles=tiles,
                rect=(rminx, rminy, rmaxx, rmaxy), viewdep=viewdep, normal=n,
                G=G, g=g, phi=uvs, opacity=opacities.reshape(-1), grid=(gx, gy), radius_f=radius_f, rect_f=rect_f)


def bin_and_sort(pre, depth_f32_bits=True):
    """K2-K5 of SURVEY Appendix A.3: offsets, (tile<<32 | depthbits, id) pairs, stable sort, ranges."""
    gx, gy = pre['grid']
    tiles = pre['tiles']
    offsets = torch.cumsum(tiles, 0)
    D = int(offsets[-1]) if tiles.numel() else 0
    rminx, rminy, rmaxx, rmaxy = pre['rect']
    if D == 0:
        z = torch.zeros(0, dtype=torch.int64)
        return dict(D=0, offsets=offsets, keys=z, vals=z, keys_sorted=z, point_list=z,
                    ranges=torch.zeros(gx * gy, 2, dtype=torch.int64))
    dbits = pre['depth'].detach().to(torch.float32).view(torch.int32).to(torch.int64)
    # instance k of Gaussian i sits at offsets[i-1]+k and covers rect cell (row-major) k
    vals = torch.repeat_interleave(torch.arange(tiles.numel()), tiles)
    start = (offsets - tiles)[vals]
    k = torch.arange(D) - start
    wrect = (rmaxx - rminx)[vals]
    ty = rminy[vals] + k // wrect
    tx = rminx[vals] + k % wrect
    keys = ((ty * gx + tx) << 32) | dbits[vals]
    order = torch.sort(keys, stable=True).indices
    ks, pl = keys[order], vals[order]
    tile_of = ks >> 32
    ranges = torch.zeros(gx * gy, 2, dtype=torch.int64)
    first = torch.ones(D, dtype=torch.bool)
    first[1:] = tile_of[1:] != tile_of[:-1]
    last = torch.ones(D, dtype=torch.bool)
    last[:-1] = first[1:]
    idx = torch.arange(D)
    ranges[tile_of[first], 0] = idx[first]
    ranges[tile_of[last], 1] = idx[last] + 1
    return dict(D=D, offsets=offsets, keys=keys, vals=vals, keys_sorted=ks, point_list=pl, ranges=ranges)


def cubemap_fetch(uv, texture):
    """uv (...,3) not necessarily unit; texture [6,R,R,3].  Face = argmax |component| (ties x>y>z),
    order +x,-x,+y,-y,+z,-z; in-face (col,row) inverts NVDIFFREC/util.py:94-101; texel centres at
    (i+0.5)/R; bilinear, clamp-to-edge inside the face.  Returns (rgb, face_margin)."""
    R = texture.shape[1]
    x, y, z = uv[..., 0], uv[..., 1], uv[..., 2]
    ax, ay, az = x.abs(), y.abs(), z.abs()
    isx = (ax >= ay) & (ax >= az)
    isy = (~isx) & (ay >= az)
    isz = ~(isx | isy)
    ma = torch.where(isx, ax, torch.where(isy, ay, az))
    ma = torch.clamp(ma, min=MA_MIN)
    sc = torch.where(isx, torch.where(x >= 0, -z, z), torch.where(isy, x, torch.where(z >= 0, x, -x)))
    tc = torch.where(isx, -y, torch.where(isy, torch.where(y >= 0, z, -z), -y))
    face = torch.where(isx, torch.where(x >= 0, 0, 1),
                       torch.where(isy, torch.where(y >= 0, 2, 3), torch.where(z >= 0, 4, 5)))
    col = (sc / ma + 1.0) * (0.5 * R) - 0.5
    row = (tc / ma + 1.0) * (0.5 * R) - 0.5
    x0 = torch.floor(col.detach())
    y0 = torch.floor(row.detach())
    fx = col - x0
    fy = row - y0
    x0i = x0.to(torch.int64)
    y0i = y0.to(torch.int64)
    x0c, x1c = x0i.clamp(0, R - 1), (x0i + 1).clamp(0, R - 1)
    y0c, y1c = y0i.clamp(0, R - 1), (y0i + 1).clamp(0, R - 1)
    t00 = texture[face, y0c, x0c]
    t01 = texture[face, y0c, x1c]
    t10 = texture[face, y1c, x0c]
    t11 = texture[face, y1c, x1c]
    fx_, fy_ = fx[..., None], fy[..., None]
    tex = ((1 - fx_) * (1 - fy_)) * t00 + (fx_ * (1 - fy_)) * t01 + ((1 - fx_) * fy_) * t10 + (fx_ * fy_) * t11
    srt = torch.sort(torch.stack([ax, ay, az], -1).detach(), dim=-1).values
    margin = (srt[..., 2] - srt[..., 1]) / torch.clamp(srt[..., 2], min=MA_MIN)
    return tex, margin


def render(pre, binning, texture, st, dtype, chunk=128, tile_subset=None, extra_attrs=None):
    """K6 of SURVEY Appendix A.4, tile by tile, instance chunks processed front to back.  `extra_attrs` [N,C] (the lineage
    operator's 10th kwarg, render/uv_tex_render.py:66): blended like depth / normals, extra_c = sum_i w_i e_ic -- channels 8.. of out."""
    H, W = int(st.image_height), int(st.image_width)
    gx, gy = pre['grid']
    bg = st.bg.to(dtype)
    CE = 0 if extra_attrs is None else int(extra_attrs.shape[1])
    out = torch.zeros(8 + CE, H, W, dtype=dtype)       # r,g,b,depth,nx,ny,nz,alpha (, extra...)
    final_T = torch.ones(H, W, dtype=dtype)
    n_contrib = torch.zeros(H, W, dtype=torch.int64)
    ambiguity = torch.full((H, W), float('inf'), dtype=dtype)
    out_tiles = {}
    pl = binning['point_list']
    ranges = binning['ranges']
    ly, lx = torch.meshgrid(torch.arange(TILE), torch.arange(TILE), indexing='ij')
    for tile in (range(gx * gy) if tile_subset is None else tile_subset):
        r0, r1 = int(ranges[tile, 0]), int(ranges[tile, 1])
        ty0, tx0 = (tile // gx) * TILE, (tile % gx) * TILE
        py = (ty0 + ly).reshape(-1)
        px = (tx0 + lx).reshape(-1)
        inside = (py < H) & (px < W)
        npx = py.numel()
        pix = torch.stack([px, py], 1).to(dtype)
        T = torch.ones(npx, dtype=dtype)
        acc = torch.zeros(npx, 8 + CE, dtype=dtype)
        done = ~inside
        ncon = torch.zeros(npx, dtype=torch.int64)
        amb = torch.full((npx,), float('inf'), dtype=dtype)
        pos = r0
        while pos < r1 and not bool(done.all()):
            ids = pl[pos:min(pos + chunk, r1)]
            K = ids.numel()
            d = pre['xy'][ids][None, :, :] - pix[:, None, :]             # (P,K,2)  d = xy - pix
            con = pre['conic'][ids]
            power = -0.5 * (con[None, :, 0] * d[..., 0] * d[..., 0] + con[None, :, 2] * d[..., 1] * d[..., 1]) \
                    - con[None, :, 1] * d[..., 0] * d[..., 1]
            araw = pre['opacity'][ids][None, :] * torch.exp(power)
            alpha = araw + (torch.clamp(araw, max=ALPHA_MAX) - araw).detach()   # straight-through clamp
            ok = (power <= 0) & (alpha >= ALPHA_MIN)
            alpha_eff = torch.where(ok, alpha, torch.zeros_like(alpha))
            # sequential semantics inside the chunk
            Tafter = T[:, None] * torch.cumprod(1.0 - alpha_eff, dim=1)
            Tbefore = torch.cat([T[:, None], Tafter[:, :-1]], dim=1)
            alive = (Tafter.detach() >= T_EPS) & (~done[:, None])       # monotone in k
            contrib = ok & alive
            w = torch.where(contrib, alpha_eff * Tbefore, torch.zeros_like(alpha))
            # decision margins (relative) for the parity tests' ambiguity mask
            with torch.no_grad():
                m1 = (alpha - ALPHA_MIN).abs() / ALPHA_MIN
                m2 = torch.where(ok, (Tafter - T_EPS).abs() / T_EPS, torch.full_like(Tafter, float('inf')))
                m3 = torch.where(power.abs() < 1e-5, torch.zeros_like(power), torch.full_like(power, float('inf')))
                seen = (~done[:, None]) & torch.cat([torch.ones(npx, 1, dtype=torch.bool), alive[:, :-1]], dim=1)
                mm = torch.minimum(torch.minimum(m1, m2), m3)
                mm = torch.where(seen, mm, torch.full_like(mm, float('inf')))
            # texture branch
            dp = -d
            gk = pre['g'][ids]
            Gk = pre['G'][ids]
            den = 1.0 + (gk[None, :, :] * dp).sum(-1)
            num = (Gk[None, :, :, :] * dp[:, :, None, :]).sum(-1)         # (P,K,3)
            good = den.detach() >= DEN_MIN
            den_s = torch.where(good, den, torch.ones_like(den))
            uv = pre['phi'][ids][None, :, :] + torch.where(good[..., None], num / den_s[..., None], torch.zeros_like(num))
            tex, fmargin = cubemap_fetch(uv, texture)
            col = torch.clamp(SH_C0 * tex + pre['viewdep'][ids][None, :, :] + 0.5, min=0.0)
            with torch.no_grad():
                m4 = torch.where(contrib, torch.minimum(fmargin, (den - DEN_MIN).abs()), torch.full_like(fmargin, float('inf')))
                amb = torch.minimum(amb, torch.minimum(mm, m4).min(dim=1).values)
            feat = torch.cat([col,
                              pre['depth'][ids][None, :, None].expand(npx, K, 1),
                              pre['normal'][ids][None, :, :].expand(npx, K, 3),
                              torch.ones(npx, K, 1, dtype=dtype)], dim=-1)
            if CE:
                feat = torch.cat([feat, extra_attrs[ids][None, :, :].expand(npx, K, CE)], dim=-1)
            acc = acc + (w[..., None] * feat).sum(1)
            # bookkeeping: last contributor position (1-based within the tile list)
            with torch.no_grad():
                kidx = torch.arange(1, K + 1)[None, :].expand(npx, K)
                lastk = torch.where(contrib, kidx, torch.zeros_like(kidx)).max(dim=1).values
                ncon = torch.where(lastk > 0, (pos - r0) + lastk, ncon)
                newly_done = (~alive[:, -1]) & (~done)
            # T after the last contributing Gaussian of this chunk
            has = contrib.any(dim=1)
            # last contributing Tafter = min over contributing (monotone decreasing)
            big = torch.full_like(Tafter, 2.0)
            Tlast = torch.where(contrib, Tafter, big).min(dim=1).values
            T = torch.where(has, Tlast, T)
            done = done | newly_done
            pos += K
        img = acc[:, :3] + T[:, None] * bg[None, :]
        full = torch.cat([img, acc[:, 3:]], dim=1)
        out_tiles[tile] = (py[inside], px[inside], full[inside], T[inside], ncon[inside], amb[inside])
    # assemble (index_put keeps autograd)
    if out_tiles:
        PY = torch.cat([v[0] for v in out_tiles.values()])
        PX = torch.cat([v[1] for v in out_tiles.values()])
        VAL = torch.cat([v[2] for v in out_tiles.values()])
        out = _scatter_image(out, PY, PX, VAL)
        final_T[PY, PX] = torch.cat([v[3] for v in out_tiles.values()]).detach()
        n_contrib[PY, PX] = torch.cat([v[4] for v in out_tiles.values()])
        ambiguity[PY, PX] = torch.cat([v[5] for v in out_tiles.values()])
    return out, final_T, n_contrib, ambiguity


def _scatter_image(out, PY, PX, VAL):
    C, H, W = out.shape
    flat = torch.zeros(C, H * W, dtype=out.dtype)
    idx = (PY * W + PX)
    flat = flat.index_copy(1, idx, VAL.t())
    return flat.reshape(C, H, W)


def rasterize(means3D, means2D, shs, opacities, scales, rotations, uvs, gradient_uvs, texture,
              st: Settings, dtype=torch.float64, debug=False, color_offset=None, cov3D_precomp=None, extra_attrs=None):
    """Whole operator (reference call: render/uv_tex_render.py:56-66).  Returns
    (image[3,H,W], depth[1,H,W], norm[3,H,W], alpha[1,H,W], radii[N] int32, extra[C,H,W] or None) and, with
    debug=True, a dict of intermediates for the integer-stage parity tests."""
    cv = lambda t: None if t is None else t.to(dtype)
    pre = preprocess(cv(means3D), cv(means2D), cv(shs), cv(opacities), cv(scales), cv(rotations),
                     cv(uvs), cv(gradient_uvs), st, dtype, color_offset=cv(color_offset), cov3D_precomp=cv(cov3D_precomp))
    binning = bin_and_sort(pre)
    out, final_T, n_contrib, amb = render(pre, binning, cv(texture), st, dtype, extra_attrs=cv(extra_attrs))
    res = (out[0:3], out[3:4], out[4:7], out[7:8], pre['radius'].to(torch.int32), None if extra_attrs is None else out[8:])
    if debug:
        return res, dict(pre=pre, binning=binning, final_T=final_T, n_contrib=n_contrib, ambiguity=amb)
    return res
