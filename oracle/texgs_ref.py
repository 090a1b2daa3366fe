"""TEST INFRASTRUCTURE ONLY -- ctypes wrapper of oracle/libtexgs_ref.so (the plain-C restatement, texgs_ref.c).
Loaded only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg."""
import ctypes as C
import math
import os
import subprocess

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libtexgs_ref.so")
LIB_VARIANT = os.path.join(HERE, "libtexgs_ref_variant.so")       # differently-rounded build of the blend loops (Makefile)
REC = 24


class RefIn(C.Structure):
    _fields_ = [("H", C.c_int), ("W", C.c_int), ("N", C.c_int), ("K", C.c_int), ("R", C.c_int), ("sh_degree", C.c_int),
                ("tanfovx", C.c_float), ("tanfovy", C.c_float), ("scale_modifier", C.c_float)] + \
               [(n, C.c_void_p) for n in ["bg", "V", "P", "cam", "means", "shs", "opac", "scales", "rots", "uvs", "juv", "tex",
                                          "coff", "cov3d"]]


_lib = None
_lib_variant = None


def load_variant():
    global _lib_variant
    if _lib_variant is None:
        load()
        _lib_variant = C.CDLL(LIB_VARIANT)
    return _lib_variant


def load():
    global _lib
    if _lib is None:
        src_t = os.path.getmtime(os.path.join(HERE, "texgs_ref.c"))
        if not all(os.path.exists(f) and os.path.getmtime(f) >= src_t for f in (LIB, LIB_VARIANT)):
            subprocess.check_call(["make", "-s", "-C", HERE])
        _lib = C.CDLL(LIB)
        _lib.texgs_ref_preprocess.restype = C.c_uint32
        _lib.texgs_ref_num_threads.restype = C.c_int
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(t):
    return None if t is None else np.ascontiguousarray(t.detach().cpu().numpy().astype(np.float32))


class RefRun:
    """One forward (and optionally backward) of the C oracle.  All arrays are numpy, float32 / integer.

    `scene`: anything with the attributes of texgs.synth.Scene.  The untextured surface (`diff_gauss`, render/render.py:75-84):
    scene.texture None (uvs / gradient_uvs then unused), optional scene.color_offset [N,3] (C0 * SH_DC, or colors_precomp - 0.5)
    and scene.cov3D_precomp [N,6] instead of scales / rotations."""

    def __init__(self, scene, st, threads=0):
        lib = load()
        if threads:
            lib.texgs_ref_set_threads(int(threads))
        self.lib = lib
        self.N = scene.means3D.shape[0]
        self.K = 0 if scene.shs is None else scene.shs.shape[1]
        tex = getattr(scene, "texture", None)
        coff, cov = getattr(scene, "color_offset", None), getattr(scene, "cov3D_precomp", None)
        self.textured, self.has_coff, self.has_cov = tex is not None, coff is not None, cov is not None
        assert not (self.textured and self.has_cov), "cov3D_precomp is an input of the untextured surface only"
        self.R = tex.shape[1] if self.textured else 1
        self.H, self.W = int(st.image_height), int(st.image_width)
        self.arr = dict(bg=_f32(st.bg), V=_f32(st.viewmatrix), P=_f32(st.projmatrix), cam=_f32(st.campos),
                        means=_f32(scene.means3D), shs=_f32(scene.shs), opac=_f32(scene.opacities.reshape(-1)),
                        scales=None if self.has_cov else _f32(scene.scales), rots=None if self.has_cov else _f32(scene.rotations),
                        uvs=_f32(scene.uvs) if self.textured else None, juv=_f32(scene.gradient_uvs) if self.textured else None,
                        tex=_f32(tex), coff=_f32(coff), cov3d=_f32(cov))
        a = self.arr
        self.inp = RefIn(self.H, self.W, self.N, self.K, self.R, int(st.sh_degree), float(st.tanfovx), float(st.tanfovy),
                         float(st.scale_modifier), _p(a["bg"]), _p(a["V"]), _p(a["P"]), _p(a["cam"]), _p(a["means"]),
                         _p(a["shs"]), _p(a["opac"]), _p(a["scales"]), _p(a["rots"]), _p(a["uvs"]), _p(a["juv"]),
                         _p(a["tex"]), _p(a["coff"]), _p(a["cov3d"]))
        self.T = ((self.W + 15) // 16) * ((self.H + 15) // 16)

    def _grad_arrays(self):
        N, K = self.N, self.K
        z = lambda *s: np.zeros(s, np.float32)
        return dict(means3D=z(N, 3), means2D=z(N, 3), shs=z(N, K, 3) if K else None, opacities=z(N, 1),
                    scales=None if self.has_cov else z(N, 3), rotations=None if self.has_cov else z(N, 4),
                    uvs=z(N, 3) if self.textured else None, color_offset=z(N, 3) if self.has_coff else None,
                    cov3D=z(N, 6) if self.has_cov else None)

    def _k8(self, acc, g):
        self.lib.texgs_ref_preprocess_bwd(C.byref(self.inp), _p(self.radii), _p(acc), _p(g["means3D"]), _p(g["means2D"]),
                                          _p(g["shs"]), _p(g["opacities"]), _p(g["scales"]), _p(g["rotations"]), _p(g["uvs"]),
                                          _p(g["color_offset"]), _p(g["cov3D"]))
        return g

    def forward(self):
        N, H, W, T, lib = self.N, self.H, self.W, self.T, self.lib
        n1 = max(N, 1)
        self.rec = np.zeros((n1, REC), np.float32)
        self.depth = np.zeros(n1, np.float32)
        self.radii = np.zeros(n1, np.int32)
        self.rect = np.zeros((n1, 4), np.int32)
        self.tiles = np.zeros(n1, np.uint32)
        self.offsets = np.zeros(n1, np.uint32)
        self.D = int(lib.texgs_ref_preprocess(C.byref(self.inp), _p(self.rec), _p(self.depth), _p(self.radii),
                                              _p(self.rect), _p(self.tiles), _p(self.offsets))) if N else 0
        d1 = max(self.D, 1)
        self.keys_unsorted = np.zeros(d1, np.uint64)
        self.vals_unsorted = np.zeros(d1, np.uint32)
        self.keys_sorted = np.zeros(d1, np.uint64)
        self.point_list = np.zeros(d1, np.uint32)
        self.ranges = np.zeros((T, 2), np.uint32)
        lib.texgs_ref_bin(C.byref(self.inp), C.c_uint32(self.D), _p(self.depth), _p(self.rect), _p(self.tiles),
                          _p(self.offsets), _p(self.keys_unsorted), _p(self.vals_unsorted), _p(self.keys_sorted),
                          _p(self.point_list), _p(self.ranges))
        self.out = np.zeros((8, H, W), np.float32)
        self.final_T = np.ones((H, W), np.float32)
        self.n_contrib = np.zeros((H, W), np.uint32)
        lib.texgs_ref_render_fwd(C.byref(self.inp), _p(self.rec), _p(self.point_list), _p(self.ranges), _p(self.out),
                                 _p(self.final_T), _p(self.n_contrib))
        return self.out

    def backward(self, dout, tau_cell=0.0, cell_weight=1.0, margin=None, tau_fwd=0.0, tau_relu=0.0, cond_weight=0.0):
        """dout: float32 [8,H,W] (r,g,b,depth,nx,ny,nz,alpha).  Returns dict of input gradients.
        tau_cell > 0: also collects self.fmass [N,24] -- per Gaussian, the absolute mass of the terms that hang on a bilinear cell
        choice within tau_cell texels of flipping (texgs_ref_render_bwd_ex) -- for cell_edge_deviation().  margin [H,W] (from
        ambiguity()) + tau_fwd: also 2/255 of every term of the pairs in forward-ambiguous pixels (a marginal contributor flipping
        moves the others' transmittance by 1/255).  tau_relu > 0: also the all-or-nothing colour gradient of pairs whose
        max(0, .) argument is within tau_relu of zero (then ambiguity() need not flag their rows: pass it tau_relu=0).
        cond_weight > 0 (with tau_cell > 0 so that fmass exists): also the CONDITIONING mass of every pair -- how far its terms move under
        the fp32 rounding of its own falloff exponent, of the transmittance in front of it and of the blend behind it
        (texgs_ref.c: texgs_ref_set_cond_weight) -- times cond_weight."""
        N, K, R, lib = self.N, self.K, self.R, self.lib
        dout = np.ascontiguousarray(dout.astype(np.float32))
        acc = np.zeros((max(N, 1), REC), np.float64)
        dtex = np.zeros((6, R, R, 3), np.float32) if self.textured else None
        self.fmass = np.zeros((max(N, 1), REC), np.float64) if tau_cell > 0 else None
        slope = float(tau_relu)          # (the C argument `tex_slope` carries tau_relu of the pair-level colour-clamp treatment; 0 = off)
        lib.texgs_ref_set_cond_weight(C.c_float(float(cond_weight) if self.fmass is not None else 0.0))
        lib.texgs_ref_render_bwd_ex(C.byref(self.inp), _p(self.rec), _p(self.point_list), _p(self.ranges), _p(self.final_T),
                                    _p(self.n_contrib), _p(dout), _p(acc), _p(dtex), C.c_float(tau_cell), C.c_float(cell_weight),
                                    C.c_float(slope), _p(self.fmass),
                                    _p(None if margin is None else np.ascontiguousarray(margin, np.float32)), C.c_float(tau_fwd))
        lib.texgs_ref_set_cond_weight(C.c_float(0.0))
        g = self._k8(acc, self._grad_arrays())
        g["texture"] = dtex
        self.acc = acc
        return g

    def cell_edge_deviation(self):
        """After backward(tau_cell > 0): {name: float64[N]} -- per Gaussian and per-Gaussian output, a bound on how far the row
        moves when the uv-derivative terms of its near-cell-edge pairs change by their own size: the last stage is LINEAR in the
        per-Gaussian sums, so the collected masses are pushed through it slot by slot and the absolute responses added up (an L1
        bound: no cancellation assumed), maximum over the row's entries."""
        assert self.fmass is not None, "run backward(dout, tau_cell=...) first"
        tot = {}
        for k in np.nonzero(self.fmass.any(axis=0))[0]:
            one = np.zeros_like(self.fmass)
            one[:, k] = self.fmass[:, k]
            for n, v in self._k8(one, self._grad_arrays()).items():
                if v is None:
                    continue
                a = np.abs(v.reshape(self.N, -1).astype(np.float64))
                tot[n] = tot[n] + a if n in tot else a
        if not tot:
            return {n: np.zeros(self.N) for n, v in self._grad_arrays().items() if v is not None}
        return {n: a.max(1) for n, a in tot.items()}

    def _preprocess_bwd(self, acc):
        return self._k8(acc, self._grad_arrays())

    def accumulation_sensitive(self, delta=1e-6, trials=8, frac=0.1, row_rtol=1e-3, row_atol_frac=1e-4, seed=0):
        """After backward(): bool[N] -- Gaussians whose input gradients are ill-conditioned functions of their per-Gaussian sums:
        re-running the last stage (texgs_ref_preprocess_bwd, fp32 arithmetic) on the sums perturbed by `delta` relative (what an
        fp32 accumulation in another order does to them; this oracle accumulates in fp64) moves some entry by more than `frac` of the
        row tolerance the parity tests use (row_rtol * |row| + row_atol_frac * largest entry).  These are needle-shaped splats
        (a scale of 1e-9 beside one of 1e-2): the conic -> covariance -> scale chain cancels to 1e-4 of its terms and any
        implementation's rounding re-rolls the result.  ~110 of 10^6 Gaussians at C5, only dL/dscales; a handful at C3."""
        rng = np.random.default_rng(seed)
        base = self._preprocess_bwd(self.acc)
        names = [n for n in ("means3D", "means2D", "opacities", "scales", "rotations", "uvs", "cov3D") if base[n] is not None]
        tol = {}
        for n in names:
            e = np.abs(base[n].reshape(self.N, -1).astype(np.float64))
            tol[n] = row_rtol * e.max(1) + row_atol_frac * max(float(e.max()), 1e-300)
        worst = np.zeros(self.N)
        for _ in range(trials):
            g = self._preprocess_bwd(self.acc * (1.0 + delta * rng.standard_normal(self.acc.shape)))
            for n in names:
                d = np.abs(g[n].reshape(self.N, -1).astype(np.float64) - base[n].reshape(self.N, -1).astype(np.float64)).max(1)
                worst = np.maximum(worst, d / tol[n])
        return worst > frac

    def variant_render(self, dout=None):
        """The blend loops of the differently-rounded build on THIS run's records / lists: (out[8,H,W], grads or None)."""
        lib = load_variant()
        out = np.zeros((8, self.H, self.W), np.float32)
        fT = np.ones((self.H, self.W), np.float32)
        nc = np.zeros((self.H, self.W), np.uint32)
        lib.texgs_ref_render_fwd(C.byref(self.inp), _p(self.rec), _p(self.point_list), _p(self.ranges), _p(out), _p(fT), _p(nc))
        if dout is None:
            return out, nc, None
        N, K, R = self.N, self.K, self.R
        dout = np.ascontiguousarray(dout.astype(np.float32))
        acc = np.zeros((max(N, 1), REC), np.float64)
        dtex = np.zeros((6, R, R, 3), np.float32) if self.textured else None
        lib.texgs_ref_render_bwd(C.byref(self.inp), _p(self.rec), _p(self.point_list), _p(self.ranges), _p(fT), _p(nc), _p(dout),
                                 _p(acc), _p(dtex))
        g = self._k8(acc, self._grad_arrays())
        g["texture"] = dtex
        return out, nc, g

    def ambiguity(self, tau_fwd=2e-5, tau_cell=1e-4, tau_relu=1e-5, own_only=False):
        """After forward(): (margin[H,W], gflag[N] bool, tflag[6,R,R] bool) -- where two fp32 implementations of the operator may
        legitimately differ by more than rounding (texgs_ref_ambiguity in texgs_ref.c says which decisions are looked at)."""
        margin = np.full((self.H, self.W), np.inf, np.float32)
        gflag = np.zeros(max(self.N, 1), np.uint8)
        tflag = np.zeros(6 * self.R * self.R, np.uint8)          # (untextured: R = 1, never written)
        # self.cond [H,W]: rounding error of the falloff exponent carried to the pixel's blend weights, in units of one fp32 rounding
        # of a value of size 1 (texgs_ref_ambiguity_ex): ~1e-7 on the benchmark scenes, 1e-5..1e-4 under splats hundreds of pixels wide
        self.cond = np.zeros((self.H, self.W), np.float32)
        self.lib.texgs_ref_ambiguity_ex(C.byref(self.inp), _p(self.rec), _p(self.point_list), _p(self.ranges),
                                        C.c_float(tau_fwd), C.c_float(tau_cell), C.c_float(tau_relu), _p(margin), _p(gflag), _p(tflag),
                                        _p(self.cond), C.c_int(1 if own_only else 0))
        return margin, gflag[:self.N].astype(bool), tflag.reshape(6, self.R, self.R).astype(bool)

    @property
    def threads(self):
        return int(self.lib.texgs_ref_num_threads())
