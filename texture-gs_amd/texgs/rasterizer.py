"""Host side of the textured Gaussian rasterizer: torch.autograd.Function over the C ABI of libtexgs.so.

Mirrors the operator surface the reference imports at render/uv_tex_render.py:4 and calls at :25-38, :40,
:56-66 (GaussianRasterizationSettings / GaussianRasterizer): same names, argument meaning and 6-tuple return.
PyTorch is plumbing here (device memory through the caching allocator, the current HIP stream, autograd
bookkeeping); all arithmetic runs in the hand-written gfx950 kernels.
"""
import ctypes as C
import math
from typing import NamedTuple, Optional

import torch
from torch import nn

from . import _lib


import os as _os
USE_TEX_BINS = _os.environ.get("TEXGS_TEX_BINS", "1") != "0"       # binned two-pass texture gradient
TEX_REC_CAP = int(_os.environ.get("TEXGS_REC_CAP", "0"))           # fixed record capacity (tests); 0 = adaptive
# OPT-IN fast path: add into a leaf's existing .grad in place instead of handing autograd a temporary.  Off by default because
# it is not what autograd does: torch.autograd.grad() would then mutate .grad, and tensor hooks / post-accumulate-grad hooks /
# DDP reducer hooks on those leaves would not fire (leaves that carry hooks are skipped even when it is on).  The supported fused
# path is an explicit texgs.multiview.GradBucket sink.
DIRECT_LEAF_GRADS = _os.environ.get("TEXGS_DIRECT_LEAF_GRADS", "0") != "0"
# upper bound of ONE record buffer; there is one per (device, stream) that runs backwards (a depth-3 ViewPipeline holds three)
TEX_REC_BYTES_MAX = int(float(_os.environ.get("TEXGS_BIN_GB_MAX", "4")) * (1 << 30))
# per-Gaussian gradient outputs in the order / bit positions of TEXGS_ACC_* (texgs.h)
_ACC_BITS = dict(means3D=1, means2D=2, shs=4, opacities=8, scales=16, rotations=32, uvs=64, color_offset=128, cov3D=256)


class GaussianRasterizationSettings(NamedTuple):
    """Field names and order = keywords at render/uv_tex_render.py:25-38."""
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


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _f32c(t: torch.Tensor, name: str, device) -> torch.Tensor:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if t.device != device:
        raise ValueError(f"{name} is on {t.device}, expected {device}")
    if t.dtype != torch.float32:
        raise TypeError(f"{name} must be float32, got {t.dtype}")
    return t.contiguous()


# Two forwards of the same geometry share the tile binning and K6's survivor lists (the reference renders every training view
# twice: models/texture_gaussian3d.py:318 and :375-389 -- same camera, same Gaussians, sh_degree 0 the second time; visual_step
# :499-511 likewise).  Decided by K1's geometry fingerprint (texgs.h texgs_read_num_rendered2), so recomputed activations
# (sigmoid / exp outputs are new tensors on every render() call) share too.  One entry per (device, stream).
GEOM_CACHE = _os.environ.get("TEXGS_GEOM_CACHE", "1") != "0"
# A forward that autograd may or may not differentiate (retexture.py:27 / visual_step build a graph and never call backward)
# leaves K6's hand-off to the backward once two graphs in a row were dropped unused; the first backward that does arrive
# produces the hand-off itself (one more K6) and switches the forwards back.
LAZY_HANDOFF = _os.environ.get("TEXGS_LAZY_HANDOFF", "1") != "0"
_CAPACITY_HINT = {}
# K6 -> K7 item stream (texgs.h v15), OPT-IN (TEXGS_ITEMS=1): K6 leaves one 12-byte item per contributing (pixel, Gaussian) pair and
# K7 walks those instead of re-testing the survivor lists.  Built, parity-tested and measured in round 6: K7 executes 21 % fewer
# VALU instructions (176 M vs 222 M wave-instructions at C3) and takes the same time (the blend kernels wait on memory-side traffic,
# not on issue), while K6 pays 45 us for writing 0.22 GB of items -- 935 vs 981 views/s, so the survivor replay stays the default
# (DESIGN.md section 5.1.34).  The page buffer is sized from what earlier views of the same size needed (the sub-pool cursors of a
# forward are copied to pinned memory asynchronously and looked at by a later call -- never waited for); a buffer that is too small
# costs speed, not correctness (the survivor-replay kernel runs instead).
USE_ITEMS = _os.environ.get("TEXGS_ITEMS", "0") != "0"
ITEM_PAGES_FIXED = int(_os.environ.get("TEXGS_ITEM_PAGES", "0"))     # fixed page count (tests: force the fallback); 0 = adaptive
_ITEM_HINT = {}             # (device index, N, H, W) -> pages (1.25 x the most a view needed)
_ITEM_STAT = {}             # (device index, stream) -> _ItemStat


class _ItemStat:
    __slots__ = ("host", "event", "key", "pools")

    def __init__(self):
        self.host = torch.zeros(_lib.ITEM_CTL_WORDS, dtype=torch.int32).pin_memory()
        self.event = None
        self.key = None
        self.pools = 1

    def poll(self):
        if self.event is not None and self.event.query():
            self.event = None
            wanted = int(self.host[0:16 * self.pools:16].max()) * self.pools
            _ITEM_HINT[self.key] = max(_ITEM_HINT.get(self.key, 0), int(wanted * 1.25) + 8 * self.pools)

    def watch(self, ctl, key, pools, device):
        if self.event is None:
            self.host.copy_(ctl, non_blocking=True)
            self.key, self.pools = key, pools
            self.event = torch.cuda.Event()
            self.event.record(torch.cuda.current_stream(device))


def _item_layout(hint_key, cap, tiles):
    """(pages, sub-pools) of a forward's item buffer: >= 64 blocks per sub-pool (their demands average out), <= 64 sub-pools (one
    hot atomic word serialises at ~13 ns per page)."""
    blocks = 4 * tiles
    pools = 1
    while pools < _lib.ITEM_MAX_POOLS and pools * 2 * 64 <= blocks:
        pools *= 2
    pages = ITEM_PAGES_FIXED or _ITEM_HINT.get(hint_key) or (20 * cap) // _lib.ITEM_PAGE + 2 * blocks
    pages = max(int(pages), 2 * pools)
    return (pages + pools - 1) // pools * pools, pools
_GEOM = {}                  # (device index, stream) -> _GeomEntry of the last forward that built lists there
# the hand-off predictor: autograd forwards are numbered; per device, the highest number whose backward ran and the numbers of
# those dropped without one.  (Order-independent: a state that the garbage collector frees late cannot reset the streak.)
_SERIAL = [0]
_LAST_DIFFERENTIATED = {}
_DROPPED = {}
import threading as _threading
_PRED_LOCK = _threading.RLock()       # the predictor's dicts are touched from __del__ (any thread, any time the collector runs)


def unused_streak(dev_index):
    """Autograd forwards on this device that were dropped without a backward since the last one that was differentiated."""
    with _PRED_LOCK:
        last = _LAST_DIFFERENTIATED.get(dev_index, -1)
        return sum(1 for n in _DROPPED.get(dev_index, ()) if n > last)


def reset_handoff_predictor():
    with _PRED_LOCK:
        _LAST_DIFFERENTIATED.clear()
        _DROPPED.clear()


class _GeomEntry:
    __slots__ = ("ints", "cam_key", "fingerprint", "D", "bin", "arenas", "handoff", "counts", "cap", "items")


def geometry_cache_stats():
    """{'hits': n, 'misses': n} of the shared-geometry path since import (diagnostics / tests)."""
    return dict(_GEOM_STATS)


_GEOM_STATS = {"hits": 0, "misses": 0, "late_handoffs": 0}
# Backward scratch, ONE entry per (device, HIP stream): the moment accumulators (grow-only, sliced [:N]; all-zero between
# calls: K8 clears what it read) and the texture-gradient bins of the last resolution used on that stream.  A stream runs
# its views in order, so its scratch is never shared by two views in flight.  release_scratch() drops entries (ViewPipeline
# does that for its streams when it is closed / collected).
_SCRATCH = {}


class _StreamScratch:
    __slots__ = ("acc", "bins")

    def __init__(self):
        self.acc = None
        self.bins = None


def release_scratch(device=None, stream=None):
    """Free the backward scratch and the shared-geometry entry cached for (device, stream); None = every device / stream."""
    for cache in (_SCRATCH, _GEOM, _ITEM_STAT, _PREFETCH):
        for key in list(cache):
            if (device is None or key[0] == torch.device(device).index) and (stream is None or key[1] == int(stream)):
                del cache[key]


def scratch_bytes():
    """Bytes currently held by the module's caches (diagnostics / tests): the backward scratch per (device, stream) and the arenas
    the shared-geometry entries keep alive (the last list-building forward of every (device, stream))."""
    total = 0
    for sc in _SCRATCH.values():
        if sc.acc is not None:
            total += sc.acc.numel() * 4
        if sc.bins is not None:
            total += sc.bins.nbytes()
    for e in _GEOM.values():
        total += sum(ar.buf.numel() for ar in e.arenas if ar is not None and ar.buf is not None)
    return total


class _TexBins:
    """Scratch of the binned texture gradient for one (device, R, stream): the record buffer (TexGSGrads.tex_bins, 16 bytes per
    bilinear footprint of a view, exactly-sized contiguous lists), the per-bin fill cursors + two status words (all-zero
    between calls: the reduce kernel clears what it read) and the list offsets of the call in flight.  The buffer grows to
    what the views need: every call leaves the number of records its lists wanted in the word after the cursors; it is copied
    to pinned host memory asynchronously every few calls and looked at only when that copy has completed -- the backward
    never waits for it.  A buffer that is too small is not an error (what does not fit goes through atomics), so a stale
    size costs speed, never correctness."""

    def __init__(self, lib, device, R):
        self.device, self.R = device, R
        self.nbins = int(lib.texgs_tex_bin_count(R))
        self.cursor = torch.zeros(self.nbins + 2, dtype=torch.int32, device=device)
        self.base = torch.empty(2 * self.nbins + 1, dtype=torch.int32, device=device)      # list offsets, then the reduce launch order
        self.cap = 0
        self.rec = None
        self.host_stat = torch.zeros(1, dtype=torch.int32).pin_memory()
        self.event = None
        self.calls = 0

    def nbytes(self):
        return ((0 if self.rec is None else self.rec.numel()) + self.cursor.numel() + self.base.numel()) * 4

    def _resize(self, cap):
        cap = max(1024, min(int(cap), TEX_REC_BYTES_MAX // (4 * _lib.TEXBIN_RECORD_FLOATS)))
        if cap != self.cap:
            self.rec = None
            self.rec = torch.empty(cap * _lib.TEXBIN_RECORD_FLOATS, dtype=torch.float32, device=self.device)
            self.cap = cap

    def before_call(self, D, counts=None):
        if TEX_REC_CAP:
            self._resize(TEX_REC_CAP)
            return
        if self.rec is None and counts is not None and not torch.cuda.is_current_stream_capturing():
            # FIRST backward on this (device, stream) -- and the first after release_scratch(): size the buffer from K6's exact
            # per-bin counts.  This is ONE blocking device->host read (ADVICE r5 asked to remove or document it): it waits for this
            # view's forward, which the backward is ordered behind anyway; afterwards the asynchronous statistic below adapts the
            # size and nothing here ever waits again.  A guess from D instead (tried: 24 records per instance) sends the excess
            # of large-splat scenes through atomics on exactly the calls tests and benchmarks look at first, and makes the first
            # call take another path than the following ones.  Under stream capture: the guess below, no read.
            self._resize(int(counts.sum().item() * 1.25) + 4096)
        if self.event is not None and self.event.query():
            self.event = None
            wanted = int(self.host_stat[0])
            if wanted > self.cap:
                self._resize(int(wanted * 1.25) + 4096)
        if self.rec is None:
            self._resize(24 * max(int(D), 1024))        # first guess: ~17 contributing pixels per (tile, Gaussian) instance at C3

    def after_call(self):
        self.calls += 1
        if self.event is None and not TEX_REC_CAP and (self.calls <= 4 or self.calls % 16 == 0):
            self.host_stat.copy_(self.cursor[self.nbins:self.nbins + 1], non_blocking=True)
            self.event = torch.cuda.Event()
            self.event.record(torch.cuda.current_stream(self.device))


class _State:
    """Everything one forward leaves behind for its backward (per call: no global scratch, two forwards may
    be alive before a backward, models/texture_gaussian3d.py:318,378,410)."""
    __slots__ = ("frame", "inputs", "geom", "bin", "img", "tensors", "N", "K", "R", "H", "W", "D", "cap", "tiles",
                 "want_counts", "want_items", "lazy", "backward_ran", "shared_geometry", "serial", "__weakref__")

    def __del__(self):          # the forwards' hand-off predictor (LAZY_HANDOFF): was this autograd forward ever differentiated?
        try:
            if getattr(self, "lazy", False) and not self.backward_ran:
                dev = self.tensors["keep"][0].device.index
                with _PRED_LOCK:
                    last = _LAST_DIFFERENTIATED.get(dev, -1)
                    _DROPPED[dev] = [n for n in _DROPPED.get(dev, []) if n > last][-8:] + [self.serial]
        except Exception:
            pass


def _make_frame(st: GaussianRasterizationSettings, N, K, R, device, keep):
    H, W = int(st.image_height), int(st.image_width)
    bg = _f32c(st.bg, "bg", device).reshape(-1)
    vm = _f32c(st.viewmatrix, "viewmatrix", device).reshape(-1)
    pm = _f32c(st.projmatrix, "projmatrix", device).reshape(-1)
    cp = _f32c(st.campos, "campos", device).reshape(-1)
    if bg.numel() != 3 or vm.numel() != 16 or pm.numel() != 16 or cp.numel() != 3:
        raise ValueError("bg/campos must have 3 elements and viewmatrix/projmatrix 16")
    keep.extend([bg, vm, pm, cp])
    f = _lib.Frame(H, W, float(st.tanfovx), float(st.tanfovy), float(st.scale_modifier), int(st.sh_degree),
                   int(K), int(R), int(N), 1 if st.debug else 0, _ptr(bg), _ptr(vm), _ptr(pm), _ptr(cp))
    return f


# ---- forward prefetch: K1 (+ the D readback, + K2) of a LATER view issued early ---------------------------------------------------
# A forward waits once for the device: D, the instance count that sizes K3..K6's launches.  On a stream that runs its views in order,
# that wait is for everything the stream still has queued -- the previous view's whole backward.  With a few views pipelined over a
# few streams the GPU stays busy meanwhile, but the host is then never more than a view or two ahead of the device: any host thread
# stall of a few milliseconds (a neighbour process on the box's cores) drains the queue.  prefetch_forward issues K1 and the readback
# of a view AHEAD of the backward that is about to be queued on the same stream; when that view's forward is called, with the same
# tensors (same storage, same version counters) and settings, its D arrived long ago and nothing waits.  One pending forward per key;
# a forward that does not match anything pending runs as usual.
_PREFETCH = {}             # (device index, stream) -> [(key, finish closure)]
_PIN_POOL = {}             # words -> [pinned int32 tensors]


def _pin_take(words):
    pool = _PIN_POOL.setdefault(words, [])
    return pool.pop() if pool else torch.empty(max(words, 1), dtype=torch.int32).pin_memory()


def _pin_give(t):
    pool = _PIN_POOL.setdefault(t.numel(), [])
    if len(pool) < 16:
        pool.append(t)


def _prefetch_key(st, tensors, flags):
    t = lambda x: None if x is None else (x.data_ptr(), x._version, tuple(x.shape), x.requires_grad)
    return (int(st.image_height), int(st.image_width), float(st.tanfovx), float(st.tanfovy), float(st.scale_modifier), int(st.sh_degree),
            bool(st.debug), t(st.bg), t(st.viewmatrix), t(st.projmatrix), t(st.campos)) + tuple(t(x) for x in tensors) + tuple(flags)


def _prefetch_take(device, key):
    lst = _PREFETCH.get((device.index, int(torch.cuda.current_stream(device).cuda_stream)))
    if not lst:
        return None
    for k, (key_k, fin) in enumerate(lst):
        if key_k == key:
            del lst[k]
            return fin
    return None


def prefetch_forward(st, means3D, shs, opacities, scales, rotations, uvs, gradient_uvs, texture, color_offset=None,
                     for_backward=True, cov3D_precomp=None, count_bins=None, lazy=False, want_items=None):
    """Begin a forward now -- K1, the asynchronous D readback, K2 -- on the current stream; a later forward_raw / GaussianRasterizer
    call on this stream with the same arguments (same tensor storages and versions, same settings, same flags) finishes it without
    waiting for the device.  At most four forwards are kept pending per stream (older ones are dropped: their K1 was wasted, nothing
    else).  Returns nothing."""
    fin = forward_raw(st, means3D, shs, opacities, scales, rotations, uvs, gradient_uvs, texture, color_offset, for_backward,
                      cov3D_precomp, count_bins, lazy, want_items, _begin_only=True)
    key = _prefetch_key(st, (means3D, shs, opacities, scales, rotations, uvs, gradient_uvs, texture, color_offset, cov3D_precomp),
                        (for_backward, count_bins, lazy, want_items))
    lst = _PREFETCH.setdefault((means3D.device.index, int(torch.cuda.current_stream(means3D.device).cuda_stream)), [])
    lst.append((key, fin))
    del lst[:-4]


def forward_raw(st, means3D, shs, opacities, scales, rotations, uvs, gradient_uvs, texture, color_offset=None,
                for_backward=True, cov3D_precomp=None, count_bins=None, lazy=False, want_items=None, _begin_only=False):
    """Run K1..K6.  Returns (outputs, state).  No autograd here.

    `lazy` (the autograd path sets it): the hand-off may be left to the backward (see LAZY_HANDOFF).

    `for_backward`: K6 leaves the per-block survivor lists its backward replays.  `count_bins` (default: for_backward and
    there is a texture): K6 also counts the texture-gradient footprints per texture bin (the exact list sizes of
    backward_raw's binned texture gradient); a caller that will not ask for dL/dtexture saves that work.
    `want_items` (default: for_backward): K6 also leaves its ITEM STREAM (one {T, alpha_raw, Gaussian | pixel} per contributing
    pair) for the backward's per-Gaussian stages; a caller that will only ask for dL/dtexture saves that work.
    `texture=None`: the untextured surface (diff_gauss, render/render.py:75-84): uvs / gradient_uvs may be None too.
    `cov3D_precomp` f32[N,6] (untextured surface only): used instead of scales / rotations."""
    lib = _lib.load()
    device = means3D.device
    if device.type != "cuda":
        raise RuntimeError("the textured rasterizer runs on an AMD GPU (torch device 'cuda' = HIP); "
                           f"got tensors on {device}. There is no CPU fallback.")
    if _PREFETCH and not _begin_only:        # was this very forward begun earlier (prefetch_forward)?  Then K1 ran long ago: finish it
        pend = _prefetch_take(device, _prefetch_key(st, (means3D, shs, opacities, scales, rotations, uvs, gradient_uvs, texture,
                                                         color_offset, cov3D_precomp), (for_backward, count_bins, lazy, want_items)))
        if pend is not None:
            return pend()
    N = means3D.shape[0]
    means3D = _f32c(means3D, "means3D", device)
    if means3D.dim() != 2 or means3D.shape[1] != 3:
        raise ValueError("means3D must be [N,3]")
    opacities = _f32c(opacities, "opacities", device)
    if opacities.numel() != N:
        raise ValueError(f"per-Gaussian inputs must have exactly N rows (N={N}; opacities {tuple(opacities.shape)})")
    if cov3D_precomp is not None:
        if texture is not None:
            raise ValueError("cov3D_precomp is an input of the untextured surface only")
        cov3D_precomp = _f32c(cov3D_precomp, "cov3D_precomp", device)
        if cov3D_precomp.shape != (N, 6):
            raise ValueError(f"cov3D_precomp must be [N,6], got {tuple(cov3D_precomp.shape)}")
        scales = rotations = None
    else:
        scales = _f32c(scales, "scales", device)
        rotations = _f32c(rotations, "rotations", device)
        if scales.shape != (N, 3) or rotations.shape != (N, 4):
            raise ValueError("per-Gaussian inputs must have exactly N rows "
                             f"(N={N}; scales {tuple(scales.shape)}, rotations {tuple(rotations.shape)})")
    if texture is not None:
        uvs = _f32c(uvs, "uvs", device)
        gradient_uvs = _f32c(gradient_uvs, "gradient_uvs", device)
        texture = _f32c(texture, "texture", device)
        if uvs.shape != (N, 3) or gradient_uvs.numel() != 9 * N:
            raise ValueError("per-Gaussian inputs must have exactly N rows "
                             f"(N={N}; uvs {tuple(uvs.shape)}, gradient_uvs {tuple(gradient_uvs.shape)})")
        if texture.dim() != 4 or texture.shape[0] != 6 or texture.shape[1] != texture.shape[2] or texture.shape[3] != 3:
            raise ValueError(f"texture must be [6,R,R,3], got {tuple(texture.shape)}")
        R = texture.shape[1]
    else:
        uvs = gradient_uvs = None
        R = 1
    if count_bins is None:
        count_bins = for_backward
    count_bins = bool(count_bins and for_backward and texture is not None)
    K = 0
    if shs is not None:
        shs = _f32c(shs, "shs", device)
        if shs.dim() != 3 or shs.shape[0] != N or shs.shape[2] != 3:
            raise ValueError(f"shs must be [N,K,3], got {tuple(shs.shape)}")
        K = shs.shape[1]
        if K > 15:
            raise ValueError("shs holds at most 15 view-dependent coefficients (degree 3)")
    if color_offset is not None:
        color_offset = _f32c(color_offset, "color_offset", device)
        if color_offset.shape != (N, 3):
            raise ValueError(f"color_offset must be [N,3], got {tuple(color_offset.shape)}")
    if int(st.sh_degree) < 0 or int(st.sh_degree) > 3:
        raise ValueError("sh_degree must be in [0,3]")
    H, W = int(st.image_height), int(st.image_width)
    tiles = ((W + _lib.TILE - 1) // _lib.TILE) * ((H + _lib.TILE - 1) // _lib.TILE)
    cur_stream = torch.cuda.current_stream(device)
    stream = cur_stream.cuda_stream
    keep = [means3D, shs, opacities, scales, rotations, uvs, gradient_uvs, texture, color_offset, cov3D_precomp]
    handoff = bool(for_backward) and not (lazy and LAZY_HANDOFF and unused_streak(device.index) >= 2)
    want_counts = bool(count_bins and USE_TEX_BINS)
    want_items = bool((for_backward if want_items is None else want_items) and for_backward and USE_ITEMS)

    with torch.cuda.device(device):
        frame = _make_frame(st, N, K, R, device, keep)
        inputs = _lib.Inputs(_ptr(means3D), _ptr(shs), _ptr(opacities), _ptr(scales), _ptr(rotations),
                             _ptr(uvs), _ptr(gradient_uvs), _ptr(texture), _ptr(color_offset), _ptr(cov3D_precomp))
        i32, f32, u8 = torch.int32, torch.float32, torch.uint8
        n1 = max(N, 1)
        # a forward that may share an earlier one's lists: same sizes, same camera tensors -- K1's fingerprint decides
        gints = (N, H, W, R if texture is not None else 0, float(st.tanfovx), float(st.tanfovy), float(st.scale_modifier),
                 cov3D_precomp is not None)
        cam_key = tuple((t.data_ptr(), t._version) for t in (st.viewmatrix, st.projmatrix, st.campos))
        gkey = (device.index, int(stream))
        entry = _GEOM.get(gkey) if GEOM_CACHE else None
        candidate = entry is not None and entry.ints == gints and entry.cam_key == cam_key
        if entry is not None and not candidate:     # another view / size: the old lists cannot be shared any more -- let go of their
            del _GEOM[gkey]                         # arenas before this forward allocates its own (they stay alive only while a
            entry = None                            # state that shares them does)
        # Everything the kernels keep between forward and backward lives in TWO allocations (one sized by N / the image, one by
        # the instance capacity): ~20 separate torch.empty calls were ~0.1 ms of host time per view.  Tensor views of the
        # pieces are made on demand (tests, diagnostics).
        scan_bytes = lib.texgs_scan_temp_bytes(N)
        fix = _Arena(device)
        fix.add("rec", (n1, _lib.REC_TEST_FLOATS), f32)          # test records: xy, conic, opacity, cull aids
        fix.add("rec_shade", (n1, _lib.REC_SHADE_FLOATS), f32)
        fix.add("depth", (n1,), f32)
        fix.add("rect", (n1, 2), i32)
        fix.add("tiles_touched", (n1,), i32)
        fix.add("offsets", (n1,), i32)
        fix.add("scan_temp", (scan_bytes,), u8)
        fix.add("final_T", (H, W), f32)
        fix.add("n_contrib", (H, W), i32)

        def add_lists(ar):              # per-tile pieces of the lists (the candidate path adds them only if it has to build lists)
            ar.add("ranges", (tiles, 2), i32)              # zero-filled by K3
            ar.add("tile_order", (tiles,), i32)
            if want_counts and handoff:
                ar.add("tex_bin_count", (2 * int(lib.texgs_tex_bin_count(R)),), i32)     # reserved | overflow footprints per bin
                ar.add("tex_bin_resv", (4 * tiles, _lib.RESV_WORDS), i32)      # K6's per-block reservations in the record lists
            if handoff:
                ar.add("surv_count", (4 * tiles,), i32)
        if not candidate:
            add_lists(fix)
        fix.commit()
        radii = torch.empty(N, dtype=i32, device=device)          # K1 writes every entry (0 for culled)
        geom = _lib.Geom(fix.ptr("rec"), fix.ptr("rec_shade"), fix.ptr("depth"), _ptr(radii), fix.ptr("rect"),
                         fix.ptr("tiles_touched"), fix.ptr("offsets"), fix.ptr("scan_temp"), scan_bytes)
        # everything is allocated BEFORE the one device->host sync, the D-sized buffers from a capacity hint
        # (largest D seen on this device x 1.25): K1 -> sync (K2 runs meanwhile) -> K3..K6 with little host work between
        out_color = torch.empty(3, H, W, dtype=f32, device=device)
        out_depth = torch.empty(1, H, W, dtype=f32, device=device)
        out_norm = torch.empty(3, H, W, dtype=f32, device=device)
        out_alpha = torch.empty(1, H, W, dtype=f32, device=device)
        img = _lib.Image(_ptr(out_color), _ptr(out_depth), _ptr(out_norm), _ptr(out_alpha), fix.ptr("final_T"),
                         fix.ptr("n_contrib"), None, None, None, None, None)

        def alloc_bin(cap, lists_ar):
            c1 = max(cap, 1)
            sort_bytes = lib.texgs_sort_temp_bytes(cap, tiles)
            ar = _Arena(device)
            ar.add("keys_unsorted", (c1,), torch.int64)
            ar.add("keys_sorted", (c1,), torch.int64)
            ar.add("point_list", (c1,), i32)
            ar.add("sort_temp", (sort_bytes,), u8)
            if lists_ar is None:
                add_lists(ar)
            if handoff:        # K6 -> K7 hand-off of the per-block survivor lists: four blocks per tile, each at most the tile's list length
                ar.add("survivors", (4 * c1, 2), i32)
                ar.add("surv_qmask", (4 * c1,), torch.int16)
            item_pages = item_pools = 0
            if handoff and want_items:      # ... and of K6's item stream
                item_pages, item_pools = _item_layout(hint_key, c1, tiles)
                ar.add("item_ctl", (_lib.ITEM_CTL_WORDS,), i32)
                ar.add("item_tail", (4 * tiles, 2), i32)
                ar.add("item_link", (item_pages,), i32)
                ar.add("item_pages", (item_pages, 3, _lib.ITEM_PAGE), i32)
            ar.commit()
            la = lists_ar if lists_ar is not None else ar
            b = _lib.Binning(0, ar.ptr("keys_unsorted"), ar.ptr("keys_sorted"), ar.ptr("point_list"), la.ptr("ranges"),
                             la.ptr("tile_order"), ar.ptr("sort_temp"), sort_bytes)
            img.tex_bin_count, img.surv_count = la.ptr("tex_bin_count"), la.ptr("surv_count")
            img.tex_bin_resv = la.ptr("tex_bin_resv")
            img.survivors, img.surv_qmask = ar.ptr("survivors"), ar.ptr("surv_qmask")
            img.item_pages, img.item_link = ar.ptr("item_pages"), ar.ptr("item_link")
            img.item_tail, img.item_ctl = ar.ptr("item_tail"), ar.ptr("item_ctl")
            img.item_page_cap, img.item_sub_pools = item_pages, item_pools
            return b, ar
        hint_key = (device.index, N, H, W)
        istat = None
        if handoff and want_items:
            istat = _ITEM_STAT.get(gkey)
            if istat is None:
                istat = _ITEM_STAT[gkey] = _ItemStat()
            istat.poll()
        cap = _CAPACITY_HINT.get(hint_key, max(4 * N, 1024))
        binning = bin_ar = None
        if not candidate:
            binning, bin_ar = alloc_bin(cap, fix)
        _lib.check(lib.texgs_preprocess_forward(C.byref(frame), C.byref(inputs), C.byref(geom), stream), "texgs_preprocess_forward")
        # the one device->host read of a forward (D and K1's geometry fingerprint), in two steps: the copy into pinned memory is issued
        # here (with K2 behind it unless earlier lists are expected to be shared), the host waits for it in finish() -- immediately when
        # this is an ordinary forward, or several views later when it was PREFETCHED (prefetch_forward: K1 of a later view issued ahead
        # of this stream's pending backward, so that its forward never waits for the stream to drain)
        pin = _pin_take(int(lib.texgs_num_rendered_words(N)))
        _lib.check(lib.texgs_num_rendered_begin(C.byref(geom), N, pin.data_ptr(), pin.numel(), 0, stream), "texgs_num_rendered_begin")
        d_ev = torch.cuda.Event()
        d_ev.record(cur_stream)
        if not candidate:           # K2 behind the event: the host waits for K1 + the copy only, the device goes on sorting
            _lib.check(lib.texgs_depth_sort_scan(C.byref(geom), N, stream), "texgs_depth_sort_scan")

    def finish():
        nonlocal binning, bin_ar, cap
        with torch.cuda.device(device):
            d_ev.synchronize()
            d_host, fp_host = C.c_uint32(0), C.c_uint64(0)
            _lib.check(lib.texgs_num_rendered_reduce(pin.data_ptr(), N, C.byref(d_host), C.byref(fp_host)), "texgs_num_rendered_reduce")
            _pin_give(pin)
            D, fp = int(d_host.value), int(fp_host.value)
            shared = None
            if candidate and fp == entry.fingerprint and D == entry.D:
                # same geometry as the forward that built `entry`: its lists are this forward's lists; K6 alone, no hand-off work
                _GEOM_STATS["hits"] += 1
                shared = entry
                binning = _lib.Binning(D, *entry.bin)
                _lib.check(lib.texgs_render_forward(C.byref(frame), C.byref(inputs), C.byref(geom), C.byref(binning), C.byref(img),
                                                    stream), "texgs_render_forward")
                if for_backward and entry.handoff and (entry.counts or not want_counts):
                    img.survivors, img.surv_qmask, img.surv_count = entry.handoff
                    img.tex_bin_count, img.tex_bin_resv = entry.counts if want_counts else (None, None)
                    if want_items and entry.items:      # T and alpha_raw of every pair are functions of the shared geometry too
                        (img.item_pages, img.item_link, img.item_tail, img.item_ctl, img.item_page_cap, img.item_sub_pools) = entry.items
                cap = entry.cap
            else:
                _GEOM_STATS["misses"] += 1
                if candidate:           # expected to share, cannot: build the lists after all (K2 was not started before the sync)
                    binning, bin_ar = alloc_bin(max(cap, int(D * 1.25) + 1024), None)
                    cap = max(cap, int(D * 1.25) + 1024)
                    _lib.check(lib.texgs_depth_sort_scan(C.byref(geom), N, stream), "texgs_depth_sort_scan")
                elif D > cap:           # rare: grow
                    cap = int(D * 1.25) + 1024
                    binning, bin_ar = alloc_bin(cap, fix)
                binning.num_rendered = D
                _lib.check(lib.texgs_bin_sort_render_forward(C.byref(frame), C.byref(inputs), C.byref(geom), C.byref(binning),
                                                             C.byref(img), stream), "texgs_bin_sort_render_forward")
                _CAPACITY_HINT[hint_key] = max(_CAPACITY_HINT.get(hint_key, 0), int(D * 1.25) + 1024)
                if istat is not None and img.item_ctl and not ITEM_PAGES_FIXED:
                    istat.watch(bin_ar.view("item_ctl"), hint_key, int(img.item_sub_pools), device)
                if GEOM_CACHE:
                    e = _GeomEntry()
                    e.ints, e.cam_key, e.fingerprint, e.D, e.cap = gints, cam_key, fp, D, cap
                    e.bin = (binning.keys_unsorted, binning.keys_sorted, binning.point_list, binning.ranges, binning.tile_order,
                             binning.sort_temp, binning.sort_temp_bytes)
                    e.arenas = (fix, bin_ar)
                    e.handoff = (img.survivors, img.surv_qmask, img.surv_count) if handoff else None
                    e.counts = (img.tex_bin_count, img.tex_bin_resv) if (handoff and img.tex_bin_count) else None
                    e.items = (img.item_pages, img.item_link, img.item_tail, img.item_ctl, int(img.item_page_cap),
                               int(img.item_sub_pools)) if (handoff and img.item_pages) else None
                    _GEOM[gkey] = e
        s = _State()
        s.frame, s.inputs, s.geom, s.bin, s.img = frame, inputs, geom, binning, img
        s.N, s.K, s.R, s.H, s.W, s.D, s.cap, s.tiles = N, K, R, H, W, D, cap, tiles
        s.want_counts, s.lazy, s.backward_ran, s.shared_geometry = want_counts, bool(lazy and for_backward), False, shared is not None
        s.want_items = want_items
        with _PRED_LOCK:
            _SERIAL[0] += 1
            s.serial = _SERIAL[0]
        arenas = (fix,) + ((bin_ar,) if bin_ar is not None else ()) + (tuple(shared.arenas) if shared is not None else ())
        # (NOT the output tensors: autograd hangs its node on them, the node holds this state -- a cycle that kept every dropped
        #  graph's buffers alive until the garbage collector ran)
        s.tensors = _Tensors(arenas, keep=keep, radii=radii)
        if img.survivors is None:           # no hand-off in this state (forward-only call, lazy mode, or a shared entry without one)
            s.tensors["survivors"] = None
            s.tensors["surv_qmask"] = None
            s.tensors["surv_count"] = None
        if img.tex_bin_count is None:
            s.tensors["tex_bin_count"] = None
        s.tensors["for_backward"] = bool(for_backward)
        return (out_color, out_depth, out_norm, out_alpha, radii), s

    if _begin_only:
        return finish
    return finish()


def _late_handoff(s: _State):
    """The forward left no survivor lists (LAZY_HANDOFF, or lists shared from a forward that had none): produce them now with one
    more K6 on the state's lists -- outputs into scratch (bit-identical to the forward's, which stay untouched)."""
    lib = _lib.load()
    device = s.tensors["keep"][0].device
    stream = torch.cuda.current_stream(device).cuda_stream
    i32, f32 = torch.int32, torch.float32
    ar = _Arena(device)
    ar.add("survivors", (4 * max(s.cap, 1), 2), i32)
    ar.add("surv_qmask", (4 * max(s.cap, 1),), torch.int16)
    ar.add("surv_count", (4 * s.tiles,), i32)
    if s.want_counts:
        ar.add("tex_bin_count", (2 * int(lib.texgs_tex_bin_count(s.R)),), i32)
        ar.add("tex_bin_resv", (4 * s.tiles, _lib.RESV_WORDS), i32)
    item_pages = item_pools = 0
    if s.want_items:
        item_pages, item_pools = _item_layout((device.index, s.N, s.H, s.W), max(s.cap, 1), s.tiles)
        ar.add("item_ctl", (_lib.ITEM_CTL_WORDS,), i32)
        ar.add("item_tail", (4 * s.tiles, 2), i32)
        ar.add("item_link", (item_pages,), i32)
        ar.add("item_pages", (item_pages, 3, _lib.ITEM_PAGE), i32)
    ar.add("scratch_out", (8, s.H, s.W), f32)
    ar.add("final_T", (s.H, s.W), f32)
    ar.add("n_contrib", (s.H, s.W), i32)
    ar.commit()
    so = ar.ptr("scratch_out")
    hw = 4 * s.H * s.W
    img = _lib.Image(so, so + 3 * hw, so + 4 * hw, so + 7 * hw, ar.ptr("final_T"), ar.ptr("n_contrib"), ar.ptr("tex_bin_count"),
                     ar.ptr("survivors"), ar.ptr("surv_qmask"), ar.ptr("surv_count"), ar.ptr("tex_bin_resv"),
                     ar.ptr("item_pages"), ar.ptr("item_link"), ar.ptr("item_tail"), ar.ptr("item_ctl"), item_pages, item_pools)
    _lib.check(lib.texgs_render_forward(C.byref(s.frame), C.byref(s.inputs), C.byref(s.geom), C.byref(s.bin), C.byref(img), stream),
               "texgs_render_forward (late hand-off)")
    s.img.survivors, s.img.surv_qmask, s.img.surv_count = img.survivors, img.surv_qmask, img.surv_count
    s.img.tex_bin_count, s.img.tex_bin_resv = img.tex_bin_count, img.tex_bin_resv
    s.img.item_pages, s.img.item_link, s.img.item_tail, s.img.item_ctl = img.item_pages, img.item_link, img.item_tail, img.item_ctl
    s.img.item_page_cap, s.img.item_sub_pools = item_pages, item_pools
    s.tensors._arenas = tuple(s.tensors._arenas) + (ar,)
    for n in ("survivors", "surv_qmask", "surv_count", "tex_bin_count", "tex_bin_resv", "item_pages", "item_link", "item_tail", "item_ctl"):
        s.tensors.pop(n, None)
    if not s.want_counts:
        s.tensors["tex_bin_count"] = None
    _GEOM_STATS["late_handoffs"] += 1


_ITEMSIZE = {torch.float32: 4, torch.int32: 4, torch.int64: 8, torch.int16: 2, torch.uint8: 1, torch.float64: 8}


class _Arena:
    """One device allocation carved into named, 256-byte-aligned pieces."""

    def __init__(self, device):
        self.device, self.specs, self.size, self.buf = device, {}, 0, None

    def add(self, name, shape, dtype):
        off = (self.size + 255) & ~255
        self.specs[name] = (off, tuple(shape), dtype)
        self.size = off + math.prod(shape) * _ITEMSIZE[dtype]

    def commit(self):
        self.buf = torch.empty(max(self.size, 1), dtype=torch.uint8, device=self.device)

    def ptr(self, name):
        return self.buf.data_ptr() + self.specs[name][0] if name in self.specs else None

    def view(self, name):
        off, shape, dtype = self.specs[name]
        n = math.prod(shape) * _ITEMSIZE[dtype]
        return self.buf[off:off + n].view(dtype).view(shape)


class _Tensors(dict):
    """What a forward leaves behind, by name: real tensors (inputs kept alive, outputs) plus views into the arenas, made on
    first access.  A name that was not allocated (e.g. the survivor lists of a forward-only call) reads as None."""

    def __init__(self, arenas, **real):
        super().__init__(**real)
        self._arenas = arenas

    def __missing__(self, name):
        for ar in self._arenas:
            if name in ar.specs:
                self[name] = v = ar.view(name)
                return v
        self[name] = None
        return None

    def get(self, name, default=None):
        v = self[name]
        return default if v is None else v


def backward_raw(s: _State, dL_dcolor, dL_ddepth, dL_dnorm, dL_dalpha, sinks=None, before_accumulate=None,
                 want=_lib.WANT_ALL):
    """Run K7+K8.  Returns grads (means3D, means2D, shs, opacities, scales, rotations, uvs, texture).

    `want` (TEXGS_WANT_* bits): WANT_TEXTURE = dL/dtexture, WANT_GAUSSIANS = every per-Gaussian gradient.  What is not wanted
    is not computed (K7 is compiled in four flavours, K8 is skipped without WANT_GAUSSIANS) and comes back as None.

    `before_accumulate` (optional callable): invoked between K7 + bin reduce and K8 -- the point where a multi-view
    pipeline makes this stream wait for the previous view's K8 (texgs.multiview.ViewPipeline).

    `sinks` (optional): dict name -> existing float32 gradient buffer of the input's shape.  K8 ADDS into the sink of
    every per-Gaussian output that has one (TexGSGrads.accumulate bit mask: fused multi-view accumulation, or a leaf's
    existing .grad) and writes the others into one fresh allocation; a texture sink is used the same way (dL_dtexture is
    always accumulated into).  Outputs written into a sink come back as None."""
    lib = _lib.load()
    if not s.tensors["for_backward"]:
        raise RuntimeError("this forward ran with for_backward=False (no survivor lists were kept): its backward cannot run")
    means3D = s.tensors["keep"][0]
    device = means3D.device
    f32 = dict(dtype=torch.float32, device=device)
    N, K, R = s.N, s.K, s.R
    stream = torch.cuda.current_stream(device).cuda_stream
    textured = s.tensors["keep"][7] is not None
    has_cov = s.tensors["keep"][9] is not None
    want &= _lib.WANT_ALL if textured else _lib.WANT_GAUSSIANS
    want_g, want_t = bool(want & _lib.WANT_GAUSSIANS), bool(want & _lib.WANT_TEXTURE)
    if not want:
        return (None,) * 8

    def g(t, shape):
        if t is None:
            return None
        t = t.to(torch.float32).contiguous()
        assert tuple(t.shape) == shape, (tuple(t.shape), shape)
        return t
    H, W = s.H, s.W
    dc, dd, dn, da = g(dL_dcolor, (3, H, W)), g(dL_ddepth, (1, H, W)), g(dL_dnorm, (3, H, W)), g(dL_dalpha, (1, H, W))
    with torch.cuda.device(device):
        if s.img.survivors is None:
            _late_handoff(s)
        s.backward_ran = True
        if s.lazy:
            with _PRED_LOCK:
                _LAST_DIFFERENTIATED[device.index] = max(_LAST_DIFFERENTIATED.get(device.index, -1), s.serial)
        skey = (device.index, int(stream))
        sc = _SCRATCH.pop(skey, None) or _StreamScratch()     # re-cached only after a successful call (an exception drops it)
        acc = None
        if want_g:
            if sc.acc is None or sc.acc.shape[0] < max(N, 1):
                sc.acc = torch.zeros(max(N, 1), _lib.ACC_FLOATS, **f32)
            acc = sc.acc
        sinks = sinks or {}
        has_coff = s.tensors["keep"][8] is not None
        shapes = {}
        if want_g:
            shapes = dict(means3D=(N, 3), means2D=(N, 3), opacities=(N, 1))
            if has_cov:
                shapes["cov3D"] = (N, 6)
            else:
                shapes.update(scales=(N, 3), rotations=(N, 4))
            if textured:
                shapes["uvs"] = (N, 3)
            if K > 0:
                shapes["shs"] = (N, K, 3)
            if has_coff:
                shapes["color_offset"] = (N, 3)
        # one allocation for every per-Gaussian output without a sink
        fresh = [n for n in shapes if n not in sinks]
        flat = torch.empty(sum(math.prod(shapes[n]) for n in fresh), **f32) if fresh else None
        outs, off, mask = {}, 0, 0
        for n, shp in shapes.items():
            if n in sinks:
                t = sinks[n]
                assert t.dtype == torch.float32 and t.is_contiguous() and t.numel() == math.prod(shp), n
                outs[n] = t
                mask |= _ACC_BITS[n]
            else:
                cnt = math.prod(shp)
                outs[n] = flat[off:off + cnt].view(shp)
                off += cnt
        tex_sink = sinks.get("texture") if want_t else None
        d_tex = None
        if want_t:
            d_tex = tex_sink if tex_sink is not None else torch.zeros(6, R, R, 3, **f32)
        bins = None
        if want_t and USE_TEX_BINS and s.tensors.get("tex_bin_count") is not None:
            bins = sc.bins if (sc.bins is not None and sc.bins.R == R) else _TexBins(lib, device, R)
            sc.bins = None
            bins.before_call(s.D, s.tensors.get("tex_bin_count"))
        grads = _lib.Grads(_ptr(dc), _ptr(dd), _ptr(dn), _ptr(da), _ptr(acc), _ptr(outs.get("means3D")), _ptr(outs.get("means2D")),
                           _ptr(outs.get("shs")), _ptr(outs.get("opacities")), _ptr(outs.get("scales")), _ptr(outs.get("rotations")),
                           _ptr(outs.get("uvs")), _ptr(d_tex), _ptr(outs.get("color_offset")), _ptr(outs.get("cov3D")), want,
                           _ptr(bins.rec) if bins else None, _ptr(bins.cursor) if bins else None,
                           _ptr(bins.base) if bins else None, bins.cap if bins else 0, mask)
        if before_accumulate is None:
            _lib.check(lib.texgs_backward(C.byref(s.frame), C.byref(s.inputs), C.byref(s.geom), C.byref(s.bin),
                                          C.byref(s.img), C.byref(grads), stream), "texgs_backward")
        else:
            _lib.check(lib.texgs_backward_render(C.byref(s.frame), C.byref(s.inputs), C.byref(s.geom), C.byref(s.bin),
                                                 C.byref(s.img), C.byref(grads), stream), "texgs_backward_render")
            before_accumulate()
            _lib.check(lib.texgs_backward_preprocess(C.byref(s.frame), C.byref(s.inputs), C.byref(s.geom),
                                                     C.byref(grads), stream), "texgs_backward_preprocess")
    if bins is not None:
        bins.after_call()
        sc.bins = bins
    _SCRATCH[skey] = sc
    res = {n: (None if n in sinks else outs[n]) for n in shapes}
    s.tensors["d_color_offset"] = res.get("color_offset")
    s.tensors["d_cov3D"] = res.get("cov3D")
    return (res.get("means3D"), res.get("means2D"), res.get("shs"), res.get("opacities"), res.get("scales"), res.get("rotations"),
            res.get("uvs"), None if tex_sink is not None else d_tex)


def _leaf_grad_sink(t):
    """A leaf's existing, contiguous float32 .grad of the right shape (the kernels then add into it directly: no zero-filled
    temporary, no AccumulateGrad pass), else None."""
    if t is None or not (t.is_leaf and t.requires_grad):
        return None
    if getattr(t, "_backward_hooks", None) or getattr(t, "_post_accumulate_grad_hooks", None):
        return None             # a hook must see the gradient: let autograd deliver it
    gr = t.grad
    if gr is None or gr.dtype != torch.float32 or not gr.is_contiguous() or gr.shape != t.shape or gr.device != t.device:
        return None
    return gr


# inputs of _RasterizeGaussians.apply by position; the per-Gaussian ones decide TEXGS_WANT_GAUSSIANS
_ARG = dict(means3D=0, means2D=1, shs=2, opacities=3, scales=4, rotations=5, uvs=6, gradient_uvs=7, texture=8, st=9,
            color_offset=10, grad_sink=11, cov3D_precomp=12)
_GAUSSIAN_ARGS = ("means3D", "means2D", "shs", "opacities", "scales", "rotations", "uvs", "color_offset", "cov3D_precomp")


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, shs, opacities, scales, rotations, uvs, gradient_uvs, texture, st, color_offset=None,
                grad_sink=None, cov3D_precomp=None):
        ctx.grad_sink = grad_sink
        ctx.set_materialize_grads(False)     # outputs without an upstream gradient arrive as None, not as zero-filled tensors
        # the inputs by name: the backward looks at them again (bucket slices / leaf .grad buffers to accumulate into)
        ctx.named = dict(means3D=means3D, means2D=means2D, shs=shs, opacities=opacities, scales=scales,
                         rotations=rotations, uvs=uvs, texture=texture, color_offset=color_offset, cov3D=cov3D_precomp)
        # what the backward will be asked for decides what the forward prepares and what the backward computes: the texture
        # gradient (K6's per-bin footprint counts, K7's records, the bin reduce) only if the texture wants one; the per-Gaussian
        # stages (K7's recurrence + moment sums, K8) only if some per-Gaussian input does (texgs.h TEXGS_WANT_*)
        nig = tuple(ctx.needs_input_grad) + (False,) * len(_ARG)      # (apply() may be called without the trailing optional inputs)
        want = (_lib.WANT_TEXTURE if (texture is not None and nig[_ARG["texture"]]) else 0) \
            | (_lib.WANT_GAUSSIANS if any(nig[_ARG[n]] for n in _GAUSSIAN_ARGS) else 0)
        ctx.want = want
        ctx.nargs = len(ctx.needs_input_grad)
        # (autograd does not record inside Function.forward: the inputs go in as they are, no detach() copies of the tensor objects)
        outs, state = forward_raw(st, means3D, shs, opacities, scales, rotations, uvs, gradient_uvs, texture, color_offset,
                                  for_backward=bool(want), cov3D_precomp=cov3D_precomp,
                                  count_bins=bool(want & _lib.WANT_TEXTURE), lazy=True,
                                  want_items=bool(want & _lib.WANT_GAUSSIANS))
        color, depth, norm, alpha, radii = outs
        ctx.state = state
        # the backward re-reads the inputs through raw pointers (K8 recomputes geometry, K7 re-fetches texels): remember
        # their autograd version counters so an in-place update in between (optimizer.step, change_texture) is an error,
        # not a silently wrong gradient
        ctx.versions = [(t, t._version) for t in state.tensors["keep"] if t is not None]
        ctx.op_shape = opacities.shape
        ctx.mark_non_differentiable(radii)
        return color, depth, norm, alpha, radii

    @staticmethod
    def backward(ctx, dL_dcolor, dL_ddepth, dL_dnorm, dL_dalpha, _dradii):
        s = ctx.state
        if s is None:
            raise RuntimeError("the rasterizer's backward ran a second time: its per-call state was released after the "
                               "first backward (retain_graph=True is not supported); run the forward again")
        for t, v in ctx.versions:
            if t._version != v:
                raise RuntimeError("an input of the rasterizer was modified in place between its forward and backward "
                                   "(the backward re-reads inputs through saved pointers); clone it before modifying")
        # where the gradients go: (1) the slices of a texgs.multiview.GradBucket (fused multi-view accumulation) when one is
        # attached; (2) opt-in (TEXGS_DIRECT_LEAF_GRADS=1), for an input that is a hook-free LEAF whose .grad already exists,
        # that .grad itself; (3) fresh tensors handed back to autograd for everything else (the default).
        sinks, bucket = {}, False
        if ctx.grad_sink is not None:
            sinks = {k: g for k, g in ((k, ctx.grad_sink.sink_for(v)) for k, v in ctx.named.items() if v is not None)
                     if g is not None}
            bucket = bool(sinks)
        if not bucket and DIRECT_LEAF_GRADS:
            sinks = {k: g for k, g in ((k, _leaf_grad_sink(v)) for k, v in ctx.named.items()) if g is not None}
        d_means3D, d_means2D, d_shs, d_op, d_scales, d_rot, d_uvs, d_tex = backward_raw(
            s, dL_dcolor, dL_ddepth, dL_dnorm, dL_dalpha, sinks=sinks or None,
            before_accumulate=getattr(ctx.grad_sink, "before_accumulate", None) if bucket else None, want=ctx.want)
        d_coff = s.tensors.get("d_color_offset")
        d_cov = s.tensors.get("d_cov3D")
        ctx.state = None
        ctx.versions = None
        ctx.named = None
        if d_op is not None:
            d_op = d_op.reshape(ctx.op_shape)
        return (d_means3D, d_means2D, d_shs, d_op, d_scales, d_rot, d_uvs, None, d_tex, None, d_coff, None, d_cov)[:ctx.nargs]


EXTRA_ATTRS_MAX = 32


def blend_extra_attrs(st, means3D, means2D, opacities, scales, rotations, extra_attrs, grad_sink=None, cov3D_precomp=None):
    """extra[C,H,W] = sum_i w_i * extra_attrs[i, c]: the lineage operator's `extra_attrs` kwarg / sixth return value
    (render/uv_tex_render.py:66,76, render/render.py:84 -- the reference itself always passes None).  Blended like depth and normals:
    no background term, no clamp.  NOT a fused path: ceil(C / 3) more passes of the UNTEXTURED operator (K1 + K6 on the lists of the
    main pass -- same camera, same Gaussians: the shared-geometry path -- and its own backward), three channels at a time as the
    pass's colour.  The colour path clamps at zero, so the attributes go in shifted by m = min(extra_attrs) - 1 (> 0 everywhere: the
    clamp never acts, its gradient mask never bites) and m * alpha is added back.  Differentiable w.r.t. extra_attrs and every
    geometry input through plain autograd."""
    N = means3D.shape[0]
    if not isinstance(extra_attrs, torch.Tensor) or extra_attrs.dim() != 2 or extra_attrs.shape[0] != N:
        raise ValueError(f"extra_attrs must be a [N, C] tensor with N = {N} rows")
    C_ = int(extra_attrs.shape[1])
    if C_ > EXTRA_ATTRS_MAX:
        raise ValueError(f"extra_attrs has {C_} channels; at most {EXTRA_ATTRS_MAX} are blended")
    H, W = int(st.image_height), int(st.image_width)
    ea = extra_attrs.to(torch.float32)
    if C_ == 0 or N == 0:
        return torch.zeros(C_, H, W, dtype=torch.float32, device=means3D.device) + 0.0 * ea.sum()
    m = ea.detach().amin() - 1.0                       # (a device scalar: no sync)
    st0 = st._replace(bg=torch.zeros_like(st.bg), sh_degree=0)
    if means2D is None:
        means2D = torch.zeros_like(means3D)
    outs = []
    for c0 in range(0, C_, 3):
        grp = ea[:, c0:c0 + 3]
        if grp.shape[1] < 3:
            grp = torch.cat([grp, (m + 1.0).expand(N, 3 - grp.shape[1])], dim=1)
        color, _, _, alpha, _ = _RasterizeGaussians.apply(means3D, means2D, None, opacities, scales, rotations, None, None, None, st0,
                                                          ((grp - m) - 0.5).contiguous(), grad_sink, cov3D_precomp)
        outs.append(color + m * alpha)
    return torch.cat(outs, dim=0)[:C_]


class GaussianRasterizer(nn.Module):
    """Same call surface as the reference's rasterizer (render/uv_tex_render.py:40,56-66)."""

    def __init__(self, raster_settings: GaussianRasterizationSettings, grad_sink=None):
        super().__init__()
        self.raster_settings = raster_settings
        # optional texgs.multiview.GradBucket: gradients of inputs that are its registered leaves are accumulated
        # into the bucket by the kernels themselves (no autograd AccumulateGrad pass); not part of the reference API
        self.grad_sink = grad_sink

    def markVisible(self, positions):
        """Frustum test only (lineage API; unused by the reference)."""
        lib = _lib.load()
        st = self.raster_settings
        positions = _f32c(positions, "positions", positions.device)
        N = positions.shape[0]
        keep = []
        frame = _make_frame(st, N, 0, 1, positions.device, keep)
        vis = torch.zeros(N, dtype=torch.uint8, device=positions.device)
        stream = torch.cuda.current_stream(positions.device).cuda_stream
        with torch.cuda.device(positions.device):
            _lib.check(lib.texgs_mark_visible(C.byref(frame), positions.data_ptr(), vis.data_ptr(), stream),
                       "texgs_mark_visible")
        return vis.bool()

    def prefetch(self, means3D, means2D, opacities, shs=None, scales=None, rotations=None, uvs=None,
                 gradient_uvs=None, texture=None, extra_attrs=None):
        """Begin this forward now (K1 + the asynchronous instance-count readback + K2, on the current stream); the same call through
        forward() later -- same tensors, unchanged in between, same settings tensors, same stream -- finishes it without waiting for
        whatever the stream was given in the meantime (prefetch_forward).  Not part of the reference API; a no-op for its results."""
        grad = torch.is_grad_enabled()
        rg = lambda t: bool(grad and t is not None and t.requires_grad)
        want_tex = rg(texture)
        want_g = any(rg(t) for t in (means3D, means2D, shs, opacities, scales, rotations, uvs))
        prefetch_forward(self.raster_settings, means3D, shs, opacities, scales, rotations, uvs, gradient_uvs, texture, None,
                         for_backward=want_tex or want_g, cov3D_precomp=None, count_bins=want_tex, lazy=True, want_items=want_g)

    def forward(self, means3D, means2D, opacities, shs=None, scales=None, rotations=None, uvs=None,
                gradient_uvs=None, texture=None, extra_attrs=None):
        st = self.raster_settings
        if scales is None or rotations is None:
            raise ValueError("scales and rotations are required (the textured operator has no cov3D_precomp path)")
        if uvs is None or gradient_uvs is None or texture is None:
            raise ValueError("uvs, gradient_uvs and texture are required")
        if means2D is None:
            means2D = torch.zeros_like(means3D)
        color, depth, norm, alpha, radii = _RasterizeGaussians.apply(
            means3D, means2D, shs, opacities, scales, rotations, uvs, gradient_uvs, texture, st, None, self.grad_sink)
        extra = None
        if extra_attrs is not None:         # (always None in the reference, render/uv_tex_render.py:66; see blend_extra_attrs)
            extra = blend_extra_attrs(st, means3D, means2D, opacities, scales, rotations, extra_attrs, self.grad_sink)
        return color, depth, norm, alpha, radii, extra
