"""ctypes binding of libtexgs.so (C ABI in include/texgs.h).  Fails loudly when the HIP library is missing:
there is no CPU / eager fallback for the product path."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TEXGS_LIB") or os.path.join(os.path.dirname(_HERE), "libtexgs.so")   # TEXGS_LIB: experiment builds only

ABI_VERSION = 15
ERR_CAPACITY = 1000
TILE = 16
REC_TEST_FLOATS = 8
REC_SHADE_FLOATS = 20
TEXBIN_RECORD_FLOATS = 4        # 16-byte texture-gradient records (TexGSGrads.tex_bins)
RESV_WORDS = 192         # per 8x8 pixel block: 64 reservation entries x {bin, offset, count} (TexGSImage.tex_bin_resv)
ITEM_PAGE = 256          # items per page of the K6 -> K7 item stream (TexGSImage.item_pages: three planes of ITEM_PAGE words per page)
ITEM_MAX_POOLS = 64
ITEM_CTL_FLAG = 16 * ITEM_MAX_POOLS
ITEM_CTL_WORDS = 16 * ITEM_MAX_POOLS + 16
ACC_FLOATS = 32
WANT_TEXTURE, WANT_GAUSSIANS, WANT_ALL = 1, 2, 3

_fp = C.c_void_p  # device pointers travel as integers


class Frame(C.Structure):
    _fields_ = [("image_height", C.c_int32), ("image_width", C.c_int32), ("tanfovx", C.c_float),
                ("tanfovy", C.c_float), ("scale_modifier", C.c_float), ("sh_degree", C.c_int32),
                ("sh_coeffs", C.c_int32), ("tex_res", C.c_int32), ("num_gaussians", C.c_int32),
                ("debug", C.c_int32), ("bg", _fp), ("viewmatrix", _fp), ("projmatrix", _fp), ("campos", _fp)]


class Inputs(C.Structure):
    _fields_ = [("means3D", _fp), ("shs", _fp), ("opacities", _fp), ("scales", _fp), ("rotations", _fp),
                ("uvs", _fp), ("gradient_uvs", _fp), ("texture", _fp), ("color_offset", _fp), ("cov3D_precomp", _fp)]


class Geom(C.Structure):
    _fields_ = [("rec_test", _fp), ("rec_shade", _fp), ("depth", _fp), ("radii", _fp), ("rect", _fp), ("tiles_touched", _fp),
                ("offsets", _fp), ("scan_temp", _fp), ("scan_temp_bytes", C.c_size_t)]


class Binning(C.Structure):
    _fields_ = [("num_rendered", C.c_uint32), ("keys_unsorted", _fp), ("keys_sorted", _fp), ("point_list", _fp),
                ("ranges", _fp), ("tile_order", _fp), ("sort_temp", _fp), ("sort_temp_bytes", C.c_size_t)]


class Image(C.Structure):
    _fields_ = [("out_color", _fp), ("out_depth", _fp), ("out_norm", _fp), ("out_alpha", _fp),
                ("final_T", _fp), ("n_contrib", _fp), ("tex_bin_count", _fp),
                ("survivors", _fp), ("surv_qmask", _fp), ("surv_count", _fp), ("tex_bin_resv", _fp),
                ("item_pages", _fp), ("item_link", _fp), ("item_tail", _fp), ("item_ctl", _fp),
                ("item_page_cap", C.c_uint32), ("item_sub_pools", C.c_uint32)]


class Grads(C.Structure):
    _fields_ = [("dL_dcolor", _fp), ("dL_ddepth", _fp), ("dL_dnorm", _fp), ("dL_dalpha", _fp), ("acc", _fp),
                ("dL_dmeans3D", _fp), ("dL_dmeans2D", _fp), ("dL_dshs", _fp), ("dL_dopacities", _fp),
                ("dL_dscales", _fp), ("dL_drotations", _fp), ("dL_duvs", _fp), ("dL_dtexture", _fp),
                ("dL_dcolor_offset", _fp), ("dL_dcov3D", _fp), ("want", C.c_uint32), ("tex_bins", _fp), ("tex_bin_cursor", _fp), ("tex_bin_base", _fp),
                ("tex_rec_cap", C.c_uint32),
                ("accumulate", C.c_int32)]


class UVNetStruct(C.Structure):
    _fields_ = [(n, _fp) for n in ("W1", "b1", "W2", "b2", "emb", "W3", "b3", "W4", "b4", "W5", "b5", "xyz_offset", "xyz_scale")] \
        + [("hidden", C.c_int32)]


class UVNetGradStruct(C.Structure):
    _fields_ = [(n, _fp) for n in ("dW1", "db1", "dW2", "db2", "dW3", "db3", "dW4", "db4", "dW5", "db5")]


EXPORTS = ["texgs_abi_version", "texgs_build_id", "texgs_last_error", "texgs_scan_temp_bytes", "texgs_sort_temp_bytes",
           "texgs_preprocess_forward", "texgs_read_num_rendered", "texgs_read_num_rendered2", "texgs_num_rendered_words", "texgs_num_rendered_begin", "texgs_num_rendered_reduce", "texgs_depth_sort_scan", "texgs_bin_sort_render_forward",
           "texgs_render_forward", "texgs_forward", "texgs_backward", "texgs_backward_render", "texgs_backward_preprocess",
           "texgs_rgb_alpha_loss", "texgs_mark_visible", "texgs_profile_enable", "texgs_tex_bin_count",
           "texgs_profile_read", "texgs_profile_select", "texgs_selftest_waveops", "texgs_geom_losses", "texgs_norm_from_depth", "texgs_uv_taylor", "texgs_uv_taylor_temp_bytes", "texgs_uv_pack", "texgs_uv_taylor_packed",
           "texgs_uv_pack_bf16x3", "texgs_uv_taylor_packed_bf16x3", "texgs_uv_backward", "texgs_uv_backward_mixed", "texgs_uv_backward_temp_bytes", "texgs_uv_pack_mixed", "texgs_uv_taylor_packed_mixed"]
KERNEL_NAMES = ["preprocess_fwd", "scan", "duplicate", "sort", "ranges", "render_fwd", "render_bwd", "preprocess_bwd",
                "texgrad_reduce"]

_lib = None


def load():
    """Load libtexgs.so once.  Raises RuntimeError (never falls back) if it is absent or stale."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"libtexgs.so not found at {LIB_PATH}: build it with `python texture-gs_amd/build.py` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback for the rasterizer.")
    lib = C.CDLL(LIB_PATH)
    P = C.POINTER
    lib.texgs_abi_version.restype = C.c_int
    lib.texgs_last_error.restype = C.c_char_p
    lib.texgs_scan_temp_bytes.restype = C.c_size_t
    lib.texgs_scan_temp_bytes.argtypes = [C.c_int32]
    lib.texgs_sort_temp_bytes.restype = C.c_size_t
    lib.texgs_sort_temp_bytes.argtypes = [C.c_uint32, C.c_uint32]
    lib.texgs_tex_bin_count.restype = C.c_size_t
    lib.texgs_tex_bin_count.argtypes = [C.c_int32]
    lib.texgs_preprocess_forward.argtypes = [P(Frame), P(Inputs), P(Geom), C.c_void_p]
    lib.texgs_read_num_rendered.argtypes = [P(Geom), C.c_int32, P(C.c_uint32), C.c_void_p]
    lib.texgs_read_num_rendered2.argtypes = [P(Geom), C.c_int32, P(C.c_uint32), P(C.c_uint64), C.c_int32, C.c_void_p]
    lib.texgs_read_num_rendered2.restype = C.c_int
    lib.texgs_num_rendered_words.argtypes = [C.c_int32]
    lib.texgs_num_rendered_words.restype = C.c_size_t
    lib.texgs_num_rendered_begin.argtypes = [P(Geom), C.c_int32, C.c_void_p, C.c_size_t, C.c_int32, C.c_void_p]
    lib.texgs_num_rendered_begin.restype = C.c_int
    lib.texgs_num_rendered_reduce.argtypes = [C.c_void_p, C.c_int32, P(C.c_uint32), P(C.c_uint64)]
    lib.texgs_num_rendered_reduce.restype = C.c_int
    lib.texgs_depth_sort_scan.argtypes = [P(Geom), C.c_int32, C.c_void_p]
    lib.texgs_depth_sort_scan.restype = C.c_int
    lib.texgs_bin_sort_render_forward.argtypes = [P(Frame), P(Inputs), P(Geom), P(Binning), P(Image), C.c_void_p]
    lib.texgs_render_forward.argtypes = [P(Frame), P(Inputs), P(Geom), P(Binning), P(Image), C.c_void_p]
    lib.texgs_forward.argtypes = [P(Frame), P(Inputs), P(Geom), P(Binning), C.c_uint32, P(Image), P(C.c_uint32), C.c_void_p]
    lib.texgs_forward.restype = C.c_int
    lib.texgs_backward.argtypes = [P(Frame), P(Inputs), P(Geom), P(Binning), P(Image), P(Grads), C.c_void_p]
    lib.texgs_backward_render.argtypes = [P(Frame), P(Inputs), P(Geom), P(Binning), P(Image), P(Grads), C.c_void_p]
    lib.texgs_backward_preprocess.argtypes = [P(Frame), P(Inputs), P(Geom), P(Grads), C.c_void_p]
    lib.texgs_mark_visible.argtypes = [P(Frame), C.c_void_p, C.c_void_p, C.c_void_p]
    lib.texgs_rgb_alpha_loss.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_float,
                                         C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.texgs_rgb_alpha_loss.restype = C.c_int
    lib.texgs_geom_losses.argtypes = [C.c_void_p] * 6 + [C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_float, C.c_float,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.texgs_geom_losses.restype = C.c_int
    lib.texgs_uv_taylor_temp_bytes.restype = C.c_size_t
    lib.texgs_norm_from_depth.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_int32, C.c_int32, C.c_float,
                                          C.c_void_p, C.c_void_p, C.c_void_p]
    lib.texgs_norm_from_depth.restype = C.c_int
    lib.texgs_uv_taylor.argtypes = [P(UVNetStruct), C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.texgs_uv_taylor.restype = C.c_int
    lib.texgs_uv_pack.argtypes = [P(UVNetStruct), C.c_void_p, C.c_void_p]
    lib.texgs_uv_pack.restype = C.c_int
    lib.texgs_uv_taylor_packed.argtypes = [P(UVNetStruct), C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.texgs_uv_taylor_packed.restype = C.c_int
    lib.texgs_uv_pack_bf16x3.argtypes = [P(UVNetStruct), C.c_void_p, C.c_void_p]
    lib.texgs_uv_pack_bf16x3.restype = C.c_int
    lib.texgs_uv_taylor_packed_bf16x3.argtypes = [P(UVNetStruct), C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.texgs_uv_taylor_packed_bf16x3.restype = C.c_int
    lib.texgs_uv_pack_mixed.argtypes = [P(UVNetStruct), C.c_void_p, C.c_void_p]
    lib.texgs_uv_pack_mixed.restype = C.c_int
    lib.texgs_uv_taylor_packed_mixed.argtypes = [P(UVNetStruct), C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.texgs_uv_taylor_packed_mixed.restype = C.c_int
    lib.texgs_uv_backward_temp_bytes.argtypes = [C.c_int32]
    lib.texgs_uv_backward_temp_bytes.restype = C.c_size_t
    lib.texgs_uv_backward.argtypes = [P(UVNetStruct), C.c_void_p, C.c_void_p, C.c_int32, P(UVNetGradStruct), C.c_void_p, C.c_void_p]
    lib.texgs_uv_backward.restype = C.c_int
    lib.texgs_uv_backward_mixed.argtypes = lib.texgs_uv_backward.argtypes
    lib.texgs_uv_backward_mixed.restype = C.c_int
    lib.texgs_selftest_waveops.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.texgs_selftest_waveops.restype = C.c_int
    lib.texgs_profile_enable.argtypes = [C.c_int]
    lib.texgs_profile_enable.restype = C.c_int
    lib.texgs_profile_select.argtypes = [C.c_uint32]
    lib.texgs_profile_select.restype = C.c_int
    lib.texgs_profile_read.argtypes = [P(C.c_float), P(C.c_uint32)]
    lib.texgs_profile_read.restype = C.c_int
    for name in ("texgs_preprocess_forward", "texgs_read_num_rendered", "texgs_bin_sort_render_forward",
                 "texgs_render_forward", "texgs_forward", "texgs_backward", "texgs_backward_render",
                 "texgs_backward_preprocess", "texgs_rgb_alpha_loss", "texgs_mark_visible"):
        getattr(lib, name).restype = C.c_int
    v = lib.texgs_abi_version()
    if v != ABI_VERSION:
        # (no bypass in the shipped loader -- the struct layouts differ between versions; experiment scripts that A/B an
        #  older build must run it under that build's own host layer)
        raise RuntimeError(f"libtexgs.so ABI version {v} != expected {ABI_VERSION}; rebuild it")
    lib.texgs_build_id.restype = C.c_char_p
    global BUILD_ID
    BUILD_ID = lib.texgs_build_id().decode()
    if not os.environ.get("TEXGS_LIB"):          # the product library must be built from THIS tree's sources (experiment builds: any)
        want = tree_build_id()
        if want is not None and BUILD_ID != want:
            raise RuntimeError(f"libtexgs.so was built from other sources (build id {BUILD_ID}, this tree is {want}); "
                               "rebuild it with `python texture-gs_amd/build.py`")
    _lib = lib
    return lib


BUILD_ID = None


def tree_build_id():
    """The build id of the sources in this tree (texture-gs_amd/build.py build_id()); None if the build script is not there."""
    import importlib.util
    path = os.path.join(os.path.dirname(_HERE), "build.py")
    if not os.path.exists(path):
        return None
    spec = importlib.util.spec_from_file_location("texgs_build_script", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.build_id()


def check(rc, what):
    if rc != 0:
        msg = load().texgs_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what} failed (code {rc}): {msg}")


def profile_enable(on: bool, only=None):
    """Bracket kernel launches with HIP events.  `only`: iterable of KERNEL_NAMES to restrict to (an event pair costs a few
    microseconds of stream time, so a timed region should bracket as little as it needs)."""
    mask = 0xFFFFFFFF if only is None else sum(1 << KERNEL_NAMES.index(n) for n in only)
    check(load().texgs_profile_select(mask), "texgs_profile_select")
    check(load().texgs_profile_enable(1 if on else 0), "texgs_profile_enable")


def profile_read():
    """-> {kernel_name: (total_ms, launches)} since the last read (synchronises the recorded events)."""
    n = len(KERNEL_NAMES)
    ms = (C.c_float * n)()
    cnt = (C.c_uint32 * n)()
    check(load().texgs_profile_read(ms, cnt), "texgs_profile_read")
    return {KERNEL_NAMES[i]: (float(ms[i]), int(cnt[i])) for i in range(n)}
