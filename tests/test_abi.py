"""-m "not gpu": the C-ABI library builds (hipcc cross-compiles without a GPU), loads, exports every symbol that
include/texgs.h declares, and the ctypes mirrors in texgs/_lib.py have the C structs' sizes and field offsets.
No compute calls here."""
import ctypes
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "include", "texgs.h")


def declared_functions():
    src = open(HDR).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(texgs_[a-z_0-9]+)\s*\(", src)))


def test_header_declares_expected_entry_points():
    names = declared_functions()
    for n in ["texgs_preprocess_forward", "texgs_read_num_rendered", "texgs_bin_sort_render_forward",
              "texgs_render_forward", "texgs_backward", "texgs_mark_visible", "texgs_abi_version", "texgs_last_error"]:
        assert n in names


def test_library_exports_every_declared_symbol(lib_built):
    lib = ctypes.CDLL(lib_built)
    for n in declared_functions():
        assert hasattr(lib, n), f"{n} declared in include/texgs.h but not exported by libtexgs.so"
    from texgs import _lib
    assert sorted(_lib.EXPORTS) == declared_functions()
    assert _lib.load().texgs_abi_version() == _lib.ABI_VERSION


def test_ctypes_structs_match_c_layout(tmp_path):
    from texgs import _lib
    structs = {"TexGSFrame": _lib.Frame, "TexGSInputs": _lib.Inputs, "TexGSGeom": _lib.Geom,
               "TexGSBinning": _lib.Binning, "TexGSImage": _lib.Image, "TexGSGrads": _lib.Grads,
               "TexGSUVNet": _lib.UVNetStruct, "TexGSUVNetGrad": _lib.UVNetGradStruct}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "texgs.h"', 'int main(void) {']
    for cname, cls in structs.items():
        lines.append(f'printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    lines.append('printf("consts %d %d %d %d %d %d\\n", TEXGS_ABI_VERSION, TEXGS_TILE, TEXGS_REC_TEST_FLOATS, TEXGS_REC_SHADE_FLOATS, TEXGS_ACC_FLOATS, TEXGS_TEXBIN_RECORD_FLOATS);')
    lines.append('return 0; }')
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    out = subprocess.check_output([str(exe)]).decode().split("\n")
    got = {l.split()[0]: l.split()[1:] for l in out if l.strip()}
    for cname, cls in structs.items():
        assert int(got[cname][0]) == ctypes.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert int(got[f"{cname}.{fname}"][0]) == getattr(cls, fname).offset, (cname, fname)
    assert [int(x) for x in got["consts"]] == [_lib.ABI_VERSION, _lib.TILE, _lib.REC_TEST_FLOATS, _lib.REC_SHADE_FLOATS,
                                               _lib.ACC_FLOATS, _lib.TEXBIN_RECORD_FLOATS]


def test_header_is_plain_c(tmp_path):
    src = tmp_path / "c.c"
    src.write_text('#include "texgs.h"\nint main(void){return 0;}\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
                           "-c", str(src), "-o", str(tmp_path / "c.o")])


def test_num_rendered_reduce_is_host_only_arithmetic(lib_built):
    """texgs_num_rendered_words / texgs_num_rendered_reduce (the second half of the two-step instance-count readback, texgs.h) run on
    the host: D = sum of K1's per-workgroup partial sums, the fingerprint = the two wrapped 32-bit sums, an instance count past
    2^32 - 1 is an error, NULL arguments are errors.  (The first half needs a GPU: tests/test_contract_gpu.py.)"""
    import ctypes as C
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "texture-gs_amd"))
    from texgs import _lib
    lib = _lib.load()
    for N in (1, 255, 256, 257, 300_000):
        nblk = (N + 255) // 256
        assert lib.texgs_num_rendered_words(N) == 3 * nblk
        rng = np.random.RandomState(N)
        buf = np.concatenate([rng.randint(0, 5000, nblk), rng.randint(0, 2**32, nblk, dtype=np.int64),
                              rng.randint(0, 2**32, nblk, dtype=np.int64)]).astype(np.uint32)
        d, fp = C.c_uint32(0), C.c_uint64(0)
        rc = lib.texgs_num_rendered_reduce(buf.ctypes.data, N, C.byref(d), C.byref(fp))
        assert rc == 0
        assert d.value == int(buf[:nblk].astype(np.uint64).sum())
        fa = int(buf[nblk:2 * nblk].astype(np.uint64).sum()) & 0xFFFFFFFF
        fb = int(buf[2 * nblk:].astype(np.uint64).sum()) & 0xFFFFFFFF
        assert fp.value == (fb << 32) | fa
    assert lib.texgs_num_rendered_words(0) == 0 and lib.texgs_num_rendered_words(-3) == 0
    d = C.c_uint32(7)
    assert lib.texgs_num_rendered_reduce(buf.ctypes.data, 0, C.byref(d), None) == 0 and d.value == 0       # no Gaussians: D = 0
    big = np.full(3 * 2, 0xFFFFFFFF, dtype=np.uint32)                                                        # two workgroups of 2^32 - 1 each
    assert lib.texgs_num_rendered_reduce(big.ctypes.data, 512, C.byref(d), None) != 0
    assert b"2^32" in lib.texgs_last_error()
    assert lib.texgs_num_rendered_reduce(None, 4, C.byref(d), None) != 0
