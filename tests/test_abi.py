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
