"""Build libtexgs.so (hipcc, gfx950 only) in-tree.  Usage: python texture-gs_amd/build.py [--force] [--verbose]

Objects are rebuilt only when their sources are newer.  preprocess.hip is built with -ffp-contract=off (its
fp32 operation order is part of the bit-exact key/rect/radius contract); the render kernels allow contraction
and use hardware fp32 atomics (-munsafe-fp-atomics).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, os.environ.get("TEXGS_LIB_NAME", "libtexgs.so"))     # TEXGS_LIB_NAME: experiment builds
OBJ = os.path.join(HERE, os.environ.get("TEXGS_OBJ_DIR", "build"))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"),
          "-I" + CSRC, "-Wall"]
UNITS = {
    "preprocess.hip": ["-ffp-contract=off"],
    "binning.hip": [],
    # -fno-slp-vectorize: the SLP vectoriser pairs fp32 ops into v_pk_* (packed fp32); in K6 / K7 the register-pair shuffling
    # (v_mov) that feeds them costs more issue slots than the packing saves and 9-14 VGPRs: K6 289 -> 265 us, K7 820 -> 772 us
    "render.hip": ["-munsafe-fp-atomics", "-ffp-contract=fast", "-fno-slp-vectorize"],
    "loss.hip": [],
    "selftest.hip": [],
    "uvnet.hip": [],
    "abi.hip": [],
}
HEADERS = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "wave_ops.h"), os.path.join(CSRC, "render_bwd_body.h"),
           os.path.join(CSRC, "render_bwd_stream.h"), os.path.join(ROOT, "include", "texgs.h")]


def build_id():
    """Identity of what libtexgs.so is built FROM: sha256 over csrc/*, include/texgs.h and the compile flags (16 hex digits).
    build() bakes it into the library (texgs_build_id()); texgs/_lib.py recomputes it from the tree and refuses a library that
    was built from other sources -- the prebuilt .so is what travels to the GPU box, mtimes do not."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))) + [os.path.join(ROOT, "include", "texgs.h")]
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    h.update(repr((COMMON[:4], sorted(UNITS.items()), os.environ.get("TEXGS_EXTRA_FLAGS", ""))).encode())
    return h.hexdigest()[:16]


def _newer(src_list, target):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in src_list)


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    objs = []
    bid = build_id()
    idfile = os.path.join(OBJ, "build_id.txt")
    stale_id = not os.path.exists(idfile) or open(idfile).read().strip() != bid
    for src, flags in UNITS.items():
        sp = os.path.join(CSRC, src)
        op = os.path.join(OBJ, src.replace(".hip", ".o"))
        objs.append(op)
        if src == "abi.hip":        # carries the build id: rebuilt whenever any source or flag changed
            flags = flags + ['-DTEXGS_BUILD_ID="%s"' % bid]
        if force or _newer([sp] + HEADERS + [os.path.abspath(__file__)], op) or (src == "abi.hip" and stale_id):
            cmd = [HIPCC] + COMMON + flags + os.environ.get("TEXGS_EXTRA_FLAGS", "").split() + ["-c", sp, "-o", op]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
    if force or _newer(objs, LIB):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    with open(idfile, "w") as f:
        f.write(bid + "\n")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv or True))
