"""Turn the FETCH_SIZE / WRITE_SIZE PMC passes of scripts/prof.sh into profiles/<tag>_traffic.json (bytes per launch).

Corrections as calibrated on this part with scripts/ubench/traffic_cal.hip (profiles/r04_traffic_calibration.json; the guide's
HBM section prescribes the first, the others were uncalibrated before round 4):
  FETCH_SIZE   counts one 128-B line request as 64 B: x2 -- measured 0.5000 of the known bytes on a 16-B-per-lane streaming
               read AND 64 B per miss on 12-byte (dwordx3) gathers that each touch their own 128-B line, so the x2 figure is the
               LINE traffic of the taps (128 B per missing tap, 10.7x its 12 useful bytes), not an overstatement;
  WRITE_SIZE   x1 for coalesced stores (1.0000 measured) and for 64-B runs of fp32 atomics (accumulator rows: 1.0000);
               a scattered 4-byte atomic is counted as one 32-B sector (8.0x its 4 bytes); atomics fetch nothing (FETCH ~ 0).
The file records a hash of the kernel sources it was measured on; bench.py ignores a traffic file whose hash differs from the
sources of the library it runs.
usage: make_traffic.py <prof dir> <out.json>"""
import collections
import csv
import glob
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# kernel name fragment -> bench.py kernel group (texgs.h TEXGS_K_*); k_bin_offsets is launched inside the render_bwd bracket
GROUPS = (("k_render_fwd", "render_fwd"), ("k_render_bwd", "render_bwd"), ("k_bin_offsets", "render_bwd"), ("k_preprocess_fwd", "preprocess_fwd"),
          ("k_preprocess_bwd", "preprocess_bwd"), ("k_texgrad_reduce", "texgrad_reduce"), ("k_duplicate", "duplicate"),
          ("k_ranges", "ranges"), ("k_tile_order", "ranges"), ("k_depth_", "scan"), ("k_group_prefix", "duplicate"), ("k_radix_", "sort"))


def source_hash():
    """Identity of the kernel sources a traffic file was measured on = the library's build id (texture-gs_amd/build.py build_id())."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("texgs_build_script", os.path.join(ROOT, "texture-gs_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.build_id()


if __name__ == "__main__":
    root, out = sys.argv[1], sys.argv[2]
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    views = collections.defaultdict(int)
    for f in glob.glob(os.path.join(root, "pmc*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            c = r["Counter_Name"]
            if c in ("FETCH_SIZE", "WRITE_SIZE"):
                n = r["Kernel_Name"]
                if "k_render_fwd" in n:
                    views[c] += 1                  # one K6 launch per profiled view
                for k, short in GROUPS:
                    if k in n:
                        tot[short][c] += float(r["Counter_Value"])
                        break
    res = {}
    for k, c in tot.items():
        f = c["FETCH_SIZE"] / max(views["FETCH_SIZE"], 1)
        w = c["WRITE_SIZE"] / max(views["WRITE_SIZE"], 1)
        res[k] = {"fetch_bytes": int(2 * f * 1024), "write_bytes": int(w * 1024), "traffic_bytes": int((2 * f + w) * 1024)}
    res["_views_profiled"] = dict(views)
    res["_kernel_source_hash"] = source_hash()
    res["_calibration"] = ("FETCH_SIZE x2 = 128-B line traffic (streaming reads 0.5000 of known bytes; 12-B gathers 64 B counted per miss); "
                           "WRITE_SIZE x1 for coalesced stores and 64-B atomic runs, 32 B per scattered 4-B atomic "
                           "(profiles/r04_traffic_calibration.json, scripts/ubench/traffic_cal.hip)")
    res["_note"] = ("per VIEW and kernel group (all launches of the group's kernels in the profiled views / views), "
                    "`bench.py --steps 1 --warmup 2 --streams 1` (scripts/prof.sh); traffic past the L2 (Infinity-Cache hits are counted, "
                    "not excluded)")
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res)[:800])
