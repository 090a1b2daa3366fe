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
KERNEL_SOURCES = ["render.hip", "render_bwd_body.h", "preprocess.hip", "binning.hip", "common.h", "wave_ops.h"]


def source_hash():
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        h.update(open(os.path.join(ROOT, "texture-gs_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    root, out = sys.argv[1], sys.argv[2]
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(root, "pmc*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
                n = r["Kernel_Name"]
                for k, short in (("k_render_fwd", "render_fwd"), ("k_render_bwd", "render_bwd"), ("k_preprocess_fwd", "preprocess_fwd"),
                                 ("k_preprocess_bwd", "preprocess_bwd"), ("k_texgrad_reduce", "texgrad_reduce"), ("k_duplicate", "duplicate"),
                                 ("k_ranges", "ranges")):
                    if k in n:
                        vals[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
    res = {}
    for k, c in vals.items():
        f = sum(c["FETCH_SIZE"]) / max(len(c["FETCH_SIZE"]), 1)
        w = sum(c["WRITE_SIZE"]) / max(len(c["WRITE_SIZE"]), 1)
        res[k] = {"fetch_bytes": int(2 * f * 1024), "write_bytes": int(w * 1024), "traffic_bytes": int((2 * f + w) * 1024)}
    res["_kernel_source_hash"] = source_hash()
    res["_calibration"] = ("FETCH_SIZE x2 = 128-B line traffic (streaming reads 0.5000 of known bytes; 12-B gathers 64 B counted per miss); "
                           "WRITE_SIZE x1 for coalesced stores and 64-B atomic runs, 32 B per scattered 4-B atomic "
                           "(profiles/r04_traffic_calibration.json, scripts/ubench/traffic_cal.hip)")
    res["_note"] = ("per launch, averaged over the profiled launches of `bench.py --steps 1 --warmup 2 --streams 1` (scripts/prof.sh); "
                    "traffic past the L2 (Infinity-Cache hits are counted, not excluded)")
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res)[:800])
