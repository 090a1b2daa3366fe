"""Turn the FETCH_SIZE / WRITE_SIZE PMC passes of scripts/prof.sh into profiles/<tag>_traffic.json (bytes per launch).
Correction as MI355X_MICROARCH.md section HBM prescribes: counters are in KB; FETCH_SIZE counts 128-B requests at 64 B,
so it is doubled; WRITE_SIZE is taken as is (uncalibrated for partial-line / atomic traffic -- noted in the file)."""
import csv, glob, json, os, sys, collections
root, out = sys.argv[1], sys.argv[2]
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "pmc*", "*counter_collection.csv")):
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
res["_note"] = ("per launch, averaged over the profiled launches of `bench.py --steps 2 --warmup 1`; FETCH_SIZE x2 (gfx950 "
                "correction), WRITE_SIZE uncalibrated: memory-side fp32 atomics are counted as sector read-modify-writes")
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res)[:600])
