"""gpurun_out/traffic_cal/{f,w}/*counter_collection.csv + timing.jsonl -> correction factors of FETCH_SIZE / WRITE_SIZE (KB counters)
per access pattern: counter bytes / known bytes."""
import csv, glob, json, os, sys, collections
root, out = sys.argv[1], sys.argv[2]
known = {}
for l in open(os.path.join(root, "timing.jsonl")):
    if l.startswith("{"):
        d = json.loads(l); known[d["kernel"]] = d
vals = collections.defaultdict(dict)
for f in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        for k in known:
            if k in r["Kernel_Name"] and r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
                vals[k][r["Counter_Name"]] = float(r["Counter_Value"]) * 1024.0
res = {}
for k, d in known.items():
    kb = d["known_MB"] * 1e6
    res[k] = {"known_bytes": kb, "ms": d["ms"], "GBps_of_known_bytes": d["GBps"],
              "FETCH_SIZE_bytes": vals[k].get("FETCH_SIZE"), "WRITE_SIZE_bytes": vals[k].get("WRITE_SIZE"),
              "FETCH_over_known": (vals[k]["FETCH_SIZE"] / kb) if "FETCH_SIZE" in vals[k] else None,
              "WRITE_over_known": (vals[k]["WRITE_SIZE"] / kb) if "WRITE_SIZE" in vals[k] else None}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
