"""Summarise rocprofv3 CSV output under gpurun_out/prof into a small text report (per-kernel averages)."""
import csv, glob, os, sys, collections
root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof"
def short(n):
    for k in ("k_render_fwd", "k_render_bwd", "k_preprocess_fwd", "k_preprocess_bwd", "k_duplicate", "k_ranges", "k_texgrad_reduce"):
        if k in n: return k
    if "k_radix_count" in n: return "k_radix_count"
    if "k_radix_scatter" in n: return "k_radix_scatter:" + n.split("<")[-1][:24]
    for k in ("k_depth_count", "k_depth_scatter", "k_depth_group_sort", "k_bin_offsets", "k_tile_order"):
        if k in n: return k
    return n[:60]
# stats
for f in sorted(glob.glob(os.path.join(root, "trace*", "**", "*kernel_stats.csv"), recursive=True)):
    print("== kernel stats", f)
    for r in list(csv.DictReader(open(f)))[:25]:
        print(f"{short(r['Name']):50s} calls {r['Calls']:>5s} avg_ns {float(r['AverageNs']):12.0f} total% {r['Percentage']}")
# pmc
for d in sorted(glob.glob(os.path.join(root, "pmc*"))):
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        print("== pmc", f)
        for k, cs in agg.items():
            if not k.startswith("k_"): continue
            print("  ", k, {c: round(sum(v) / len(v), 1) for c, v in cs.items()}, "n", len(next(iter(cs.values()))))
