"""Per-kernel averages of the binning front (K1-K3) from a rocprofv3 --stats directory: python k2_stats.py <dir> <tag>."""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
print(sys.argv[2])
tot = 0.0
for r in csv.DictReader(open(f)):
    n = r["Name"]
    m = re.search(r"(k_[a-z_0-9]+(<[^>]*>)?)", n)
    if m and any(k in n for k in ("k_depth", "k_duplicate", "k_preprocess_fwd", "k_tile", "k_ranges", "k_sort")):
        us = float(r["AverageNs"]) / 1e3
        tot += us * int(r["Calls"]) / 49.0
        print("  %-50s calls %5s avg_us %8.1f" % (m.group(1)[:50], r["Calls"], us))
