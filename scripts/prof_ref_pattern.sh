#!/bin/bash
# rocprofv3 kernel trace of the reference call pattern (scripts/ref_pattern.py): every kernel by name -> gpurun_out/prof_ref/
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_ref
rm -rf $OUT; mkdir -p $OUT
for MODE in plain iteration; do
  rocprofv3 --kernel-trace --stats -d $OUT/$MODE -o t --output-format csv -- python $R/scripts/ref_pattern.py 32 $MODE > $OUT/$MODE.json 2> $OUT/$MODE.err
  tail -1 $OUT/$MODE.json
  python - <<PY
import csv, glob, collections
f = glob.glob("$OUT/$MODE/**/*kernel_stats.csv", recursive=True)
if f:
    rows = list(csv.DictReader(open(f[0])))
    lib = ("k_render", "k_preprocess", "k_texgrad", "k_bin_offsets", "k_depth", "k_radix", "k_duplicate", "k_ranges", "k_tile_order")
    # 6 warm-up + 32 timed + 32 issue-timed + 8 profiled views
    views = 78.0
    tot = collections.OrderedDict()
    for r in rows:
        n = r["Name"]
        cat = "library" if any(k in n for k in lib) else ("copy/fill" if ("copyBuffer" in n or "fillBuffer" in n) else "torch: " + n[:60])
        t = tot.setdefault(cat, [0.0, 0]); t[0] += float(r["TotalDurationNs"]); t[1] += int(r["Calls"])
    print("$MODE: kernel time per view by category (us), calls per view")
    for c, (ns, calls) in sorted(tot.items(), key=lambda kv: -kv[1][0])[:14]:
        print("  %-72s %9.1f %7.2f" % (c, ns / 1e3 / views, calls / views))
PY
done
