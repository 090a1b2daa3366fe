#!/bin/bash
# rocprofv3 kernel trace of the iteration leg (scripts/bench_iteration.py): which kernels an iteration of the reference's texture
# stage is made of on this stack -- the library's, the UV map's (k_uv_taylor, k_uv_backward), torch's (losses, activations).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_iter
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/t -o t --output-format csv -- python $R/bench.py --leg iteration --steps 10 --warmup 3 > $OUT/iter.json 2> $OUT/iter.err
python - <<PY
import csv, glob
f = glob.glob("$OUT/t/**/*kernel_stats.csv", recursive=True)
if f:
    rows = list(csv.DictReader(open(f[0])))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print("iteration leg, kernels by total time (the run holds the 400-step UV fit, 2 x 13 timed iterations, 10 split, 4 profiled, 2 x 7 UV-backward timings)")
    print("%-74s %7s %10s %7s" % ("kernel", "calls", "avg us", "% time"))
    for r in rows[:28]:
        print("%-74s %7d %10.1f %7.2f" % (r["Name"][:74], int(r["Calls"]), float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
