#!/bin/bash
# rocprofv3 evidence for the kernels of the "next" rows (uv_taylor: MFMA; loss kernels: HBM)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_next
rm -rf $OUT; mkdir -p $OUT
python $R/scripts/bench_next_rows.py 20 > $OUT/timings.json 2> $OUT/timings.err
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace --output-format csv -- python $R/scripts/bench_next_rows.py 5 > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --kernel-trace -d $OUT/pmc1 -o pmc1 --output-format csv -- python $R/scripts/bench_next_rows.py 2 > $OUT/pmc1.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc2 -o pmc2 --output-format csv -- python $R/scripts/bench_next_rows.py 2 > $OUT/pmc2.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc3 -o pmc3 --output-format csv -- python $R/scripts/bench_next_rows.py 2 > $OUT/pmc3.log 2>&1
cat $OUT/timings.json
python - <<PY
import csv, glob, collections
f = glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True)
if f:
    for r in csv.DictReader(open(f[0])):
        n = r["Name"]
        if any(k in n for k in ("k_uv", "k_ssim", "k_geom", "k_norm_from", "k_l1", "k_loss")):
            print("%-60s calls %5s avg_ns %10s" % (n[:60], r["Calls"], r["AverageNs"]))
for d in ("pmc1", "pmc2", "pmc3"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % d, recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            for k in ("k_uv_taylor", "k_ssim", "k_geom", "k_norm_from"):
                if k in n: agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in agg.items():
            print(d, k, {c: round(sum(v) / len(v) / 1e6, 3) for c, v in cs.items()}, "(millions)")
PY
