#!/bin/bash
# rocprofv3 of the UV map's kernels alone: kernel stats, then one PMC pass (MFMA issue / busy, LDS)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_uv
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace --output-format csv -- python $R/scripts/bench_uv_backward.py 10 > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --kernel-trace -d $OUT/pmc1 -o pmc1 --output-format csv -- python $R/scripts/bench_uv_backward.py 3 > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU --kernel-trace -d $OUT/pmc2 -o pmc2 --output-format csv -- python $R/scripts/bench_uv_backward.py 3 > $OUT/pmc2.log 2>&1
tail -1 $OUT/trace.log
python - <<PY
import csv, glob, collections
f = glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True)
if f:
    for r in csv.DictReader(open(f[0])):
        if "k_uv" in r["Name"]:
            print("%-70s calls %5s avg_ns %10s" % (r["Name"][:70], r["Calls"], r["AverageNs"]))
for d in ("pmc1", "pmc2"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % d, recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            for k in ("k_uv_taylor_bf16x3", "k_uv_taylor_mixed", "k_uv_taylor", "k_uv_backward_reduce", "k_uv_backward<true>", "k_uv_backward<false>", "k_uv_backward"):
                if k in n:
                    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"])); break
        for k, cs in agg.items():
            print(d, k, {c: round(sum(v) / len(v) / 1e6, 3) for c, v in cs.items()}, "(millions per launch)")
PY
