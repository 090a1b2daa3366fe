#!/bin/bash
# usage (GPU box): scripts/pmc_quick.sh <tag> ["ENV=.. ENV2=.."]  -> SQ / traffic counters of the blend kernels for one short serial C3 run
# (separate --pmc passes, kernel trace only: the node-safe combination); output under gpurun_out/pmcq_<tag>/summary.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; ENVS=$2
OUT=$R/gpurun_out/pmcq_$TAG
rm -rf $OUT; mkdir -p $OUT
B="python $R/bench.py --steps 1 --warmup 2 --no-cpu-baseline --no-kernel-table --streams 1"
env $ENVS rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --kernel-trace -d $OUT/p1 -o p1 --output-format csv -- $B > /dev/null 2>&1
env $ENVS rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace -d $OUT/p2 -o p2 --output-format csv -- $B > /dev/null 2>&1
env $ENVS rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/p3 -o p3 --output-format csv -- $B > /dev/null 2>&1
env $ENVS rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $OUT/p4 -o p4 --output-format csv -- $B > /dev/null 2>&1
python - <<PY | tee $OUT/summary.txt
import csv, glob, collections
names = ("k_render_fwd", "k_render_bwd_stream", "k_render_bwd", "k_texgrad_reduce", "k_preprocess_bwd")
def short(n):
    for k in names:
        if k + "<" in n or k + "I" in n or n.startswith(k + "(") or (k in n and not any(k2 != k and k in k2 and k2 in n for k2 in names)): return k
    return None
for d in ("p1", "p2", "p3", "p4"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % d, recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        dur = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k:
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
                if "Start_Timestamp" in r: dur[k].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
        for k, cs in agg.items():
            print("$TAG", d, k, {c: round(sum(v) / len(v) / 1e6, 2) for c, v in cs.items()}, "(millions per launch)",
                  "us under pmc", round(sum(dur[k]) / max(len(dur[k]), 1), 1))
PY
