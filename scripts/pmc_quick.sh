#!/bin/bash
# usage (GPU box): scripts/pmc_quick.sh <lib.so>  -> SQ counters of the blend kernels for one short bench run
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmcq_$(basename $1 .so)
rm -rf $OUT; mkdir -p $OUT
export TEXGS_LIB=$R/texture-gs_amd/$1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --kernel-trace -d $OUT/p1 -o p1 --output-format csv -- python $R/bench.py --steps 1 --warmup 2 --no-cpu-baseline --no-kernel-table --streams 1 > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace -d $OUT/p2 -o p2 --output-format csv -- python $R/bench.py --steps 1 --warmup 2 --no-cpu-baseline --no-kernel-table --streams 1 > /dev/null 2>&1
python - <<PY
import csv, glob, collections
for d in ("p1", "p2"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % d, recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            for k in ("k_render_fwd", "k_render_bwd", "k_texgrad_reduce", "k_preprocess_bwd"):
                if k in n: agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in agg.items():
            print(k, {c: round(sum(v) / len(v) / 1e6, 2) for c, v in cs.items()}, "(millions)")
PY
