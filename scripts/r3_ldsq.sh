#!/bin/bash
# LDS bank-conflict cycles of K7 by stage: PMC pass per ablated build
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for V in libtexgs.so libtexgs_noc1.so libtexgs_noc2.so libtexgs_nob.so libtexgs_aonly.so libtexgs_chunk.so; do
  OUT=$R/gpurun_out/ldsq_$(basename $V .so); rm -rf $OUT; mkdir -p $OUT
  TEXGS_LIB=$R/texture-gs_amd/$V rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --kernel-trace -d $OUT -o p --output-format csv -- python $R/bench.py --steps 1 --warmup 2 --no-cpu-baseline --no-kernel-table --streams 1 > /dev/null 2>&1
  python - <<PY
import csv, glob, collections
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_render_bwd" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("$V", {k: round(sum(v) / len(v) / 1e6, 2) for k, v in agg.items()}, "(millions per launch)")
PY
done
