#!/bin/bash
# usage (GPU box): scripts/r4_pipe.sh "<bench args>" lib1.so lib2.so ...  -- pipelined C3 views/s per library (no kernel table)
cd $GRAFT_REPO_ROOT
A=$1; shift
for V in "$@"; do
TEXGS_LIB=$GRAFT_REPO_ROOT/texture-gs_amd/$V timeout 300 python bench.py --no-cpu-baseline --no-kernel-table --steps 15 --warmup 4 $A 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$V', '$A', d['value'], d['ms_per_step_percentiles']['median'], round(d['roofline']['avg_launch_us']))
"
done
