#!/bin/bash
# usage (GPU box): scripts/r3_ab2.sh lib1.so lib2.so ...   -- serial C3 + C2 bench per library, binning kernel groups
cd $GRAFT_REPO_ROOT
for V in "$@"; do
for WL in c3 c2; do
TEXGS_LIB=$GRAFT_REPO_ROOT/texture-gs_amd/$V timeout 400 python bench.py --workload $WL --no-cpu-baseline --steps 6 --warmup 3 --streams 1 2> /dev/null | python -c "
import sys, json, re
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$V $WL', d['value'], {k:round(v['avg_us'],1) for k,v in d['kernels'].items() if k in ('scan','sort','duplicate','ranges')})
"
done
done
