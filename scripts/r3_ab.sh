#!/bin/bash
# usage (GPU box): scripts/r3_ab.sh <kernel list regex> lib1.so lib2.so ...   -- serial C3 bench per library, selected kernel times
cd $GRAFT_REPO_ROOT
K=$1; shift
for rep in 1 2; do
for V in "$@"; do
TEXGS_LIB=$GRAFT_REPO_ROOT/texture-gs_amd/$V timeout 400 python bench.py --no-cpu-baseline --steps 6 --warmup 3 --streams 1 2> /dev/null | python -c "
import sys, json, re
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$V', d['value'], {k:round(v['avg_us']) for k,v in d['kernels'].items() if re.search('$K', k)})
"
done
done
