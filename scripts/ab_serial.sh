#!/bin/bash
# usage (GPU box): scripts/ab_serial.sh lib1.so lib2.so ...   -- C3 serial kernel table per library (short)
cd $GRAFT_REPO_ROOT
for V in "$@"; do
TEXGS_LIB=$GRAFT_REPO_ROOT/texture-gs_amd/$V timeout 300 python bench.py --no-cpu-baseline --steps 4 --warmup 2 --streams 1 2> gpurun_out/ab_err.log | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$V', d['value'], {k:round(v['avg_us']) for k,v in d['kernels'].items()})
"
done
tail -2 gpurun_out/ab_err.log
