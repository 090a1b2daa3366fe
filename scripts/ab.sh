#!/bin/bash
# usage (GPU box): scripts/r4_ab.sh lib1.so lib2.so ...   -- C3 bench per library: serial kernel table + pipelined views/s
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for V in "$@"; do
for MODE in "--streams 1" ""; do
TEXGS_LIB=$GRAFT_REPO_ROOT/texture-gs_amd/$V timeout 400 python bench.py --no-cpu-baseline --steps 8 --warmup 3 $MODE 2> gpurun_out/ab_err.log | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$V', '$MODE', d['value'], {k:round(v['avg_us']) for k,v in d['kernels'].items()}, (d.get('reference_call_pattern') or {}).get('views_per_s'))
"
done
done
tail -3 gpurun_out/ab_err.log
