#!/bin/bash
# usage (GPU box): scripts/ab_env.sh "VAR=1 VAR2=x" "VAR=0" ...   -- C3 bench per environment setting: serial kernel table and
# pipelined views/s (e.g. "TEXGS_ITEMS=1" "TEXGS_ITEMS=0", or "TEXGS_LIB=$PWD/texture-gs_amd/libtexgs_x.so")
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for V in "$@"; do
for MODE in "--streams 1" ""; do
env $V timeout 400 python bench.py --no-cpu-baseline --no-extra-legs --steps 8 --warmup 3 $MODE 2> gpurun_out/ab_err.log | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$V', '$MODE', d['value'], {k:round(v['avg_us']) for k,v in (d.get('kernels') or {}).items()})
"
done
done
tail -3 gpurun_out/ab_err.log
