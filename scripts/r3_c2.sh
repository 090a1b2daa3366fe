#!/bin/bash
cd $GRAFT_REPO_ROOT
for st in 1 3; do
timeout 300 python bench.py --workload c2 --streams $st --no-cpu-baseline --steps 6 --warmup 3 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('[c2 streams $st]', d['value'], d['ms_per_step_percentiles'], {k:round(v['avg_us']) for k,v in d['kernels'].items()})
"
done
