#!/bin/bash
cd $GRAFT_REPO_ROOT
for S in 2 3 4; do for O in accumulate none; do
timeout 300 python bench.py --no-cpu-baseline --no-kernel-table --steps 15 --warmup 4 --streams $S --order $O 2> /dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('streams $S order $O', d['value'], d['ms_per_step_percentiles'])
"
done; done
