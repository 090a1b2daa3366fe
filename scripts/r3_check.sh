#!/bin/bash
# quick gate after a kernel change: HIP vs fp64 oracle (fwd + autograd bwd, 20 cases) and the serial / pipelined C3 bench
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_parity_gpu.py -m gpu -q -x 2>&1 | grep -E "passed|failed|FAILED|ERROR|rror" | tail -4
for S in 1 3; do
timeout 400 python bench.py --no-cpu-baseline --steps 10 --warmup 3 --streams $S 2> /dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('c3 streams $S', d['value'], {k:round(v['avg_us']) for k,v in d['kernels'].items()}, d['reference_call_pattern']['views_per_s'])
"
done
