#!/bin/bash
# gate for binning changes: integer stages bit-exact vs the C oracle (small, C2, C3, stress, 66 049 tiles) + serial C3 / C2 bench
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_parity_c_oracle_gpu.py -m gpu -q -x -k "bit_exact or stress or 65536" 2>&1 | grep -E "passed|failed|FAILED|ERROR|rror" | tail -4
timeout 400 python bench.py --no-cpu-baseline --steps 10 --warmup 3 --streams 1 2> /dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('c3 serial', d['value'], {k:round(v['avg_us']) for k,v in d['kernels'].items()})
"
timeout 400 python bench.py --workload c2 --no-cpu-baseline --steps 10 --warmup 3 2> /dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('c2', d['value'], {k:round(v['avg_us']) for k,v in d['kernels'].items()})
"
