#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_contract_gpu.py tests/test_parity_c_oracle_gpu.py -m gpu -q -x 2>&1 | grep -E "passed|failed|FAILED|ERROR|rror" | tail -6
for S in 1 3; do
timeout 400 python bench.py --no-cpu-baseline --steps 10 --warmup 3 --streams $S 2> /dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('c3 streams $S', d['value'], {k:round(v['avg_us']) for k,v in d['kernels'].items()}, d['reference_call_pattern']['views_per_s'])
"
done
