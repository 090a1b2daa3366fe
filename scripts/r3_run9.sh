#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_contract_gpu.py tests/test_parity_c_oracle_gpu.py tests/test_losses.py -m gpu -q -x 2>&1 | grep -E "passed|failed|FAILED|ERROR|rror" | tail -6
python scripts/bench_next_rows.py 20 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
for k, v in d.items(): print(k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items()})
"
bash scripts/r3_quick.sh notest
