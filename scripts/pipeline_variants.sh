#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_contract_gpu.py -m gpu -x -q -k "pipeline or c4 or zero_grad" 2>&1 | tail -8 > gpurun_out/t2.log
for cfg in "1 backward" "2 backward" "2 accumulate" "2 none" "3 accumulate" "3 none" "4 none"; do
  set -- $cfg
  timeout 300 python bench.py --workload c3 --no-cpu-baseline --streams $1 --order $2 > gpurun_out/b_c3_s$1_$2.json 2> gpurun_out/b_c3_s$1_$2.err
done
cat gpurun_out/t2.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/b_c3_s*_*.json')):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, j['value'], j['ms_per_step_percentiles'], {k:round(v['avg_us']) for k,v in j['kernels'].items()})
    except Exception as e: print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-800:])
PY
