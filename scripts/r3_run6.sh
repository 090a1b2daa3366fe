#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "== quick gpu tests"
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_contract_gpu.py tests/test_parity_c_oracle_gpu.py -m gpu -q -x 2>&1 | grep -E "passed|failed|FAILED|ERROR|rror" | tail -6
echo "== variants (serial)"
for v in "" _aonly _rawonly _noc2; do
  TEXGS_LIB=$PWD/texture-gs_amd/libtexgs$v.so timeout 300 python bench.py --streams 1 --no-cpu-baseline --steps 3 --warmup 2 2> gpurun_out/abl$v.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$v', d['value'], {k:round(v['avg_us']) for k,v in d['kernels'].items()})
"
done | tee gpurun_out/variants.log
TEXGS_LIB=$PWD/texture-gs_amd/libtexgs_stats.so timeout 300 python scripts/exp_stats.py c3 2>&1 | grep -v amdgpu.ids | grep K7 | tee gpurun_out/stats_c3.log
