#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "== quick gpu tests"
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_contract_gpu.py -m gpu -q -x 2>&1 | tail -4
echo "== variants (serial)"
for v in "" _bq64 _bq96 _aonly _fnodense _noc2; do
  TEXGS_LIB=$PWD/texture-gs_amd/libtexgs$v.so timeout 300 python bench.py --streams 1 --no-cpu-baseline --steps 3 --warmup 2 2> gpurun_out/abl$v.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$v', d['value'], {k:round(v['avg_us']) for k,v in d['kernels'].items()})
"
done | tee gpurun_out/variants.log
