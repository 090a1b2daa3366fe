#!/bin/bash
# usage: r3_quick.sh [test|notest] [lib suffixes...]   quick parity subset + serial bench of the product and of the listed variant libs
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
if [ "$1" = "test" ]; then
  timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_contract_gpu.py tests/test_parity_c_oracle_gpu.py -m gpu -q -x 2>&1 | grep -E "passed|failed|FAILED|ERROR|rror" | tail -6
fi
shift
for v in "" "$@"; do
  TEXGS_LIB=$PWD/texture-gs_amd/libtexgs$v.so timeout 300 python bench.py --streams 1 --no-cpu-baseline --steps 4 --warmup 2 2> gpurun_out/abl$v.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('[$v]', d['value'], {k:round(v['avg_us']) for k,v in d['kernels'].items()})
"
done | tee gpurun_out/variants.log
TEXGS_LIB=$PWD/texture-gs_amd/libtexgs.so timeout 300 python bench.py --workload c2 --streams 1 --no-cpu-baseline --steps 4 --warmup 2 2> gpurun_out/c2.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('[c2 serial]', d['value'], {k:round(v['avg_us']) for k,v in d['kernels'].items()})
"
