#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/parity_report.jsonl
echo "== gpu tests (contract + parity subset first)"
timeout 1500 python -m pytest tests -m gpu -q -rA 2>&1 | grep -E "passed|failed|FAILED|ERROR|Error|error|assert" | tail -30 > gpurun_out/gpu_tests.log
tail -12 gpurun_out/gpu_tests.log
echo "== ablations (serial)"
for v in "" _noc2 _nob _noapp _nocur _aonly _fnodense; do
  TEXGS_LIB=$PWD/texture-gs_amd/libtexgs$v.so timeout 300 python bench.py --streams 1 --no-cpu-baseline --steps 3 --warmup 2 2> gpurun_out/abl$v.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$v', d['value'], {k:round(v['avg_us']) for k,v in d['kernels'].items()})
"
done | tee gpurun_out/ablations.log
echo "== pipelined"
timeout 400 python bench.py --no-cpu-baseline --steps 10 --warmup 3 2> gpurun_out/bench_c3.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('c3 pipelined', d['value'], {k:round(v['avg_us']) for k,v in d['kernels'].items()}, d['reference_call_pattern'])
"
