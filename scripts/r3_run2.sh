#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "== stats"
TEXGS_LIB=$PWD/texture-gs_amd/libtexgs_stats.so timeout 300 python scripts/exp_stats.py c3 2>&1 | grep -v amdgpu.ids | tee gpurun_out/stats_c3.log
echo "== ablations (serial)"
for v in "" _noc2 _noc1 _nob _noapp _nocur _aonly _fnodense; do
  TEXGS_LIB=$PWD/texture-gs_amd/libtexgs$v.so timeout 300 python bench.py --streams 1 --no-cpu-baseline --steps 3 --warmup 2 2> gpurun_out/abl$v.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$v', d['value'], {k:round(v['avg_us']) for k,v in d['kernels'].items()})
"
done | tee gpurun_out/ablations.log
echo "== pmc"
bash scripts/pmc_quick.sh libtexgs.so 2>&1 | tail -12 | tee gpurun_out/pmc_quick.log
