#!/bin/bash
# round 3: timing-only ablations of the final K6 / K7 (scripts/exp_instrument.py builds) + the dynamic work counters of one C3 view
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
for V in libtexgs.so libtexgs_noc1.so libtexgs_noc2.so libtexgs_nob.so libtexgs_aonly.so libtexgs_chunk.so libtexgs_noacc.so libtexgs_nobins.so libtexgs_nostore.so libtexgs_fnodense.so libtexgs_fnotest.so; do
TEXGS_LIB=$GRAFT_REPO_ROOT/texture-gs_amd/$V timeout 300 python bench.py --no-cpu-baseline --steps 4 --warmup 3 --streams 1 2> /dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$V', {k:round(v['avg_us']) for k,v in d['kernels'].items() if k in ('render_fwd','render_bwd','texgrad_reduce','preprocess_bwd')})
"
done
TEXGS_LIB=$PWD/texture-gs_amd/libtexgs_stats.so timeout 300 python scripts/exp_stats.py c3 2>&1 | grep -v amdgpu.ids
} | tee gpurun_out/r3_ablate.log
