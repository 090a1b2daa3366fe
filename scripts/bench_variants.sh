#!/bin/bash
# usage (on the GPU box): scripts/bench_variants.sh lib1.so lib2.so ...   -> one short C3 bench line per library
R=${GRAFT_REPO_ROOT:-/root/repo}
for lib in "$@"; do
  env TEXGS_LIB=$R/texture-gs_amd/$lib $BENCH_ENV timeout 300 python $R/bench.py --steps ${STEPS:-5} --warmup ${WARMUP:-3} --no-cpu-baseline ${BENCH_ARGS} 2>/dev/null | \
    python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        print(json.dumps({'lib': '$lib', 'views_per_s': d['value'], 'ms_per_view': d['ms_per_view'], 'dominant_solo_us': (d.get('roofline') or {}).get('solo_launch_us'), 'kernel_avg_us': {k: round(v['avg_us'], 1) for k, v in d['kernels'].items()}}))
"
done
