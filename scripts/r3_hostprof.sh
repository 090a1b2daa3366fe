#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 300 python -m cProfile -o /tmp/c2.prof bench.py --workload c2 --no-cpu-baseline --no-kernel-table --steps 40 --warmup 5 2>/dev/null | cut -c1-200
python - <<'PY'
import pstats
p = pstats.Stats('/tmp/c2.prof'); p.sort_stats('tottime').print_stats(28)
PY
