cd $GRAFT_REPO_ROOT
for S in 2 3 4; do
timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --no-kernel-table --steps 12 --warmup 4 --streams $S 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('streams $S', d['value'], d['value_long'])
"
done
scripts/gpu_suite.sh
