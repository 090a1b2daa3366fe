cd $GRAFT_REPO_ROOT
for V in libtexgs.so libtexgs_fulllds.so; do
for MODE in "--streams 1" ""; do
TEXGS_LIB=$PWD/texture-gs_amd/$V timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --steps 12 --warmup 4 $MODE 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$V $MODE', d['value'], d['value_long'], {k:round(v['avg_us']) for k,v in (d.get('kernels') or {}).items() if 'render' in k or 'reduce' in k})
"
done
done
