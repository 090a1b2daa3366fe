cd $GRAFT_REPO_ROOT
for V in libtexgs_base.so libtexgs.so libtexgs_base.so libtexgs.so; do
for MODE in "--streams 1" ""; do
TEXGS_LIB=$PWD/texture-gs_amd/$V timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --steps 16 --warmup 4 $MODE 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$V $MODE', d['value'], d['value_long'], {k:round(v['avg_us']) for k,v in (d.get('kernels') or {}).items() if 'render' in k or 'reduce' in k})
"
done
done
timeout 600 python -m pytest tests/test_parity_gpu.py tests/test_contract_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | grep -E "passed|failed|^E  |FAILED|Error" | head
