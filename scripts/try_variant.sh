#!/bin/bash
# usage: scripts/try_variant.sh "<extra hipcc flags>"  -> rebuild render.hip with flags, run short bench on GPU
cd /root/repo
touch texture-gs_amd/csrc/render.hip
TEXGS_EXTRA_FLAGS="-DTEXGS_EXPERIMENTS $1" python texture-gs_amd/build.py > /dev/null 2>&1 || { echo BUILD FAILED; exit 1; }
/usr/local/graft/bin/gpurun --timeout 600 -- 'timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith(\"{\"):
        d=json.loads(l); print(d[\"value\"], {k:round(v[\"avg_us\"]) for k,v in d[\"kernels\"].items()})
"' 2>&1 | tail -1
