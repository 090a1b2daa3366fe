#!/bin/bash
# usage: scripts/kres.sh <render.o> [pattern]  -- VGPR / SGPR / LDS / spills of the kernels of one hipcc -c object (from its code object notes)
OBJ=$1; PAT=${2:-render}
TMP=$(mktemp -d)
/opt/rocm/lib/llvm/bin/llvm-objcopy --dump-section .hip_fatbin=$TMP/fat.bin $OBJ 2>/dev/null
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$TMP/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$TMP/k.co 2>/dev/null
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $TMP/k.co | python3 -c "
import sys, re
txt = sys.stdin.read()
for blk in re.split(r'\n\s+- \.agpr_count', txt)[1:]:
    m = re.search(r'\.name:\s+(\S+)', blk)
    if not m or '$PAT' not in m.group(1): continue
    g = lambda k: (re.search(r'\.' + k + r':\s+(\d+)', blk) or [None, '?'])[1]
    print(m.group(1)[:60], 'vgpr', g('vgpr_count'), 'sgpr', g('sgpr_count'), 'lds', g('group_segment_fixed_size'), 'spill', g('vgpr_spill_count'), 'scratch', g('private_segment_fixed_size'))
"
rm -rf $TMP
