#!/bin/bash
# Round 5, GPU call 8: UV backward tests (kink-safe points); when / where K7's blocks run (trace build)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
T0=$(date +%s)
timeout 400 python -m pytest tests/test_uvnet.py -m gpu -q -p no:cacheprovider 2>&1 | tail -12 > gpurun_out/c8_tests.log
echo "tests: $(tail -1 gpurun_out/c8_tests.log)  [$(( $(date +%s) - T0 )) s]"
grep -E "^FAILED|^ERROR|Error|assert" gpurun_out/c8_tests.log | head
TEXGS_ABI_ANY=1 TEXGS_LIB=$GRAFT_REPO_ROOT/texture-gs_amd/libtexgs_trace.so timeout 300 python scripts/k7_trace.py 2>gpurun_out/c8_trace.err | tee gpurun_out/c8_k7_trace.json | cut -c1-6000
tail -3 gpurun_out/c8_trace.err
echo "[$(( $(date +%s) - T0 )) s]"
