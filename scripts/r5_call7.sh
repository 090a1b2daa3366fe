#!/bin/bash
# Round 5, GPU call 7: the fused UVNet backward kernel -- parity tests, then the iteration leg with it
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
T0=$(date +%s)
timeout 400 python -m pytest tests/test_uvnet.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -25 > gpurun_out/c7_tests.log
echo "tests: $(tail -1 gpurun_out/c7_tests.log)  [$(( $(date +%s) - T0 )) s]"
grep -E "^FAILED|^ERROR|Error|assert" gpurun_out/c7_tests.log | head
timeout 300 python bench.py --leg iteration --steps 10 --warmup 3 2>gpurun_out/c7_iter.err | tee gpurun_out/c7_iteration_fp32.json | cut -c1-1600
echo "[$(( $(date +%s) - T0 )) s]"
TEXGS_UV_PRECISION=bf16x3 timeout 300 python bench.py --leg iteration --steps 10 --warmup 3 2>>gpurun_out/c7_iter.err | tee gpurun_out/c7_iteration_bf16x3.json | cut -c1-1600
echo "[$(( $(date +%s) - T0 )) s]"
tail -5 gpurun_out/c7_iter.err
grep uv_backward gpurun_out/parity_report.jsonl | tail -8
