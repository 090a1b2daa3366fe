#!/bin/bash
# Round 5, GPU call 15: the mixed-precision UV forward (value f32, tangents split-bf16): tests, counters, the iteration leg with it
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
T0=$(date +%s)
timeout 400 python -m pytest tests/test_uvnet.py -m gpu -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|^FAILED|^ERROR|assert|Error" | head -12
echo "[$(( $(date +%s) - T0 )) s]"
grep uv_taylor_mixed gpurun_out/parity_report.jsonl | tail -2
bash scripts/prof_uv_backward.sh > gpurun_out/uv_kernels_profile.txt 2>&1
grep -v "tool finalization" gpurun_out/uv_kernels_profile.txt | cut -c1-500
echo "[$(( $(date +%s) - T0 )) s]"
TEXGS_UV_PRECISION=mixed timeout 300 python bench.py --leg iteration --steps 10 --warmup 3 > gpurun_out/bench_iteration_mixed.json 2> gpurun_out/bench_iteration_mixed.err
cut -c1-1500 gpurun_out/bench_iteration_mixed.json; tail -2 gpurun_out/bench_iteration_mixed.err
echo "[$(( $(date +%s) - T0 )) s]"
