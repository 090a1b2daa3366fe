#!/bin/bash
# Round 5, GPU call 13: UV backward reduce with batched loads -- tests, the UV kernels under rocprofv3 (stats + MFMA / LDS counters),
# the iteration leg again
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
T0=$(date +%s)
timeout 400 python -m pytest tests/test_uvnet.py -m gpu -q -p no:cacheprovider 2>&1 | tail -6 > gpurun_out/c13_tests.log
echo "tests: $(tail -1 gpurun_out/c13_tests.log)  [$(( $(date +%s) - T0 )) s]"
grep -E "^FAILED|^ERROR" gpurun_out/c13_tests.log | head
bash scripts/prof_uv_backward.sh > gpurun_out/uv_kernels_profile.txt 2>&1
cat gpurun_out/uv_kernels_profile.txt | cut -c1-600
echo "[$(( $(date +%s) - T0 )) s]"
timeout 300 python bench.py --leg iteration --steps 10 --warmup 3 > gpurun_out/bench_iteration_fp32.json 2> gpurun_out/bench_iteration_fp32.err
TEXGS_UV_PRECISION=bf16x3 timeout 300 python bench.py --leg iteration --steps 10 --warmup 3 > gpurun_out/bench_iteration_bf16x3.json 2> gpurun_out/bench_iteration_bf16x3.err
cut -c1-1500 gpurun_out/bench_iteration_fp32.json; cut -c1-1500 gpurun_out/bench_iteration_bf16x3.json
bash scripts/prof_iteration.sh > gpurun_out/iteration_profile.txt 2>&1
head -24 gpurun_out/iteration_profile.txt
echo "[$(( $(date +%s) - T0 )) s]"
