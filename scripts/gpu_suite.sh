#!/bin/bash
# usage (GPU box): scripts/gpu_suite.sh [pytest args]   -- the -m gpu suite (log + parity report under gpurun_out/), then smoke()
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T0=$(date +%s)
rm -f gpurun_out/parity_report.jsonl
timeout 1800 python -m pytest tests -m gpu -q -rA --durations=15 -p no:cacheprovider "$@" 2>&1 | grep -E "passed|failed|PASSED|FAILED|ERROR|rror|assert|^E  |^[0-9.]+s (call|setup)" | tail -220 > gpurun_out/gpu_tests.log
tail -3 gpurun_out/gpu_tests.log; grep -E "^FAILED|^ERROR|^E  " gpurun_out/gpu_tests.log | head -40
echo "[tests $(( $(date +%s) - T0 )) s]"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
echo "[$(( $(date +%s) - T0 )) s]"
