#!/bin/bash
# final confirmation of the committed tree: the whole -m gpu suite (log + parity report), smoke()
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T0=$(date +%s)
rm -f gpurun_out/parity_report.jsonl
timeout 1500 python -m pytest tests -m gpu -q -rA --durations=15 -p no:cacheprovider 2>&1 | grep -E "passed|failed|PASSED|FAILED|ERROR|rror|^[0-9.]+s (call|setup)" | tail -170 > gpurun_out/gpu_tests.log
tail -3 gpurun_out/gpu_tests.log; grep -E "^FAILED|^ERROR" gpurun_out/gpu_tests.log | head
echo "[tests $(( $(date +%s) - T0 )) s]"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
echo "[$(( $(date +%s) - T0 )) s]"
