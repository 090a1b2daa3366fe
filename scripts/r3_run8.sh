#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/parity_report.jsonl
timeout 1500 python -m pytest tests -m gpu -q -rA 2>&1 | grep -E "passed|failed|FAILED|ERROR|Error|error|assert" | tail -20 > gpurun_out/gpu_tests.log
tail -8 gpurun_out/gpu_tests.log
bash scripts/prof_next_rows.sh 2>&1 | tail -30
