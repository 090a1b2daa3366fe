cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_parity_gpu.py -m gpu -q -p no:cacheprovider -k "full_resolution" 2>&1 | grep -E "passed|failed|^E  |FAILED|Error" | head -30
grep subset gpurun_out/parity_report.jsonl | tail -2
