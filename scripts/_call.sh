cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_uvnet.py -m gpu -q -p no:cacheprovider -k "c3" 2>&1 | grep -E "passed|failed|^E  |FAILED|Error" | head -30
grep c3_image gpurun_out/parity_report.jsonl | tail -1
