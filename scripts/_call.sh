cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -q -p no:cacheprovider -k "extra_attrs" 2>&1 | grep -E "passed|failed|^E  |FAILED|Error" | head -30
