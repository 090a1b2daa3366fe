cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_contract_gpu.py -m gpu -q -p no:cacheprovider -k "rccl" 2>&1 | grep -E "passed|failed|^E  |FAILED|Error" | head
