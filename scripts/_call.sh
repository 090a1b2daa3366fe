cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_parity_c_oracle_gpu.py -m gpu -q -p no:cacheprovider -k "stress" 2>&1 | grep -E "passed|failed|^E  |FAILED|Error" | cut -c1-600 | head -20
grep "hip_vs_c32/stress/bwd" gpurun_out/parity_report.jsonl | cut -c1-700
