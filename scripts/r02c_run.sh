#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_parity_c_oracle_gpu.py -m gpu -x -q 2>&1 | tail -5 > gpurun_out/t3.log
cat gpurun_out/t3.log
STEPS=5 WARMUP=2 bash scripts/bench_variants.sh libtexgs_base.so libtexgs_v2.so libtexgs.so
