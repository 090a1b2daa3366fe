#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_losses.py tests/test_abi.py -m gpu -x -q 2>&1 | tail -4
STEPS=10 WARMUP=3 bash scripts/bench_variants.sh libtexgs.so libtexgs_lds12.so libtexgs.so libtexgs_lds12.so
