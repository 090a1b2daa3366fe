#!/bin/bash
# final confirmation of the committed tree: smoke(), the UV + contract tests, the default bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T0=$(date +%s)
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_uvnet.py tests/test_contract_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3
echo "[$(( $(date +%s) - T0 )) s]"
timeout 600 python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err
cut -c1-900 gpurun_out/final_bench.json
echo "[$(( $(date +%s) - T0 )) s]"
