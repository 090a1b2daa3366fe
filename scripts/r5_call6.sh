#!/bin/bash
# Round 5, GPU call 6: K7 with the next segment's shading records prefetched (libtexgs.so) vs not (libtexgs_nopre.so)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
T0=$(date +%s)
timeout 600 python -m pytest tests/test_parity_gpu.py tests/test_contract_gpu.py "tests/test_parity_c_oracle_gpu.py::test_backward_full_size_vs_c_oracle" \
   -m gpu -q -x -p no:cacheprovider 2>&1 | tail -15 > gpurun_out/c6_tests.log
echo "tests: $(tail -1 gpurun_out/c6_tests.log)  [$(( $(date +%s) - T0 )) s]"
grep -E "^FAILED|^ERROR" gpurun_out/c6_tests.log | head
bash scripts/ab_serial.sh libtexgs_nopre.so libtexgs.so libtexgs_nopre.so libtexgs.so 2>&1 | grep -v amdgpu.ids | tee gpurun_out/c6_ab.log
echo "[$(( $(date +%s) - T0 )) s]"
bash scripts/pipe.sh "" libtexgs_nopre.so libtexgs.so libtexgs_nopre.so libtexgs.so 2>&1 | tee gpurun_out/c6_pipe.log
echo "[$(( $(date +%s) - T0 )) s]"
timeout 300 python bench.py --leg iteration --steps 10 --warmup 3 2>gpurun_out/c6_iter.err | tee gpurun_out/c6_iteration_fp32.json | cut -c1-1400
echo "[$(( $(date +%s) - T0 )) s]"
