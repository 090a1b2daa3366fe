#!/bin/bash
# Round 5, GPU call 5: two-register-set lock-step loops (K6, K7 stage A) vs the committed kernels; the iteration leg after the
# UVNet-backward fix.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
T0=$(date +%s)
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_gating_gpu.py "tests/test_parity_c_oracle_gpu.py::test_forward_full_size_vs_c_oracle" \
   "tests/test_parity_c_oracle_gpu.py::test_backward_full_size_vs_c_oracle" "tests/test_parity_c_oracle_gpu.py::test_integer_stages_bit_exact" \
   tests/test_untextured_full_size_gpu.py tests/test_contract_gpu.py::test_block_reservations_tile_the_record_lists tests/test_uvnet.py \
   -m gpu -q -p no:cacheprovider 2>&1 | tail -15 > gpurun_out/c5_tests.log
echo "tests: $(tail -1 gpurun_out/c5_tests.log)  [$(( $(date +%s) - T0 )) s]"
grep -E "^FAILED|^ERROR" gpurun_out/c5_tests.log | head
bash scripts/ab_serial.sh libtexgs_head.so libtexgs.so libtexgs_head.so libtexgs.so 2>&1 | grep -v amdgpu.ids | tee gpurun_out/c5_ab.log
echo "[$(( $(date +%s) - T0 )) s]"
bash scripts/pipe.sh "" libtexgs_head.so libtexgs.so libtexgs_head.so libtexgs.so 2>&1 | tee gpurun_out/c5_pipe.log
echo "[$(( $(date +%s) - T0 )) s]"
timeout 300 python bench.py --leg iteration --steps 10 --warmup 3 2>gpurun_out/c5_iter.err | tee gpurun_out/c5_iteration_fp32.json | cut -c1-1400
TEXGS_UV_PRECISION=bf16x3 timeout 300 python bench.py --leg iteration --steps 10 --warmup 3 2>>gpurun_out/c5_iter.err | tee gpurun_out/c5_iteration_bf16x3.json | cut -c1-1400
echo "[$(( $(date +%s) - T0 )) s]"
timeout 200 python bench.py --workload c2 --no-cpu-baseline --no-kernel-table 2>/dev/null | cut -c1-300
TEXGS_LIB=$GRAFT_REPO_ROOT/texture-gs_amd/libtexgs_head.so timeout 200 python bench.py --workload c2 --no-cpu-baseline --no-kernel-table 2>/dev/null | cut -c1-300
tail -3 gpurun_out/c5_iter.err
