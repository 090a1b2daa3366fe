#!/bin/bash
# Round 5, GPU call 1: correctness of the reservation path (product) and of the occupancy variants, then K7 A/B timings.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
T0=$(date +%s)
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_gating_gpu.py tests/test_contract_gpu.py "tests/test_parity_c_oracle_gpu.py::test_backward_full_size_vs_c_oracle" -m gpu -q -k "not c5_full" -p no:cacheprovider 2>&1 | tail -25 > gpurun_out/c1_tests_product.log
echo "product tests: $(tail -1 gpurun_out/c1_tests_product.log)  [$(( $(date +%s) - T0 )) s]"
for V in sb g3 g4q64 g3q96 q64; do
  TEXGS_LIB=$GRAFT_REPO_ROOT/texture-gs_amd/libtexgs_$V.so timeout 400 python -m pytest tests/test_parity_gpu.py tests/test_gating_gpu.py::test_backward_flavours_equal_the_full_backward tests/test_contract_gpu.py::test_block_reservations_tile_the_record_lists tests/test_contract_gpu.py::test_texture_gradient_bins_full_and_disabled_paths_agree -m gpu -q -p no:cacheprovider 2>&1 | tail -12 > gpurun_out/c1_tests_$V.log
  echo "$V tests: $(tail -1 gpurun_out/c1_tests_$V.log)  [$(( $(date +%s) - T0 )) s]"
done
TEXGS_ABI_ANY=1 bash scripts/ab_serial.sh libtexgs_r04.so 2>&1 | tee gpurun_out/c1_ab.log
bash scripts/ab_serial.sh libtexgs.so libtexgs_sb.so libtexgs_g3.so libtexgs_g4q64.so libtexgs_g3q96.so libtexgs_q64.so 2>&1 | tee -a gpurun_out/c1_ab.log
echo "[$(( $(date +%s) - T0 )) s]"
TEXGS_ABI_ANY=1 bash scripts/pipe.sh "" libtexgs_r04.so 2>&1 | tee gpurun_out/c1_pipe.log
bash scripts/pipe.sh "" libtexgs.so libtexgs_g3.so libtexgs_g4q64.so libtexgs_q64.so 2>&1 | tee -a gpurun_out/c1_pipe.log
echo "[$(( $(date +%s) - T0 )) s]"
grep -h -E "FAILED|Error|error" gpurun_out/c1_tests_*.log | head -20
