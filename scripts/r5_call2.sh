#!/bin/bash
# Round 5, GPU call 2: the 64-entry direct-mapped reservation tables (product) -- correctness, the new parity tests, K7 A/B.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
T0=$(date +%s)
rm -f gpurun_out/parity_report.jsonl
timeout 900 python -m pytest tests/test_contract_gpu.py tests/test_parity_gpu.py tests/test_gating_gpu.py tests/test_untextured_full_size_gpu.py \
   "tests/test_parity_c_oracle_gpu.py::test_backward_full_size_vs_c_oracle" "tests/test_parity_c_oracle_gpu.py::test_forward_full_size_vs_c_oracle" \
   tests/test_parity_c_oracle_gpu.py::test_band_limited_texture_c3 tests/test_parity_c_oracle_gpu.py::test_stress_scene_vs_c_oracle \
   tests/test_parity_c_oracle_gpu.py::test_more_than_65536_tiles_three_digit_tile_sort tests/test_parity_c_oracle_gpu.py::test_depth_sort_with_crowded_depth_bins \
   -m gpu -q -k "not c5" -p no:cacheprovider 2>&1 | tail -60 > gpurun_out/c2_tests_product.log
echo "product tests: $(tail -1 gpurun_out/c2_tests_product.log)  [$(( $(date +%s) - T0 )) s]"
grep -E "^FAILED|^ERROR" gpurun_out/c2_tests_product.log | head -20
TEXGS_ABI_ANY=1 bash scripts/ab_serial.sh libtexgs_r04.so 2>&1 | grep -v amdgpu.ids | tee gpurun_out/c2_ab.log
bash scripts/ab_serial.sh libtexgs.so libtexgs_g3.so libtexgs_g4q64.so libtexgs_q64.so 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/c2_ab.log
echo "[$(( $(date +%s) - T0 )) s]"
TEXGS_ABI_ANY=1 bash scripts/pipe.sh "" libtexgs_r04.so 2>&1 | tee gpurun_out/c2_pipe.log
bash scripts/pipe.sh "" libtexgs.so libtexgs_g3.so libtexgs_g4q64.so 2>&1 | tee -a gpurun_out/c2_pipe.log
echo "[$(( $(date +%s) - T0 )) s]"
for V in g3 g4q64; do
  TEXGS_LIB=$GRAFT_REPO_ROOT/texture-gs_amd/libtexgs_$V.so timeout 300 python -m pytest tests/test_parity_gpu.py tests/test_gating_gpu.py::test_backward_flavours_equal_the_full_backward tests/test_contract_gpu.py::test_block_reservations_tile_the_record_lists -m gpu -q -x -p no:cacheprovider 2>&1 | tail -5 > gpurun_out/c2_tests_$V.log
  echo "$V tests: $(tail -1 gpurun_out/c2_tests_$V.log)  [$(( $(date +%s) - T0 )) s]"
done
