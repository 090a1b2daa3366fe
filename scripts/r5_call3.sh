#!/bin/bash
# Round 5, GPU call 3: fixed / new tests, K7 variants (gather x slot runs), flavours of the winner, the iteration leg.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
T0=$(date +%s)
rm -f gpurun_out/parity_report.jsonl
timeout 1100 python -m pytest tests/test_parity_gpu.py::test_bin_sort_render_forward_can_run_twice tests/test_untextured_full_size_gpu.py \
   "tests/test_parity_c_oracle_gpu.py::test_backward_full_size_vs_c_oracle" tests/test_parity_c_oracle_gpu.py::test_band_limited_texture_c3 \
   tests/test_parity_c_oracle_gpu.py::test_stress_scene_vs_c_oracle tests/test_parity_c_oracle_gpu.py::test_more_than_65536_tiles_three_digit_tile_sort \
   tests/test_parity_c_oracle_gpu.py::test_depth_sort_with_crowded_depth_bins tests/test_uvnet.py::test_split_bf16_kernel_vs_f32_kernel_and_float64 \
   tests/test_gating_gpu.py::test_train_eval_interleave_costs_one_late_handoff_per_switch_and_holds_no_more_memory \
   tests/test_ws2_gpu.py::test_bench_gpus8_code_path_on_one_gpu tests/test_contract_gpu.py::test_c5_full_size_vs_c_oracle \
   tests/test_contract_gpu.py::test_c5_band_limited_texture_literal_tolerance tests/test_contract_gpu.py::test_block_reservations_tile_the_record_lists \
   -m gpu -q -p no:cacheprovider --durations=12 2>&1 | tail -150 > gpurun_out/c3_tests_product.log
echo "product tests: $(tail -1 gpurun_out/c3_tests_product.log)  [$(( $(date +%s) - T0 )) s]"
grep -E "^FAILED|^ERROR" gpurun_out/c3_tests_product.log | head -20
bash scripts/ab_serial.sh libtexgs.so libtexgs_runs.so libtexgs_g4q64.so libtexgs_g4q64r.so libtexgs_g3r.so libtexgs_g3q96r.so 2>&1 | grep -v amdgpu.ids | tee gpurun_out/c3_ab.log
echo "[$(( $(date +%s) - T0 )) s]"
bash scripts/pipe.sh "" libtexgs.so libtexgs_g4q64.so libtexgs_g4q64r.so libtexgs_g3q96r.so 2>&1 | tee gpurun_out/c3_pipe.log
echo "[$(( $(date +%s) - T0 )) s]"
TEXGS_LIB=$GRAFT_REPO_ROOT/texture-gs_amd/libtexgs_g4q64r.so timeout 300 python scripts/bench_variants.py full texture_only frozen_texture diff_gauss 2>gpurun_out/c3_variants.err | cut -c1-420 | tee gpurun_out/c3_variants_g4q64r.jsonl
echo "[$(( $(date +%s) - T0 )) s]"
TEXGS_LIB=$GRAFT_REPO_ROOT/texture-gs_amd/libtexgs_g4q64r.so timeout 300 python bench.py --leg iteration --steps 8 --warmup 3 2>gpurun_out/c3_iter.err | tee gpurun_out/c3_iteration.json | cut -c1-1500
echo "[$(( $(date +%s) - T0 )) s]"
TEXGS_LIB=$GRAFT_REPO_ROOT/texture-gs_amd/libtexgs_g4q64r.so timeout 300 python -m pytest tests/test_parity_gpu.py tests/test_gating_gpu.py::test_backward_flavours_equal_the_full_backward tests/test_contract_gpu.py::test_block_reservations_tile_the_record_lists tests/test_contract_gpu.py::test_texture_gradient_bins_full_and_disabled_paths_agree "tests/test_parity_c_oracle_gpu.py::test_backward_full_size_vs_c_oracle" -m gpu -q -p no:cacheprovider 2>&1 | tail -8 > gpurun_out/c3_tests_g4q64r.log
echo "g4q64r tests: $(tail -1 gpurun_out/c3_tests_g4q64r.log)  [$(( $(date +%s) - T0 )) s]"
tail -3 gpurun_out/c3_iter.err
