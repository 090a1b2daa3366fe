#!/bin/bash
# Round 5, GPU call 11: K7 depth split inside one launch (experiment build, static buffers: single-stream legs only); finer K7
# priority levels; K6 priority
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TEXGS_ABI_ANY=1
T0=$(date +%s)
TEXGS_LIB=$GRAFT_REPO_ROOT/texture-gs_amd/libtexgs_split.so timeout 500 python -m pytest tests/test_parity_gpu.py "tests/test_contract_gpu.py::test_block_reservations_tile_the_record_lists" \
   "tests/test_parity_c_oracle_gpu.py::test_backward_full_size_vs_c_oracle" "tests/test_parity_c_oracle_gpu.py::test_integer_stages_bit_exact" \
   -m gpu -q -x -p no:cacheprovider 2>&1 | tail -15 > gpurun_out/c11_tests_split.log
echo "split tests: $(tail -1 gpurun_out/c11_tests_split.log)  [$(( $(date +%s) - T0 )) s]"
grep -E "^FAILED|^ERROR|Error" gpurun_out/c11_tests_split.log | head
bash scripts/ab_serial.sh libtexgs_base.so libtexgs.so libtexgs_k7p3.so libtexgs_k6p1.so libtexgs_k6p1k7p3.so libtexgs_split.so libtexgs_base.so libtexgs.so libtexgs_k7p3.so libtexgs_k6p1.so libtexgs_k6p1k7p3.so libtexgs_split.so 2>&1 | grep -v amdgpu.ids | tee gpurun_out/c11_ab.log
echo "[$(( $(date +%s) - T0 )) s]"
bash scripts/pipe.sh "" libtexgs.so libtexgs_k7p3.so libtexgs_k6p1.so libtexgs_k6p1k7p3.so libtexgs.so libtexgs_k7p3.so libtexgs_k6p1.so libtexgs_k6p1k7p3.so 2>&1 | tee gpurun_out/c11_pipe.log
echo "[$(( $(date +%s) - T0 )) s]"
for L in trace splittrace; do
  NB=10016; [ $L = splittrace ] && NB=20032
  K7_TRACE_NB=$NB TEXGS_LIB=$GRAFT_REPO_ROOT/texture-gs_amd/libtexgs_$L.so timeout 200 python scripts/k7_trace.py 2>/dev/null > gpurun_out/c11_$L.json
  python - <<PY
import json
d = json.load(open("gpurun_out/c11_$L.json"))
for r in d["k7_trace"]:
    print("$L", r["view"], "blocks", r["blocks_ran"], "span", r["span_us"], "sum_ms", r["sum_block_time_ms"], "util", r["slot_utilisation"], "ideal", r["ideal_span_if_all_slots_busy_us"],
          r["block_us_percentiles"], r["occupancy_of_4096_slots_by_time_decile"])
PY
done
echo "[$(( $(date +%s) - T0 )) s]"
