#!/bin/bash
# Round 5, GPU call 10: K7 issue priority by block length (prio1: static by survivors, prio2: by remaining survivors) vs none
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TEXGS_ABI_ANY=1
T0=$(date +%s)
bash scripts/ab_serial.sh libtexgs_base.so libtexgs_prio1.so libtexgs_prio2.so libtexgs_base.so libtexgs_prio1.so libtexgs_prio2.so 2>&1 | grep -v amdgpu.ids | tee gpurun_out/c10_ab.log
echo "[$(( $(date +%s) - T0 )) s]"
bash scripts/pipe.sh "" libtexgs_base.so libtexgs_prio1.so libtexgs_prio2.so libtexgs_base.so libtexgs_prio1.so libtexgs_prio2.so 2>&1 | tee gpurun_out/c10_pipe.log
echo "[$(( $(date +%s) - T0 )) s]"
for L in trace trace1 trace2; do
  TEXGS_LIB=$GRAFT_REPO_ROOT/texture-gs_amd/libtexgs_$L.so timeout 200 python scripts/k7_trace.py 2>/dev/null > gpurun_out/c10_$L.json
  python - <<PY
import json
d = json.load(open("gpurun_out/c10_$L.json"))
for r in d["k7_trace"]:
    print("$L", r["view"], "span", r["span_us"], "sum_ms", r["sum_block_time_ms"], "util", r["slot_utilisation"], "ideal", r["ideal_span_if_all_slots_busy_us"],
          r["block_us_percentiles"], r["occupancy_of_4096_slots_by_time_decile"])
PY
done
echo "[$(( $(date +%s) - T0 )) s]"
