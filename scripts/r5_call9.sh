#!/bin/bash
# Round 5, GPU call 9: texture-gradient reduce on a side stream next to K8 (TEXGS_BWD_OVERLAP) -- parity subset, then A/B
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
T0=$(date +%s)
timeout 600 python -m pytest tests/test_parity_gpu.py tests/test_contract_gpu.py "tests/test_parity_c_oracle_gpu.py::test_backward_full_size_vs_c_oracle" tests/test_gating_gpu.py \
   -m gpu -q -x -p no:cacheprovider 2>&1 | tail -15 > gpurun_out/c9_tests.log
echo "tests: $(tail -1 gpurun_out/c9_tests.log)  [$(( $(date +%s) - T0 )) s]"
grep -E "^FAILED|^ERROR" gpurun_out/c9_tests.log | head
for OV in 0 1 0 1; do
  TEXGS_BWD_OVERLAP=$OV timeout 300 python bench.py --no-cpu-baseline --steps 15 --warmup 4 --streams 1 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('overlap $OV serial', d['value'], d['ms_per_step_percentiles']['median'], {k: v for k, v in (d.get('reference_call_pattern') or {}).items() if k != 'note'}, {k: v for k, v in (d.get('reference_iteration') or {}).items() if k != 'note'})
"
done
echo "[$(( $(date +%s) - T0 )) s]"
for OV in 0 1 0 1; do
  TEXGS_BWD_OVERLAP=$OV timeout 300 python bench.py --no-cpu-baseline --no-kernel-table --steps 15 --warmup 4 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('overlap $OV pipelined', d['value'], d['ms_per_step_percentiles']['median'], {k: v for k, v in (d.get('reference_call_pattern') or {}).items() if k != 'note'})
"
done
echo "[$(( $(date +%s) - T0 )) s]"
for OV in 0 1; do
  TEXGS_BWD_OVERLAP=$OV timeout 300 python bench.py --leg iteration --steps 10 --warmup 3 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('overlap $OV iteration', d['uv_once_ms_per_iteration'], d['uv_per_render_ms_per_iteration'], d['split_uv_once_ms'])
"
done
echo "[$(( $(date +%s) - T0 )) s]"
