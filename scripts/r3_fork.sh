#!/bin/bash
# round 3: reduce beside K8 (fork/join inside texgs_backward) -- tests, bench lines, and where K6's FETCH_SIZE comes from
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_contract_gpu.py tests/test_parity_c_oracle_gpu.py tests/test_ws2_gpu.py -m gpu -q -x 2>&1 | grep -E "passed|failed|FAILED|ERROR|rror" | tail -6
for S in 3 1; do
timeout 400 python bench.py --no-cpu-baseline --steps 10 --warmup 3 --streams $S 2> gpurun_out/bench_c3_s$S.err | tee gpurun_out/bench_c3_s$S.json | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('c3 streams $S', d['value'], {k:round(v['avg_us']) for k,v in d['kernels'].items()}, d['reference_call_pattern']['views_per_s'])
"
tail -2 gpurun_out/bench_c3_s$S.err
done
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for V in libtexgs.so libtexgs_fnodense.so libtexgs_fnotest.so; do
  OUT=$R/gpurun_out/fetchq_$(basename $V .so); rm -rf $OUT; mkdir -p $OUT
  TEXGS_LIB=$R/texture-gs_amd/$V rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT -o p --output-format csv -- python $R/bench.py --steps 1 --warmup 2 --no-cpu-baseline --no-kernel-table --streams 1 > /dev/null 2>&1
  python - <<PY
import csv, glob
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "k_render_fwd" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
    print("$V", "k_render_fwd FETCH_SIZE x2 MB per launch", round(sum(v) / max(len(v), 1) * 2 * 1024 / 1e6, 1), "launches", len(v))
PY
done
