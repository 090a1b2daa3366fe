#!/bin/bash
# Round 4, first GPU call: whole -m gpu suite on the gated / byte-offset kernels, C3 bench (pipelined + serial), backward flavours.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/parity_report.jsonl
timeout 1200 python -m pytest tests -m gpu -q -x -rA 2>&1 | grep -E "passed|failed|PASSED|FAILED|ERROR|rror|assert" | tail -120 > gpurun_out/gpu_tests.log
tail -15 gpurun_out/gpu_tests.log
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/bench_c3.json 2> gpurun_out/bench_c3.err
timeout 300 python bench.py --streams 1 --no-cpu-baseline > gpurun_out/bench_c3_serial.json 2> gpurun_out/bench_c3_serial.err
timeout 300 python scripts/bench_variants.py > gpurun_out/variants.jsonl 2> gpurun_out/variants.err
cat gpurun_out/variants.jsonl; tail -3 gpurun_out/variants.err
python - <<'PY'
import json
for n in ("c3", "c3_serial"):
    try:
        j = json.loads(open(f"gpurun_out/bench_{n}.json").read().strip().splitlines()[-1])
        print(n, j["value"], j["ms_per_step_percentiles"]["median"], j["roofline"], {k: round(v["avg_us"]) for k, v in j["kernels"].items()},
              (j.get("reference_call_pattern") or {}).get("views_per_s"))
    except Exception as e:
        print(n, "ERR", e, open(f"gpurun_out/bench_{n}.err").read()[-800:])
PY
