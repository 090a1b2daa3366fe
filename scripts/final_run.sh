#!/bin/bash
# Round-end measurement on the GPU box: the whole -m gpu suite, then the three bench lines (C3 with the CPU baseline).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/parity_report.jsonl
timeout 1500 python -m pytest tests -m gpu -q -rA 2>&1 | grep -E "passed|failed|PASSED|FAILED|ERROR|error" | tail -80 > gpurun_out/gpu_tests.log
tail -3 gpurun_out/gpu_tests.log
timeout 600 python bench.py > gpurun_out/bench_c3.json 2> gpurun_out/bench_c3.err
timeout 300 python bench.py --workload c2 --no-cpu-baseline > gpurun_out/bench_c2.json 2> gpurun_out/bench_c2.err
timeout 300 python bench.py --workload c5 --no-cpu-baseline --steps 8 --warmup 3 > gpurun_out/bench_c5.json 2> gpurun_out/bench_c5.err
timeout 300 python bench.py --streams 1 --no-cpu-baseline > gpurun_out/bench_c3_serial.json 2> gpurun_out/bench_c3_serial.err
python - <<'PY'
import json
for n in ("c3", "c2", "c5", "c3_serial"):
    try:
        j = json.loads(open(f"gpurun_out/bench_{n}.json").read().strip().splitlines()[-1])
        print(n, j["value"], j["ms_per_step_percentiles"]["median"], j["roofline"], {k: round(v["avg_us"]) for k, v in j["kernels"].items()},
              (j.get("cpu_baseline") or {}).get("value"), (j.get("reference_call_pattern") or {}).get("views_per_s"))
    except Exception as e:
        print(n, "ERR", e, open(f"gpurun_out/bench_{n}.err").read()[-600:])
PY
