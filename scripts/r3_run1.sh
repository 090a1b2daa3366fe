#!/bin/bash
# round-3 first GPU call: A/B against the round-2 build, the GPU suite, short bench lines
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/parity_report.jsonl
echo "== A/B small"
timeout 300 python scripts/ab_compare.py dump build_exp/old/texture-gs_amd /tmp/old_s.pt > gpurun_out/ab_small.log 2>&1
timeout 300 python scripts/ab_compare.py dump texture-gs_amd /tmp/new_s.pt >> gpurun_out/ab_small.log 2>&1
timeout 120 python scripts/ab_compare.py cmp /tmp/new_s.pt /tmp/old_s.pt >> gpurun_out/ab_small.log 2>&1
tail -40 gpurun_out/ab_small.log
echo "== A/B c3-size"
timeout 300 python scripts/ab_compare.py dump build_exp/old/texture-gs_amd /tmp/old_l.pt 300000 1024 800 800 2 > gpurun_out/ab_large.log 2>&1
timeout 300 python scripts/ab_compare.py dump texture-gs_amd /tmp/new_l.pt 300000 1024 800 800 2 >> gpurun_out/ab_large.log 2>&1
timeout 120 python scripts/ab_compare.py cmp /tmp/new_l.pt /tmp/old_l.pt >> gpurun_out/ab_large.log 2>&1
tail -30 gpurun_out/ab_large.log
echo "== gpu tests"
timeout 1500 python -m pytest tests -m gpu -q -rA -x 2>&1 | grep -E "passed|failed|PASSED|FAILED|ERROR|Error|error|assert" | tail -60 > gpurun_out/gpu_tests.log
tail -25 gpurun_out/gpu_tests.log
echo "== bench"
timeout 400 python bench.py --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/bench_c3.json 2> gpurun_out/bench_c3.err
timeout 300 python bench.py --streams 1 --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/bench_c3_serial.json 2> gpurun_out/bench_c3_serial.err
timeout 300 python bench.py --workload c2 --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/bench_c2.json 2> gpurun_out/bench_c2.err
python - <<'PY'
import json
for n in ("c3", "c3_serial", "c2"):
    try:
        j = json.loads(open(f"gpurun_out/bench_{n}.json").read().strip().splitlines()[-1])
        print(n, j["value"], j["ms_per_step_percentiles"]["median"], {k: round(v["avg_us"]) for k, v in j["kernels"].items()},
              (j.get("reference_call_pattern") or {}).get("views_per_s"))
    except Exception as e:
        print(n, "ERR", e, open(f"gpurun_out/bench_{n}.err").read()[-1500:])
PY
