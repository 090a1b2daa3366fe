#!/bin/bash
# after the last two changes (radix totals, single-pass k_bin_offsets): gate tests, then bench lines + rocprof / PMC refresh (no full suite)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_contract_gpu.py tests/test_parity_c_oracle_gpu.py -m gpu -q -x -k "texture_gradient or c5 or bit_exact or stress or 65536 or backward_full_size" 2>&1 | grep -E "passed|failed|FAILED|ERROR|rror" | tail -4 | tee gpurun_out/gate_tests.log
timeout 600 python bench.py > gpurun_out/bench_c3.json 2> gpurun_out/bench_c3.err
timeout 300 python bench.py --workload c2 --no-cpu-baseline > gpurun_out/bench_c2.json 2> gpurun_out/bench_c2.err
timeout 300 python bench.py --workload c5 --no-cpu-baseline --steps 8 --warmup 3 > gpurun_out/bench_c5.json 2> gpurun_out/bench_c5.err
timeout 300 python bench.py --streams 1 --no-cpu-baseline > gpurun_out/bench_c3_serial.json 2> gpurun_out/bench_c3_serial.err
python - <<'PY'
import json
for n in ("c3", "c2", "c5", "c3_serial"):
    try:
        j = json.loads(open(f"gpurun_out/bench_{n}.json").read().strip().splitlines()[-1])
        print(n, j["value"], j["ms_per_step_percentiles"]["median"], {k: round(v["avg_us"]) for k, v in j["kernels"].items()},
              (j.get("cpu_baseline") or {}).get("value"), (j.get("reference_call_pattern") or {}).get("views_per_s"), j["roofline"]["frac"], j["roofline"].get("solo_frac"))
    except Exception as e:
        print(n, "ERR", e, open(f"gpurun_out/bench_{n}.err").read()[-600:])
PY
bash scripts/r3_prof.sh > gpurun_out/r3_prof.log 2>&1
grep -E "k_render|k_texgrad|k_preprocess|radix|k_bin_off|scan" gpurun_out/prof_summary.txt | grep calls | tail -20
