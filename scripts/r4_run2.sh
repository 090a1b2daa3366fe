#!/bin/bash
# Round 4, validation call: whole -m gpu suite, then the bench lines (C3 with its new legs, C2 with the retexture leg, the
# untextured surface, the 1-rank RCCL rehearsal that times the two-segment all-reduce).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/parity_report.jsonl
timeout 1500 python -m pytest tests -m gpu -q -rA 2>&1 | grep -E "passed|failed|PASSED|FAILED|ERROR|rror|assert" | tail -150 > gpurun_out/gpu_tests.log
tail -12 gpurun_out/gpu_tests.log; grep -c PASSED gpurun_out/gpu_tests.log
timeout 400 python bench.py --no-cpu-baseline > gpurun_out/bench_c3.json 2> gpurun_out/bench_c3.err
timeout 300 python bench.py --workload c2 --no-cpu-baseline > gpurun_out/bench_c2.json 2> gpurun_out/bench_c2.err
timeout 300 python bench.py --surface diff_gauss --no-cpu-baseline > gpurun_out/bench_c3_diff_gauss.json 2> gpurun_out/bench_c3_diff_gauss.err
TEXGS_FORCE_DIST=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 timeout 300 python bench.py --no-cpu-baseline --no-kernel-table > gpurun_out/bench_c3_rccl1.json 2> gpurun_out/bench_c3_rccl1.err
python - <<'PY'
import json
for n in ("c3", "c2", "c3_diff_gauss", "c3_rccl1"):
    try:
        j = json.loads([l for l in open(f"gpurun_out/bench_{n}.json").read().splitlines() if l.startswith("{")][-1])
        print(n, j["value"], {k: round(v["avg_us"]) for k, v in j["kernels"].items()})
        for k in ("reference_call_pattern", "reference_iteration", "retexture_pattern"):
            if j.get(k): print("   ", k, {a: b for a, b in j[k].items() if a != "note"})
        if j["config"].get("grad_allreduce_measured"): print("   ", j["config"]["grad_allreduce_measured"])
    except Exception as e:
        print(n, "ERR", e, open(f"gpurun_out/bench_{n}.err").read()[-1500:])
PY
