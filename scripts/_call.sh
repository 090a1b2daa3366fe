cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python bench.py > gpurun_out/bench_c3.json 2> gpurun_out/bench_c3.err
python - <<PY
import json
j=json.loads([l for l in open("gpurun_out/bench_c3.json").read().splitlines() if l.startswith("{")][-1])
print("final", j["value"], j["value_long"], j["ms_per_step_percentiles"], j["build_id"], (j["reference_call_pattern"] or {}).get("views_per_s"), (j.get("roofline") or {}).get("traffic"))
PY
