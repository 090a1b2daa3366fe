cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python bench.py --streams 1 --no-cpu-baseline > gpurun_out/bench_c3_serial.json 2> gpurun_out/bench_c3_serial.err
timeout 300 python bench.py --streams 4 --no-cpu-baseline --no-kernel-table > gpurun_out/bench_c3_s4.json 2> gpurun_out/bench_c3_s4.err
timeout 300 python bench.py --workload c2 --no-cpu-baseline > gpurun_out/bench_c2.json 2> gpurun_out/bench_c2.err
TEXGS_FORCE_DIST=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 timeout 300 python bench.py --no-cpu-baseline --no-kernel-table > gpurun_out/bench_c3_rccl1.json 2> gpurun_out/bench_c3_rccl1.err
TEXGS_ITEMS=1 timeout 300 python bench.py --no-cpu-baseline --no-extra-legs > gpurun_out/bench_c3_items.json 2> gpurun_out/bench_c3_items.err
python - <<'PY'
import json
for n in ("c3_serial","c3_s4","c2","c3_rccl1","c3_items"):
    j=json.loads([l for l in open(f"gpurun_out/bench_{n}.json").read().splitlines() if l.startswith("{")][-1])
    print(n, j["value"], j["value_long"], j["ms_per_step_percentiles"], (j.get("reference_call_pattern") or {}).get("views_per_s"), (j.get("reference_iteration") or {}).get("shared_geometry_ms_per_iteration"), (j.get("retexture_pattern") or {}).get("ratio"))
PY
