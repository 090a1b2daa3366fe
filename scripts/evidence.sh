#!/bin/bash
# Round evidence on the GPU box, one call: the whole -m gpu suite, every bench line, the backward flavours, the iteration leg, the
# reference call pattern's profile, rocprofv3 kernel stats (pipelined + serial command) and the PMC passes.  Everything lands under
# gpurun_out/; copy what is to be judged to profiles/.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T0=$(date +%s)
rm -f gpurun_out/parity_report.jsonl
timeout 1500 python -m pytest tests -m gpu -q -rA --durations=15 -p no:cacheprovider 2>&1 | grep -E "passed|failed|PASSED|FAILED|ERROR|rror|^[0-9.]+s (call|setup)" | tail -160 > gpurun_out/gpu_tests.log
tail -3 gpurun_out/gpu_tests.log; grep -E "^FAILED|^ERROR" gpurun_out/gpu_tests.log | head
echo "[tests $(( $(date +%s) - T0 )) s]"
# the PMC traffic pass first: the bench lines below then carry roofline.traffic of THESE kernel sources
bash scripts/prof.sh > gpurun_out/prof.log 2>&1
python scripts/prof_summary.py gpurun_out/prof > gpurun_out/prof_summary.txt 2>&1
python scripts/make_traffic.py gpurun_out/prof gpurun_out/traffic.json > /dev/null 2>&1
cp gpurun_out/traffic.json profiles/r06_traffic.json
echo "[prof $(( $(date +%s) - T0 )) s]"
timeout 600 python bench.py > gpurun_out/bench_c3.json 2> gpurun_out/bench_c3.err
timeout 300 python bench.py --streams 1 --no-cpu-baseline > gpurun_out/bench_c3_serial.json 2> gpurun_out/bench_c3_serial.err
timeout 300 python bench.py --streams 4 --no-cpu-baseline --no-kernel-table > gpurun_out/bench_c3_s4.json 2> gpurun_out/bench_c3_s4.err
timeout 300 python bench.py --prefetch 0 --no-cpu-baseline --no-kernel-table --no-extra-legs > gpurun_out/bench_c3_noprefetch.json 2> gpurun_out/bench_c3_noprefetch.err
timeout 300 python bench.py --no-cpu-baseline --no-kernel-table --no-extra-legs > gpurun_out/bench_c3_again.json 2> gpurun_out/bench_c3_again.err
timeout 300 python bench.py --workload c2 --no-cpu-baseline > gpurun_out/bench_c2.json 2> gpurun_out/bench_c2.err
timeout 300 python bench.py --workload c5 --no-cpu-baseline --steps 8 --warmup 3 > gpurun_out/bench_c5.json 2> gpurun_out/bench_c5.err
timeout 300 python bench.py --surface diff_gauss --no-cpu-baseline > gpurun_out/bench_c3_diff_gauss.json 2> gpurun_out/bench_c3_diff_gauss.err
TEXGS_FORCE_DIST=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 timeout 300 python bench.py --no-cpu-baseline --no-kernel-table > gpurun_out/bench_c3_rccl1.json 2> gpurun_out/bench_c3_rccl1.err
timeout 300 python scripts/bench_variants.py > gpurun_out/variants.jsonl 2> gpurun_out/variants.err
timeout 300 python bench.py --leg iteration --steps 10 --warmup 3 > gpurun_out/bench_iteration_mixed.json 2> gpurun_out/bench_iteration_mixed.err
TEXGS_UV_PRECISION=fp32 timeout 300 python bench.py --leg iteration --steps 10 --warmup 3 > gpurun_out/bench_iteration_fp32.json 2> gpurun_out/bench_iteration_fp32.err
# the C3 geometry with a 302 MB texture (past the 256 MB Infinity Cache), serial and pipelined, beside R = 1024
timeout 300 python bench.py --tex-res 2048 --no-cpu-baseline --no-extra-legs > gpurun_out/bench_c3_R2048.json 2> gpurun_out/bench_c3_R2048.err
timeout 300 python bench.py --tex-res 2048 --streams 1 --no-cpu-baseline --no-extra-legs > gpurun_out/bench_c3_R2048_serial.json 2> gpurun_out/bench_c3_R2048_serial.err
TEXGS_ITEMS=1 timeout 300 python bench.py --no-cpu-baseline --no-extra-legs > gpurun_out/bench_c3_items.json 2> gpurun_out/bench_c3_items.err
echo "[bench $(( $(date +%s) - T0 )) s]"
bash scripts/prof_ref_pattern.sh > gpurun_out/ref_pattern_profile.txt 2>&1
echo "[ref pattern $(( $(date +%s) - T0 )) s]"
bash scripts/prof_iteration.sh > gpurun_out/iteration_profile.txt 2>&1
echo "[iteration profile $(( $(date +%s) - T0 )) s]"
bash scripts/prof_uv_backward.sh > gpurun_out/uv_kernels_profile.txt 2>&1
echo "[uv kernels profile $(( $(date +%s) - T0 )) s]"
# gpurun copies back at most 64 MiB: the raw per-launch traces and counter tables have been summarised above -- drop them
find gpurun_out -name "*kernel_trace.csv" -delete; find gpurun_out -name "*counter_collection.csv" -delete
find gpurun_out -name "*agent_info.csv" -delete; find gpurun_out -name "*.db" -delete; find gpurun_out -name "*.rocpd" -delete
du -sh gpurun_out | tail -1
python - <<'PY'
import json
for n in ("c3", "c3_again", "c3_noprefetch", "c3_serial", "c3_s4", "c2", "c5", "c3_diff_gauss", "c3_rccl1", "c3_R2048", "c3_R2048_serial", "c3_items"):
    try:
        j = json.loads([l for l in open(f"gpurun_out/bench_{n}.json").read().splitlines() if l.startswith("{")][-1])
        print(n, j["value"], j.get("value_long"), j["ms_per_step_percentiles"]["median"], {k: round(v["avg_us"]) for k, v in j["kernels"].items()},
              "cpu", (j.get("cpu_baseline") or {}).get("value"), "frac", (j.get("roofline") or {}).get("frac"), (j.get("roofline") or {}).get("solo_frac"),
              (j.get("roofline") or {}).get("frac_survey_8d"))
        for k in ("reference_call_pattern", "reference_iteration", "retexture_pattern"):
            if j.get(k): print("   ", k, {a: b for a, b in j[k].items() if a != "note"})
        if j["config"].get("grad_allreduce_measured"): print("   ", j["config"]["grad_allreduce_measured"])
    except Exception as e:
        print(n, "ERR", e, open(f"gpurun_out/bench_{n}.err").read()[-800:])
for n in ("mixed", "fp32"):
    try:
        j = json.loads([l for l in open(f"gpurun_out/bench_iteration_{n}.json").read().splitlines() if l.startswith("{")][-1])
        print("iteration", n, j["uv_per_render_ms_per_iteration"], j["uv_once_ms_per_iteration"], j["split_uv_once_ms"], j["rasterizer_kernels_us_per_iteration"], j.get("uvnet_backward_us"))
    except Exception as e:
        print("iteration", n, "ERR", e, open(f"gpurun_out/bench_iteration_{n}.err").read()[-800:])
PY
cat gpurun_out/variants.jsonl | cut -c1-400
cat gpurun_out/ref_pattern_profile.txt | head -50
cat gpurun_out/iteration_profile.txt | head -34
grep -E "k_uv_backward|uv_backward_" gpurun_out/uv_kernels_profile.txt | cut -c1-420
grep -E "k_render|k_texgrad|k_preprocess|k_bin_off|k_radix|k_depth|k_dup|k_ranges|k_tile" gpurun_out/prof_summary.txt | grep calls | head -40
cat gpurun_out/traffic.json | head -40
