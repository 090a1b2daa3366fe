#!/bin/bash
# Ablation table for K7 / K6 on the GPU box (run through gpurun).  Needs texture-gs_amd/libtexgs_exp.so =
#   TEXGS_LIB_NAME=libtexgs_exp.so TEXGS_OBJ_DIR=build_exp TEXGS_EXTRA_FLAGS=-DTEXGS_EXPERIMENTS python texture-gs_amd/build.py
# Each line: which part was compiled OUT (timing only: results are wrong by construction), views/s and per-kernel avg us.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/ablation.jsonl
: > $OUT
run() {  # label, env...
  label=$1; shift
  env TEXGS_LIB=$R/texture-gs_amd/libtexgs_exp.so "$@" timeout 300 python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | \
    python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        print(json.dumps({'variant': '$label', 'views_per_s': d['value'], 'kernel_avg_us': {k: round(v['avg_us'], 1) for k, v in d['kernels'].items()}}))
" | tee -a $OUT
}
run "product (nothing ablated)" TEXGS_ABLATE=0
run "K7 without texture-gradient atomics (ABL 1)" TEXGS_ABLATE=1
run "K7 without stage-C butterfly reduce (ABL 2)" TEXGS_ABLATE=2
run "K7 without atomics and without reduce (ABL 3)" TEXGS_ABLATE=3
run "K7 without texel tap loads (ABL 8)" TEXGS_ABLATE=8
run "K7 without tap loads, atomics, reduce (ABL 11)" TEXGS_ABLATE=11
run "K6 without dense texture phase (FABL 1)" TEXGS_FWD_ABLATE=1
run "K6 dense phase without tap loads (FABL 2)" TEXGS_FWD_ABLATE=2
