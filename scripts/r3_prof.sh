#!/bin/bash
cd $GRAFT_REPO_ROOT
bash scripts/prof.sh > gpurun_out/prof_run.log 2>&1
python scripts/prof_summary.py gpurun_out/prof > gpurun_out/prof_summary.txt 2>&1
python scripts/make_traffic.py gpurun_out/prof gpurun_out/traffic.json > /dev/null 2>&1
tail -60 gpurun_out/prof_summary.txt
cat gpurun_out/traffic.json | head -60
bash scripts/prof_next_rows.sh 2>&1 | tail -14
