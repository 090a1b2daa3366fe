#!/bin/bash
# GPU box: FETCH_SIZE / WRITE_SIZE of the calibration kernels (separate --pmc passes, kernel trace only) -> gpurun_out/traffic_cal.json
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/traffic_cal
rm -rf $OUT; mkdir -p $OUT
$R/scripts/ubench/traffic_cal > $OUT/timing.jsonl
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/f -o f --output-format csv -- $R/scripts/ubench/traffic_cal > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/w -o w --output-format csv -- $R/scripts/ubench/traffic_cal > /dev/null 2>&1
python $R/scripts/traffic_cal_summary.py $OUT $R/gpurun_out/traffic_cal.json
