#!/bin/bash
# round-3 evidence: full GPU suite + bench lines (final_run.sh), then rocprof stats / PMC passes (r3_prof.sh)
cd $GRAFT_REPO_ROOT
bash scripts/final_run.sh 2>&1 | tail -12
bash scripts/r3_prof.sh > gpurun_out/r3_prof.log 2>&1
tail -5 gpurun_out/r3_prof.log
