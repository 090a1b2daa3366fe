#!/bin/bash
cd $GRAFT_REPO_ROOT
TEXGS_LIB=$PWD/texture-gs_amd/libtexgs_stats.so timeout 300 python scripts/exp_stats.py c3 2>&1 | grep -v amdgpu.ids | tee gpurun_out/stats_c3.log
