#!/bin/bash
# Profile bench.py on the GPU box: kernel trace + stats of the pipelined command and of the serial one, then PMC passes of the
# serial one (separate runs, as required; PMC collection serialises the kernels anyway).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
PIPE="--steps 6 --warmup 3 --no-cpu-baseline --no-kernel-table"
SER="--steps 3 --warmup 3 --no-cpu-baseline --no-kernel-table --streams 1"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace --output-format csv -- python $R/bench.py $PIPE > $OUT/bench_trace.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/trace_serial -o trace --output-format csv -- python $R/bench.py $SER > $OUT/bench_trace_serial.log 2>&1
PMC="--steps 1 --warmup 2 --no-cpu-baseline --no-kernel-table --streams 1"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --kernel-trace -d $OUT/pmc1 -o pmc1 --output-format csv -- python $R/bench.py $PMC > $OUT/bench_pmc1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --kernel-trace -d $OUT/pmc2 -o pmc2 --output-format csv -- python $R/bench.py $PMC > $OUT/bench_pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc3 -o pmc3 --output-format csv -- python $R/bench.py $PMC > $OUT/bench_pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $OUT/pmc4 -o pmc4 --output-format csv -- python $R/bench.py $PMC > $OUT/bench_pmc4.log 2>&1
find $OUT -name "*.csv" | head -30
tail -2 $OUT/bench_trace.log $OUT/bench_trace_serial.log $OUT/bench_pmc1.log
