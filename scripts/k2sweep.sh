python -m pytest tests/test_parity_c_oracle_gpu.py "tests/test_contract_gpu.py::test_c5_full_size_vs_c_oracle" -m gpu -q -x 2>&1 | grep -B30 "^FAILED\|Error" | grep -v "^$" | tail -40
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for lb in 9 10 11 12; do for w in c3 c2; do rm -rf /tmp/p_$w; TEXGS_DEPTH_LB=$lb rocprofv3 --kernel-trace --stats -d /tmp/p_$w -o t --output-format csv -- python $R/bench.py --workload $w --steps 3 --warmup 3 --no-cpu-baseline --no-kernel-table --streams 1 > /dev/null 2>&1; python $R/scripts/k2_stats.py /tmp/p_$w "$w lb=$lb" | grep "depth\|dupl\|lb="; done; done
