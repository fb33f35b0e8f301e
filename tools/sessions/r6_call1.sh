#!/bin/bash
# round 6, call 1: the two first changes (mid body's hub exit, closed0/bu_heads publish) on the GPU + the baseline bench line
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
T0=$(date +%s); el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
(timeout 500 python -m pytest tests/test_bfs_gpu.py tests/test_mid_gpu.py -m gpu -q -x --durations=5 -k "plan_rules or race or two_contexts or direction or mid or launch_groups" > gpurun_out/r6_c1_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r6_c1_pytest.log); el pytest
tail -5 gpurun_out/r6_c1_pytest.log
timeout 400 python bench.py > gpurun_out/r6_c1_bench.log 2> gpurun_out/r6_c1_bench.err; echo "rc $?" >> gpurun_out/r6_c1_bench.log; el bench
cp gpurun_out/bench_detail.json gpurun_out/r6_c1_bench_detail.json 2>/dev/null
head -c 1500 gpurun_out/r6_c1_bench.log; echo; tail -c 200 gpurun_out/r6_c1_bench.log
