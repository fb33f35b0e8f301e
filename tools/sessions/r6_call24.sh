#!/bin/bash
# round 6, call 24: timer with its own marker command, warp_mapped's long rows 4 neighbours per lane and trip: generic operators' BFS times,
# operator tests, the reference's k-core / PPR on both operator sets
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
T0=$(date +%s); el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
timeout 300 bin/test_operators > gpurun_out/r6_c24_test_operators.log 2>&1; el "test_operators rc $?"; tail -1 gpurun_out/r6_c24_test_operators.log
timeout 600 bash tools/bench_generic.sh > gpurun_out/r6_c24_generic.txt 2>&1; el generic
cat gpurun_out/generic_bfs.log
timeout 900 bash tools/bench_refalg.sh > gpurun_out/r6_c24_refalg.txt 2>&1; el refalg
cat gpurun_out/refalg_times.txt
(timeout 900 python -m pytest tests/test_cli.py -m gpu -q -x --durations=4 > gpurun_out/r6_c24_pytest_cli.log 2>&1; echo "pytest rc $?" >> gpurun_out/r6_c24_pytest_cli.log); el pytest
tail -4 gpurun_out/r6_c24_pytest_cli.log
