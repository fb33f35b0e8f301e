#!/bin/bash
# round 6, call 21: warp_mapped with packed short rows: operator tests, every CLI combination, the generic operators' BFS times on the LJ stand-in
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
T0=$(date +%s); el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
timeout 300 bin/test_operators > gpurun_out/r6_c21_test_operators.log 2>&1; el "test_operators rc $?"; tail -3 gpurun_out/r6_c21_test_operators.log
(timeout 900 python -m pytest tests/test_cli.py -m gpu -q -x --durations=4 > gpurun_out/r6_c21_pytest_cli.log 2>&1; echo "pytest rc $?" >> gpurun_out/r6_c21_pytest_cli.log); el pytest
tail -6 gpurun_out/r6_c21_pytest_cli.log
timeout 600 bash tools/bench_generic.sh > gpurun_out/r6_c21_generic.txt 2>&1; el generic
cat gpurun_out/generic_bfs.log
