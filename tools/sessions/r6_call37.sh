#!/bin/bash
# round 6, call 37: the log2 degree histogram of a frontier taken by the pass that scans its degrees (bucketing: no launch and no read-back of its own)
# operator tests, generic BFS times, k-core / PPR, CLI tests
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
T0=$(date +%s); el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
timeout 300 bin/test_operators > gpurun_out/r6_c37_test_operators.log 2>&1; el "test_operators rc $?"; tail -1 gpurun_out/r6_c37_test_operators.log
timeout 600 bash tools/bench_generic.sh > gpurun_out/r6_c37_generic.txt 2>&1; el generic
cat gpurun_out/generic_bfs.log
timeout 900 bash tools/bench_refalg.sh > gpurun_out/r6_c37_refalg.txt 2>&1; el refalg
grep -A3 "^==" gpurun_out/refalg_times.txt | grep "==\|Elapsed\|errors"
(timeout 900 python -m pytest tests/test_cli.py -m gpu -q -x --durations=4 > gpurun_out/r6_c37_pytest_cli.log 2>&1; echo "pytest rc $?" >> gpurun_out/r6_c37_pytest_cli.log); el pytest
tail -3 gpurun_out/r6_c37_pytest_cli.log
