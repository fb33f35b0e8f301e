#!/bin/bash
# round 5, call 27: generic operators -- small frontiers sized / filtered in ONE launch with the size written to the host's mailbox:
# operator unit tests + CLI tests (the reference's drivers and algorithm headers on these operators), then the timings
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
T0=$(date +%s); el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
(timeout 120 bin/test_operators > gpurun_out/r5c27_test_operators.log 2>&1; echo "rc $?" >> gpurun_out/r5c27_test_operators.log); el "test_operators: $(tail -2 gpurun_out/r5c27_test_operators.log | tr '\n' ' ')"
(timeout 600 python -m pytest -q -m gpu tests/test_cli.py tests/test_operators_gpu.py > gpurun_out/r5c27_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r5c27_pytest.log)
el "pytest: $(tail -3 gpurun_out/r5c27_pytest.log | tr '\n' ' ')"
timeout 300 bash tools/bench_refalg.sh > gpurun_out/r5c27_refalg.log 2>&1; el "refalg"
timeout 300 bash tools/bench_generic.sh > gpurun_out/r5c27_generic.log 2>&1; el "generic"
tail -6 gpurun_out/r5c27_pytest.log | cut -c1-200; grep -v "^rc 0" gpurun_out/refalg_times.txt; head -20 gpurun_out/generic_bfs.log
