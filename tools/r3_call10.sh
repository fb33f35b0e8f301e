#!/bin/bash
# Round 3, GPU call 10: generic operators with the fused advance + compact; operator-level and CLI tests.
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 bin/test_operators > gpurun_out/c10_test_operators.log 2>&1; echo "test_operators rc $?" >> gpurun_out/c10_test_operators.log
timeout 600 bash tools/bench_generic.sh > gpurun_out/c10_generic.log 2>&1
timeout 1200 python -m pytest tests/test_cli.py -x -q -m gpu > gpurun_out/c10_pytest_cli.log 2>&1; echo "pytest rc $?" >> gpurun_out/c10_pytest_cli.log
tail -3 gpurun_out/c10_test_operators.log; cat gpurun_out/generic_bfs.log; head -8 gpurun_out/generic_kernel_stats.md | cut -c1-200; tail -3 gpurun_out/c10_pytest_cli.log
