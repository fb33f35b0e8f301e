#!/bin/bash
# round 5, call 37: the GPU test of the two plan rules + the fuzz cases that the cut-down evidence run left out, on the final sources
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
(timeout 280 python -m pytest "tests/test_bfs_gpu.py::test_plan_rules_for_searches_from_low_degree_sources" tests/test_fuzz_gpu.py -m gpu -q --durations=5 > gpurun_out/r5c37_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r5c37_pytest.log)
tail -25 gpurun_out/r5c37_pytest.log
