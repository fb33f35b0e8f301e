#!/bin/bash
# round 5, call 38: the GPU test of the two plan rules after its correction (the first rule is a rule of the heads: a profiled run
# enters the many-levels body at level 0 and cannot show it)
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
(timeout 100 python -m pytest "tests/test_bfs_gpu.py::test_plan_rules_for_searches_from_low_degree_sources" -m gpu -q > gpurun_out/r5c38_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r5c38_pytest.log)
tail -25 gpurun_out/r5c38_pytest.log
