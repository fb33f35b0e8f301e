#!/bin/bash
# round 6, call 2: the partition on the engine's bodies -- first run on the GPU
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
T0=$(date +%s); el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
(timeout 900 python -m pytest tests/test_distributed.py -m gpu -q -x --durations=8 --deselect tests/test_distributed.py::test_c5_twitter_standin_eight_ranks_one_gpu --deselect tests/test_distributed.py::test_c5_twitter_standin_two_ranks_one_gpu > gpurun_out/r6_c2_pytest_dist.log 2>&1; echo "pytest rc $?" >> gpurun_out/r6_c2_pytest_dist.log); el pytest-dist
tail -25 gpurun_out/r6_c2_pytest_dist.log
(timeout 600 python -m pytest tests/test_bfs_gpu.py tests/test_target_matrix_gpu.py -m gpu -q -x --durations=5 -k "not twitter" > gpurun_out/r6_c2_pytest_bfs.log 2>&1; echo "pytest rc $?" >> gpurun_out/r6_c2_pytest_bfs.log); el pytest-bfs
tail -8 gpurun_out/r6_c2_pytest_bfs.log
