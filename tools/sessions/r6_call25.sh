#!/bin/bash
# round 6, call 25: block-asynchronous path moved to libgrx_block.so (tests on the variant with GRX_TEST_BLOCK=1), lazy builds fail softly,
# PageRank sorted-blocks test, only_head not under max_iterations: the affected GPU tests
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
(GRX_TEST_BLOCK=1 timeout 1500 python -m pytest tests/test_block_variant.py tests/test_block_gpu.py tests/test_sssp_gpu.py tests/test_pr_gpu.py tests/test_bfs_gpu.py tests/test_mid_gpu.py -m gpu -q -x --durations=6 > gpurun_out/r6_c25_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r6_c25_pytest.log)
tail -14 gpurun_out/r6_c25_pytest.log
