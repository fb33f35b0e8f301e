#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_distributed_sssp.py tests/test_distributed_pr.py tests/test_distributed.py -m gpu -x -q > gpurun_out/cb_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/cb_pytest.log)
tail -30 gpurun_out/cb_pytest.log
