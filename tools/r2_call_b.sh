#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_pr_gpu.py tests/test_distributed_pr.py -m gpu -x -q > gpurun_out/cb_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/cb_pytest.log)
timeout 300 python tools/run_algo.py pr kron 5 2>&1 | grep -v amdgpu > gpurun_out/cb_pr.log
tail -4 gpurun_out/cb_pytest.log; cat gpurun_out/cb_pr.log | cut -c1-300
