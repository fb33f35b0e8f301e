#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_mid_gpu.py tests/test_bfs_gpu.py tests/test_sssp_gpu.py -m gpu -x -q > gpurun_out/cb_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/cb_pytest.log)
timeout 400 python tools/ab_mid.py 3 2>&1 | grep -v amdgpu > gpurun_out/cb_ab_mid.log
timeout 300 python tools/ab_bfs.py lj claim do 2>&1 | grep -v amdgpu > gpurun_out/cb_ab_lj.log
timeout 300 python tools/ab_bfs.py kron do 2>&1 | grep -v amdgpu > gpurun_out/cb_ab_kron.log
tail -3 gpurun_out/cb_pytest.log; cat gpurun_out/cb_ab_mid.log; grep "DO default\|sweep claim" gpurun_out/cb_ab_lj.log gpurun_out/cb_ab_kron.log | cut -c1-350
