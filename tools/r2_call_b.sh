#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_mid_gpu.py tests/test_bfs_gpu.py -m gpu -x -q > gpurun_out/cb_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/cb_pytest.log)
timeout 300 python tools/ab_bfs.py lj claim > gpurun_out/cb_ab_lj.log 2>&1; echo "rc $?" >> gpurun_out/cb_ab_lj.log
timeout 300 python tools/ab_bfs.py kron claim > gpurun_out/cb_ab_kron.log 2>&1; echo "rc $?" >> gpurun_out/cb_ab_kron.log
tail -15 gpurun_out/cb_pytest.log; grep "sweep claim\|rc " gpurun_out/cb_ab_lj.log gpurun_out/cb_ab_kron.log | cut -c1-400
