#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
(timeout 600 python -m pytest tests/test_bfs_gpu.py -m gpu -x -q > gpurun_out/cb_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/cb_pytest.log)
timeout 300 python tools/ab_bfs.py lj claim > gpurun_out/cb_ab_lj.log 2>&1; echo "rc $?" >> gpurun_out/cb_ab_lj.log
timeout 300 python tools/ab_bfs.py kron claim > gpurun_out/cb_ab_kron.log 2>&1; echo "rc $?" >> gpurun_out/cb_ab_kron.log
GRX_BIN_DEBUG=1 timeout 200 python tools/bin_debug.py lj > gpurun_out/cb_dbg_l1.log 2>&1
GRX_BIN_DEBUG=2 timeout 200 python tools/bin_debug.py lj > gpurun_out/cb_dbg_l2.log 2>&1
tail -3 gpurun_out/cb_pytest.log; grep "sweep claim\|rc " gpurun_out/cb_ab_lj.log gpurun_out/cb_ab_kron.log | cut -c1-400
grep -A2 "^scatter" gpurun_out/cb_dbg_l1.log | cut -c1-330; grep -A2 "^scatter" gpurun_out/cb_dbg_l2.log | cut -c1-330
