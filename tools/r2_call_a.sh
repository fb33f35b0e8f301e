#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_bfs_gpu.py tests/test_sssp_gpu.py -m gpu -x -q > gpurun_out/ca_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/ca_pytest.log)
timeout 300 python tools/ab_bfs.py lj claim > gpurun_out/ca_ab_lj.log 2>&1; echo "rc $?" >> gpurun_out/ca_ab_lj.log
timeout 300 python tools/ab_bfs.py kron claim > gpurun_out/ca_ab_kron.log 2>&1; echo "rc $?" >> gpurun_out/ca_ab_kron.log
GRX_MID_DEBUG=1 timeout 400 python tools/ab_mid.py 2 > gpurun_out/ca_ab_mid.log 2>&1; echo "rc $?" >> gpurun_out/ca_ab_mid.log
GRX_BIN_DEBUG=1 timeout 200 python tools/bin_debug.py lj > gpurun_out/ca_dbg_l1.log 2>&1
GRX_BIN_DEBUG=2 timeout 200 python tools/bin_debug.py lj > gpurun_out/ca_dbg_l2.log 2>&1
tail -3 gpurun_out/ca_pytest.log; cat gpurun_out/ca_ab_lj.log gpurun_out/ca_ab_kron.log gpurun_out/ca_ab_mid.log | cut -c1-400; tail -22 gpurun_out/ca_dbg_l1.log | cut -c1-330; tail -22 gpurun_out/ca_dbg_l2.log | cut -c1-330
