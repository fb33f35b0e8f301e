#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
(timeout 600 python -m pytest tests/test_bfs_gpu.py tests/test_fuzz_gpu.py -m gpu -x -q -k "not full_size" > gpurun_out/c5_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c5_pytest.log)
timeout 300 python tools/ab_bfs.py lj bin > gpurun_out/c5_ab_lj.log 2>&1
timeout 300 python tools/ab_bfs.py kron bin > gpurun_out/c5_ab_kron.log 2>&1
GRX_BIN_DEBUG=1 timeout 200 python tools/bin_debug.py lj > gpurun_out/c5_dbg_l1.log 2>&1
GRX_BIN_DEBUG=2 timeout 200 python tools/bin_debug.py lj > gpurun_out/c5_dbg_l2.log 2>&1
tail -3 gpurun_out/c5_pytest.log; grep "TD bins (default\|round-1\|DO default" gpurun_out/c5_ab_lj.log gpurun_out/c5_ab_kron.log | cut -c1-330; grep -h "span\|xcc 0" gpurun_out/c5_dbg_l1.log gpurun_out/c5_dbg_l2.log
