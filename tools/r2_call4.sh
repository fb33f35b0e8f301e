#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
GRX_BIN_DEBUG=1 timeout 200 python tools/bin_debug.py lj > gpurun_out/c4_dbg_l1.log 2>&1
GRX_BIN_DEBUG=2 timeout 200 python tools/bin_debug.py lj > gpurun_out/c4_dbg_l2.log 2>&1
cat gpurun_out/c4_dbg_l1.log gpurun_out/c4_dbg_l2.log
