#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 300 python tools/ab_bfs.py lj recheck > gpurun_out/c6_ab_lj.log 2>&1
timeout 300 python tools/ab_bfs.py kron recheck > gpurun_out/c6_ab_kron.log 2>&1
grep "TD " gpurun_out/c6_ab_lj.log gpurun_out/c6_ab_kron.log | cut -c1-300
