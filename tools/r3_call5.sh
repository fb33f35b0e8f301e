#!/bin/bash
# Round 3, GPU call 5: sweep geometries (one emission per item), the multi-level body's per-phase clocks on the road
# stand-in (timers build), the multi-rank bench path after the transport / breakdown changes.
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python tools/ab_r3.py lj 20 > gpurun_out/c5_ab_lj.log 2>&1
timeout 300 python tools/ab_r3.py kron 10 > gpurun_out/c5_ab_kron.log 2>&1
timeout 400 python tools/ab_r3.py twitter 5 > gpurun_out/c5_ab_twitter.log 2>&1
for L in 1 2; do GRX_BIN_DEBUG=$L timeout 200 python tools/bin_debug.py lj > gpurun_out/c5_bin_debug_l$L.log 2>&1; done
GRX_LIB_PATH=$GRAFT_REPO_ROOT/gunrock_amd/libgrx_timers.so GRX_MID_DEBUG=1 timeout 400 python tools/ab_mid.py 2 bfs,ssspu,sssp > gpurun_out/c5_ab_mid_timers.log 2>&1
timeout 900 python -m pytest tests/test_distributed.py tests/test_bfs_gpu.py -x -q -m gpu -k "bench_multi or c5 or binned or single_rank" > gpurun_out/c5_pytest.log 2>&1
echo "pytest rc $?" >> gpurun_out/c5_pytest.log
tail -4 gpurun_out/c5_pytest.log; grep -v amdgpu gpurun_out/c5_ab_lj.log | cut -c1-330; grep -v amdgpu gpurun_out/c5_ab_mid_timers.log | cut -c1-250
