#!/bin/bash
# Round 3, GPU call 9: sweep with deeper candidate prefetch and wider emission groups.
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python tools/ab_r3.py lj 20 > gpurun_out/c9_ab_lj.log 2>&1
timeout 300 python tools/ab_r3.py kron 10 > gpurun_out/c9_ab_kron.log 2>&1
timeout 400 python tools/ab_r3.py twitter 5 > gpurun_out/c9_ab_twitter.log 2>&1
for L in 1 2; do GRX_BIN_DEBUG=$L timeout 200 python tools/bin_debug.py lj > gpurun_out/c9_bin_debug_l$L.log 2>&1; done
timeout 900 python -m pytest tests/test_bfs_gpu.py tests/test_fuzz_gpu.py -x -q -m gpu > gpurun_out/c9_pytest.log 2>&1
echo "pytest rc $?" >> gpurun_out/c9_pytest.log
tail -3 gpurun_out/c9_pytest.log; grep -hv amdgpu gpurun_out/c9_ab_lj.log gpurun_out/c9_ab_kron.log gpurun_out/c9_ab_twitter.log | cut -c1-300; for L in 1 2; do grep claim -A2 gpurun_out/c9_bin_debug_l$L.log | cut -c1-330 | head -3; done
