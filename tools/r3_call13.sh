#!/bin/bash
# Round 3, GPU call 13: bottom-up probes through the dense {first, second in-neighbour} array loaded with the row offsets.
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python tools/ab_r3.py lj 30 > gpurun_out/c13_ab_lj.log 2>&1
timeout 300 python tools/ab_r3.py kron 20 > gpurun_out/c13_ab_kron.log 2>&1
timeout 400 python tools/ab_r3.py twitter 10 > gpurun_out/c13_ab_twitter.log 2>&1
timeout 900 python -m pytest tests/test_bfs_gpu.py tests/test_fuzz_gpu.py -x -q -m gpu > gpurun_out/c13_pytest.log 2>&1
echo "pytest rc $?" >> gpurun_out/c13_pytest.log
tail -3 gpurun_out/c13_pytest.log; grep -hv amdgpu gpurun_out/c13_ab_lj.log gpurun_out/c13_ab_kron.log gpurun_out/c13_ab_twitter.log | cut -c1-300
