#!/bin/bash
# Round 3, GPU call 8: up to 480 bins (16-bit entries on the 21 M-vertex graph), BFS correctness.
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 400 python tools/ab_r3.py twitter 5 > gpurun_out/c8_ab_twitter.log 2>&1
timeout 300 python tools/ab_r3.py lj 20 > gpurun_out/c8_ab_lj.log 2>&1
timeout 300 python tools/ab_r3.py kron 10 > gpurun_out/c8_ab_kron.log 2>&1
timeout 900 python -m pytest tests/test_bfs_gpu.py tests/test_fuzz_gpu.py tests/test_distributed.py -x -q -m gpu > gpurun_out/c8_pytest.log 2>&1
echo "pytest rc $?" >> gpurun_out/c8_pytest.log
tail -4 gpurun_out/c8_pytest.log; grep -hv amdgpu gpurun_out/c8_ab_twitter.log gpurun_out/c8_ab_lj.log gpurun_out/c8_ab_kron.log | cut -c1-300
