#!/bin/bash
# Round 2, first GPU call: A/B of the forward-BFS bodies, the GPU test suite, the full bench line.
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
nproc > gpurun_out/c1_host.log; free -g | head -2 >> gpurun_out/c1_host.log; rocm-smi --showmeminfo vram 2>/dev/null | head -8 >> gpurun_out/c1_host.log
timeout 400 python tools/ab_bfs.py lj bin > gpurun_out/c1_ab_lj.log 2>&1; echo "rc $?" >> gpurun_out/c1_ab_lj.log
(timeout 900 python -m pytest tests -m gpu -x -q -k "not full_size and not kron_c4" > gpurun_out/c1_pytest_fast.log 2>&1; echo "pytest rc $?" >> gpurun_out/c1_pytest_fast.log)
timeout 900 python bench.py > gpurun_out/c1_bench.log 2>&1; echo "rc $?" >> gpurun_out/c1_bench.log
timeout 400 python tools/ab_bfs.py kron bin > gpurun_out/c1_ab_kron.log 2>&1; echo "rc $?" >> gpurun_out/c1_ab_kron.log
(timeout 1200 python -m pytest tests -m gpu -x -q -k "full_size or kron_c4" > gpurun_out/c1_pytest_big.log 2>&1; echo "pytest rc $?" >> gpurun_out/c1_pytest_big.log)
tail -3 gpurun_out/c1_pytest_fast.log; tail -3 gpurun_out/c1_pytest_big.log; tail -12 gpurun_out/c1_ab_lj.log | cut -c1-260; tail -2 gpurun_out/c1_bench.log | cut -c1-600
