#!/bin/bash
# Round 3, GPU call 1: reference-made PageRank goldens, PR parity at equal iterations, C5' (configs[4]) on one GPU
# and on two ranks sharing it, the reference GPU path on C5', the new bench line.
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export TMPDIR=/tmp
python tests/golden/make_golden_pr.py > gpurun_out/c1_golden_pr.log 2>&1
cp gpurun_out/golden_pr.npz tests/golden/golden_pr.npz 2>/dev/null
timeout 900 python -m pytest tests/test_pr_gpu.py -x -q -m gpu -s > gpurun_out/c1_pytest_pr.log 2>&1
echo "pytest pr rc $?" >> gpurun_out/c1_pytest_pr.log
timeout 600 python -m pytest tests/test_oracle_golden.py -x -q > gpurun_out/c1_pytest_oracle.log 2>&1
echo "pytest oracle rc $?" >> gpurun_out/c1_pytest_oracle.log
timeout 900 python -m pytest tests/test_bfs_gpu.py -x -q -m gpu -k "twitter" -s > gpurun_out/c1_pytest_twitter.log 2>&1
echo "pytest twitter rc $?" >> gpurun_out/c1_pytest_twitter.log
timeout 1500 python -m pytest tests/test_distributed.py -x -q -m gpu -k "c5" -s > gpurun_out/c1_pytest_c5dist.log 2>&1
echo "pytest c5 dist rc $?" >> gpurun_out/c1_pytest_c5dist.log
timeout 900 python tests/tools/bench_all.py bfs_twitter > gpurun_out/c1_bench_all_twitter.log 2>&1
timeout 1200 python bench.py > gpurun_out/c1_bench.log 2>&1
echo "bench rc $?" >> gpurun_out/c1_bench.log
tail -3 gpurun_out/c1_*.log
