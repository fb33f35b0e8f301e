#!/bin/bash
# round 5, call 14: the default bench line on the current sources (with the new bfs_deep section), the multi-rank bench test
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
T0=$(date +%s); el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
timeout 500 python bench.py > gpurun_out/r5c14_bench.log 2> gpurun_out/r5c14_bench.err; echo "rc $?" >> gpurun_out/r5c14_bench.log
cp gpurun_out/bench_detail.json gpurun_out/r5c14_bench_detail.json 2>/dev/null; el "bench"
(timeout 200 python -m pytest -q -x -m gpu tests/test_distributed.py tests/test_bench_data.py --deselect tests/test_distributed.py::test_c5_twitter_standin_two_ranks_one_gpu --deselect tests/test_distributed.py::test_c5_twitter_standin_eight_ranks_one_gpu > gpurun_out/r5c14_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r5c14_pytest.log)
el "pytest"
head -c 6000 gpurun_out/r5c14_bench.log; echo; tail -c 600 gpurun_out/r5c14_bench.err; tail -5 gpurun_out/r5c14_pytest.log
